// pk_opsel_probe.hip -- does v_pk_mul_f32 with op_sel:[0,1] (low half takes src1's HIGH register) return wrong low
// halves on gfx950, and beside what?  Follow-up of tools/sg_fault_repro.hip, which pinned the split-GEMM epilogue fault
// of round 4 to exactly that instruction form (variants 11-16 there): low half = 0 in lanes 48-63, a few hundred times
// per 8.4 M executions, while the same product without op_sel, or with op_sel_hi:[1,0], was always right.
//
// One workgroup = 8 waves: waves 0-3 ("checkers", one per SIMD) run the packed multiply in a loop and compare both
// halves with single-lane v_mul_f32 products; waves 4-7 ("partners", the second wave of each SIMD) run one of
//   0 nothing   1 v_mfma_f32_32x32x16_bf16 chains   2 single-lane VALU   3 ds_read_b128   4 global_load_dwordx4
//   5 global_store_dwordx4   6 v_pk_fma_f32   7 the checker loop itself
// Operand sources of the checker: 0 = both operands produced by VALU instructions; 1 = src1.lo comes from a global load
// and src1.hi from a v_mov (the situation of the failing epilogue: weights by global_load_dwordx3, one copied).
// Output per (form, source, partner): executions, wrong low halves by lane quarter, how many of them were exact zeros.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pk_opsel_probe.hip -o tools/pk_opsel_probe.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

struct Counters {
  unsigned long long execs;
  unsigned wrong_lo[4], wrong_hi[4], zero_lo[4];
};

template <int FORM>
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  f32x2 r;
  if (FORM == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));          // lo = a.x*b.y, hi = a.y*b.y
  if (FORM == 1) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));                        // lo = a.x*b.x, hi = a.y*b.y
  if (FORM == 2) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));        // lo = a.x*b.x, hi = a.y*b.x
  if (FORM == 3) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b));          // lo = a.y*b.x, hi = a.y*b.y
  return r;
}
__device__ __forceinline__ float mul1(float a, float b) {
  float r;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int FORM, int SRC>
__device__ void checker(Counters* c, int iters, const float* gsrc) {
  const int lane = threadIdx.x & 63, q = lane >> 4;
  f32x2 a = {1.0f + 0.001f * lane, 2.0f + 0.003f * lane};
  f32x2 b = {0.5f + 0.002f * lane, 0.25f + 0.004f * lane};
  unsigned wl = 0, wh = 0, zl = 0;
  for (int it = 0; it < iters; ++it) {
    if (SRC == 1) {
      // src1.lo from memory, src1.hi a VALU copy -- like v[64:66] <- global_load_dwordx3, v67 <- v_mov v64
      f32x4 w;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(gsrc + 4 * ((lane + it) & 1023)) : "memory");
      float hi;
      asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "v"(w.x));
      b = f32x2{w.z, hi};
    }
    const f32x2 r = pk_mul<FORM>(a, b);
    float elo, ehi;
    if (FORM == 0) { elo = mul1(a.x, b.y); ehi = mul1(a.y, b.y); }
    if (FORM == 1) { elo = mul1(a.x, b.x); ehi = mul1(a.y, b.y); }
    if (FORM == 2) { elo = mul1(a.x, b.x); ehi = mul1(a.y, b.x); }
    if (FORM == 3) { elo = mul1(a.y, b.x); ehi = mul1(a.y, b.y); }
    if (__float_as_uint(r.x) != __float_as_uint(elo)) { ++wl; if (r.x == 0.f) ++zl; }
    if (__float_as_uint(r.y) != __float_as_uint(ehi)) ++wh;
    a.x += 0.0625f; a.y -= 0.03125f;
    if (SRC == 0) { b.x += 0.015625f; b.y += 0.0078125f; }
  }
  if (wl) atomicAdd(&c->wrong_lo[q], wl);
  if (wh) atomicAdd(&c->wrong_hi[q], wh);
  if (zl) atomicAdd(&c->zero_lo[q], zl);
  if (lane == 0) atomicAdd(&c->execs, (unsigned long long)iters * 64ull);
}

template <int PARTNER, int FORM, int SRC>
__device__ void partner(Counters* c, int iters, float* sink, const float* gsrc, float* lds) {
  const int lane = threadIdx.x & 63;
  if (PARTNER == 0) return;
  if (PARTNER == 1) {
    f32x16 acc[4];
    for (int k = 0; k < 4; ++k)
      for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
    bf16x8 fa, fb;
    for (int k = 0; k < 8; ++k) { fa[k] = (__bf16)(0.5f + 0.01f * lane); fb[k] = (__bf16)(0.25f + 0.02f * k); }
    for (int it = 0; it < iters; ++it)
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[k], 0, 0, 0);
    float s = 0.f;
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][15];
    if (s == 12345.678f) sink[lane] = s;
  }
  if (PARTNER == 2) {
    float x = 1.0f + lane, y = 0.999f;
    for (int it = 0; it < iters * 8; ++it) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y)); }
    if (x == 12345.678f) sink[lane] = x;
  }
  if (PARTNER == 3) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters * 2; ++it) {
      const f32x4 v = *reinterpret_cast<const volatile f32x4*>(lds + 4 * ((lane + it) & 1023));
      s += v;
    }
    if (s.x == 12345.678f) sink[lane] = s.x;
  }
  if (PARTNER == 4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
      f32x4 v;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(gsrc + 4 * ((lane * 7 + it * 64) & 4095)) : "memory");
      s += v;
    }
    if (s.x == 12345.678f) sink[lane] = s.x;
  }
  if (PARTNER == 5) {
    float* o = sink + 4096 + (size_t)(blockIdx.x * 4 + ((threadIdx.x >> 6) & 3)) * 4096;
    for (int it = 0; it < iters; ++it)
      *reinterpret_cast<f32x4*>(o + 4 * ((lane + it * 64) & 1023)) = f32x4{(float)it, 1.f, 2.f, 3.f};
  }
  if (PARTNER == 6) {
    f32x2 x = {1.0f + lane, 2.0f}, y = {0.999f, 1.001f};
    for (int it = 0; it < iters * 8; ++it) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (x.x == 12345.678f) sink[lane] = x.x;
  }
  if (PARTNER == 7) checker<FORM, SRC>(c, iters, gsrc);
}

template <int FORM, int SRC, int PARTNER>
__global__ __launch_bounds__(512) void probe_kernel(Counters* c, int iters, float* sink, const float* gsrc) {
  __shared__ float lds[4096 + 64];
  for (int i = threadIdx.x; i < 4096 + 64; i += 512) lds[i] = 0.001f * i;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  if (wave < 4) checker<FORM, SRC>(c, iters, gsrc);
  else partner<PARTNER, FORM, SRC>(c, iters, sink, gsrc, lds);
}

template <int FORM, int SRC, int PARTNER>
static void run(Counters* dC, int iters, float* sink, const float* gsrc, int blocks) {
  CK(hipMemset(dC, 0, sizeof(Counters)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe_kernel<FORM, SRC, PARTNER>), dim3(blocks), dim3(512), 0, 0, dC, iters, sink, gsrc);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  Counters h;
  CK(hipMemcpy(&h, dC, sizeof(h), hipMemcpyDeviceToHost));
  static const char* forms[] = {"op_sel:[0,1]", "no op_sel", "op_sel_hi:[1,0]", "op_sel:[1,0]"};
  static const char* partners[] = {"none", "mfma bf16", "valu fma", "ds_read_b128", "global_load", "global_store", "pk_fma", "checker"};
  printf("%-16s src %d  partner %-13s %6.2f ms  execs %.3g  wrong lo by lane/16 [%u %u %u %u] (exact zeros [%u %u %u %u])  wrong hi [%u %u %u %u]\n",
         forms[FORM], SRC, partners[PARTNER], ms, (double)h.execs, h.wrong_lo[0], h.wrong_lo[1], h.wrong_lo[2], h.wrong_lo[3],
         h.zero_lo[0], h.zero_lo[1], h.zero_lo[2], h.zero_lo[3], h.wrong_hi[0], h.wrong_hi[1], h.wrong_hi[2], h.wrong_hi[3]);
  fflush(stdout);
}

template <int FORM, int SRC>
static void run_partners(Counters* dC, int iters, float* sink, const float* gsrc, int blocks) {
  run<FORM, SRC, 0>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 1>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 2>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 3>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 4>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 5>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 6>(dC, iters, sink, gsrc, blocks);
  run<FORM, SRC, 7>(dC, iters, sink, gsrc, blocks);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int blocks = argc > 2 ? atoi(argv[2]) : 1024;
  Counters* dC; float *sink, *gsrc;
  CK(hipMalloc(&dC, sizeof(Counters)));
  CK(hipMalloc(&sink, (4096 + (size_t)blocks * 4 * 4096) * 4));
  CK(hipMalloc(&gsrc, 8192 * 4));
  std::vector<float> h(8192);
  for (int i = 0; i < 8192; ++i) h[i] = 0.1f + 0.0001f * i;
  CK(hipMemcpy(gsrc, h.data(), 8192 * 4, hipMemcpyHostToDevice));
  printf("pk_opsel_probe: %d iterations per wave, %d workgroups of 8 waves (4 checkers + 4 partners)\n", iters, blocks);
  run_partners<0, 0>(dC, iters, sink, gsrc, blocks);
  run_partners<0, 1>(dC, iters, sink, gsrc, blocks);
  run_partners<1, 1>(dC, iters, sink, gsrc, blocks);
  run_partners<2, 1>(dC, iters, sink, gsrc, blocks);
  run_partners<3, 1>(dC, iters, sink, gsrc, blocks);
  return 0;
}
