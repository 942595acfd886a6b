// microbench.hip -- VALU / LDS-permute issue rates on gfx950, to price the VALU-bound kernels
// (FPS, MeanShift, ball_query, three_nn).  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb
// Each kernel runs ITER iterations of 16 independent instances of one instruction per wave;
// 4 waves per SIMD (1024 threads/block, 1 block per CU x 256 CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, float seed) {
  float a[16];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[16];
  for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = f2{a[i], a[i] * 0.5f}; }
  float b = seed * 0.999f, c = seed * 1e-3f;
  f2 pb = {b, b}, pc = {c, c};
  for (int it = 0; it < ITER; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define CMP64(i) asm volatile("v_cmp_gt_i64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(*(long long*)&p[i]), "v"(*(long long*)&pb), "v"(a[i]), "v"(b) : "vcc");
#define CMP32(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define BPERM(i) asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a[i]) : "v"(threadIdx.x * 4 ^ 128));
#define DPP(i) asm volatile("s_nop 1\n v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define MADU64(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(b) : "vcc");
    if (KIND == 0) { REP16(FMA) }
    if (KIND == 1) { REP16(PKFMA) }
    if (KIND == 2) { REP16(EXP) }
    if (KIND == 3) { REP16(SQRT) }
    if (KIND == 4) { REP16(MAXF) }
    if (KIND == 5) { REP16(CMP64) }
    if (KIND == 6) { REP16(CMP32) }
    if (KIND == 7) { REP16(BPERM) }
    if (KIND == 8) { REP16(DPP) }
    if (KIND == 9) { REP16(PKADD) }
    if (KIND == 10) { REP16(PKMUL) }
    if (KIND == 11) { REP16(RCP) }
    if (KIND == 12) { REP16(SUB) }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND>
void run(const char* name, float* d, int insts_per_rep) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<256, 1024>>>(d, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<256, 1024>>>(d, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per SIMD: 4 waves x ITER x 16 x insts instructions
  double wave_insts = 4.0 * ITER * 16 * insts_per_rep;
  double ns_per = ms * 1e6 / wave_insts;
  printf("%-14s %8.3f ms  %6.3f ns per wave-instruction per SIMD (= %5.2f cycles @2.4GHz)\n", name, ms, ns_per, ns_per * 2.4);
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  run<0>("v_fma_f32", d, 1);
  run<1>("v_pk_fma_f32", d, 1);
  run<9>("v_pk_add_f32", d, 1);
  run<10>("v_pk_mul_f32", d, 1);
  run<12>("v_sub_f32", d, 1);
  run<2>("v_exp_f32", d, 1);
  run<3>("v_sqrt_f32", d, 1);
  run<11>("v_rcp_f32", d, 1);
  run<4>("v_max_f32", d, 1);
  run<5>("cmp_i64+cnd", d, 2);
  run<6>("cmp_f32+cnd", d, 2);
  run<7>("ds_bpermute", d, 1);
  run<8>("max_dpp+nop", d, 1);
  return 0;
}
