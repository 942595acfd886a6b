// mfma_split_bench.hip -- can the fp32 contraction of the fused SA / FP MLP kernels (csrc/sa_mlp.hip) run on the bf16
// matrix pipe without giving up fp32 accuracy?  v_mfma_f32_32x32x2_f32 is 1/16 of the bf16 MFMA rate.  An fp32 value
// is the exact sum of three bf16 pieces x = x1 + x2 + x3 (8 + 8 + 8 mantissa bits), a bf16 x bf16 product is exact in
// fp32, so x.w = sum of partial products x_i.w_j accumulated in fp32: 9 terms give every bit, 6 terms (i + j <= 4) drop
// only contributions below 2^-24 of the product (fp32's own rounding), 3 terms (i + j <= 3) stop at 2^-16.
// Part A: accuracy of C = W (M x K) . X (K x N), K = 512, against fp64, for fp32 MFMA and the 3 / 6 / 9-term splits.
// Part B: throughput of the chain's inner loop (A fragments from global / L2, B fragments from LDS, NT row tiles x 2
//         column tiles per wave) in fp32-equivalent TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_split_bench.hip -o tools/mfma_split_bench.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef unsigned short u16;

__host__ __device__ inline u16 bf16_rne(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (u16)(u >> 16);
}
__host__ __device__ inline float bf16_f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
__host__ __device__ inline void split3(float x, u16& a, u16& b, u16& c) {
  a = bf16_rne(x);
  const float r1 = x - bf16_f(a);
  b = bf16_rne(r1);
  const float r2 = r1 - bf16_f(b);
  c = bf16_rne(r2);
}

// ---------------------------------------------------------------- part A: one wave, one 32 x 32 tile, K = 512
// Wp[piece][m][k], Xp[piece][n][k] bf16; Wf[m][k], Xf[n][k] fp32
template <int TERMS>
__global__ void acc_kernel(const u16* __restrict__ Wp, const u16* __restrict__ Xp, const float* __restrict__ Wf,
                           const float* __restrict__ Xf, int K, float* __restrict__ C) {
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (TERMS == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wf[r * K + k + half], Xf[r * K + k + half], acc, 0, 0, 0);
  } else {
    for (int k = 0; k < K; k += 16) {
      bf16x8 a[3], b[3];
      for (int p = 0; p < 3; ++p) {
        a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wp + ((size_t)p * 32 + r) * K + k + 8 * half));
        b[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xp + ((size_t)p * 32 + r) * K + k + 8 * half));
      }
      // smallest terms first
      if (TERMS >= 9) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[1], acc, 0, 0, 0);
      }
      if (TERMS >= 6) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * half) * 32 + r] = acc[i];
}

// ---------------------------------------------------------------- part B: inner-loop throughput
// MODE 0: fp32 MFMA (float2 A per pair of k-steps from global, 4 LDS b32 reads per pair) -- the shipping loop
// MODE 6 / 9: split bf16: per 16-k slab 3 x 16-byte A loads per row tile from global, 3 x 16-byte LDS reads per column
//             tile, TERMS MFMAs per (row tile, column tile)
template <int MODE, int NT>
__global__ __launch_bounds__(256, 2) void loop_kernel(const void* __restrict__ Wv, float* __restrict__ out, int K, int reps) {
  extern __shared__ char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, col = lane & 31;
  f32x16 acc[NT][2];
  for (int t = 0; t < NT; ++t) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
  if (MODE == 0) {
    float* H = reinterpret_cast<float*>(smem);
    for (int i = tid; i < K * 64; i += 256) H[i] = 1e-3f * (i & 255);
    __syncthreads();
    const float2* W = reinterpret_cast<const float2*>(Wv) + (size_t)wave * 64 + lane;
    const size_t pstride = (size_t)4 * NT * 64, tstride = 4 * 64;
    const int pairs = K / 4;
    for (int rep = 0; rep < reps; ++rep) {
      float2 ring[8][NT];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[u][t] = W[(size_t)u * pstride + t * tstride];
      for (int p0 = 0; p0 < pairs; p0 += 8) {
        const float* r0 = H + (size_t)p0 * 4 * 64 + half * 64;
        float bb[2][4];
        bb[0][0] = r0[col]; bb[0][1] = r0[col + 32]; bb[0][2] = r0[128 + col]; bb[0][3] = r0[128 + col + 32];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u + 1 < 8) {
            const float* r1 = r0 + (u + 1) * 4 * 64;
            bb[(u + 1) & 1][0] = r1[col]; bb[(u + 1) & 1][1] = r1[col + 32];
            bb[(u + 1) & 1][2] = r1[128 + col]; bb[(u + 1) & 1][3] = r1[128 + col + 32];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].x, bb[u & 1][0], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].x, bb[u & 1][1], acc[t][1], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].y, bb[u & 1][2], acc[t][0], 0, 0, 0);
            acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].y, bb[u & 1][3], acc[t][1], 0, 0, 0);
          }
          const int pn = min(p0 + 8 + u, pairs - 1);
#pragma unroll
          for (int t = 0; t < NT; ++t) ring[u][t] = W[(size_t)pn * pstride + t * tstride];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    // LDS: 3 pieces x 64 columns x (K bf16 + 8 pad): row stride 2K + 16 bytes (16 x odd: conflict-free b128 reads)
    const int rs = 2 * K + 16;
    for (int i = tid; i < 3 * 64 * rs / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0x3c003c00u + (i & 0xff);
    __syncthreads();
    // A: [k16][row tile of the workgroup (4 * NT)][piece][64 lanes] x 16 bytes; ring of D slabs, statically indexed
    constexpr int D = 2;
    const uint4* W = reinterpret_cast<const uint4*>(Wv) + lane;
    const size_t kstride = (size_t)4 * NT * 3 * 64;
    const int slabs = K / 16;
    const char* bbase = smem + (size_t)col * rs + half * 16;
    for (int rep = 0; rep < reps; ++rep) {
      uint4 ring[D][NT][3];
#pragma unroll
      for (int u = 0; u < D; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int p = 0; p < 3; ++p) ring[u][t][p] = W[(size_t)u * kstride + ((size_t)(wave + 4 * t) * 3 + p) * 64];
      bf16x8 b[2][2][3];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[0][c][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(bbase + ((size_t)p * 64 + c * 32) * rs));
      // slab s is multiplied from ring[s & 1]; the refill of the OTHER slot (consumed by the previous slab) is issued at
      // the start of the slab, so at the next slab's start it is the only load outstanding: vmcnt(0) is then exact, and
      // the prefetch distance is one slab of MFMAs (the compiler puts vmcnt(0) at a loop head whatever is in flight)
      for (int s0 = 0; s0 < slabs; s0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
          const int s = s0 + u;
          if (s > 0) {
            const int sn = min(s + 1, slabs - 1);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int p = 0; p < 3; ++p)
                ring[u ^ 1][t][p] = W[(size_t)sn * kstride + ((size_t)(wave + 4 * t) * 3 + p) * 64];
          }
          const int sb = min(s + 1, slabs - 1);
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 3; ++p)
              b[(u + 1) & 1][c][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(
                  bbase + ((size_t)p * 64 + c * 32) * rs + (size_t)sb * 32));
          __builtin_amdgcn_sched_barrier(0);
#define MM(PA, PB)                                                                                                       \
  _Pragma("unroll") for (int t = 0; t < NT; ++t) _Pragma("unroll") for (int c = 0; c < 2; ++c)                           \
      acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ring[u][t][PA]), b[u & 1][c][PB], acc[t][c], 0, 0, 0)
          if (MODE >= 9) { MM(2, 2); MM(1, 2); MM(2, 1); }
          MM(0, 2); MM(2, 0); MM(1, 1);
          MM(0, 1); MM(1, 0); MM(0, 0);
#undef MM
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  float s = 0;
  for (int t = 0; t < NT; ++t) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[t][c][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NT>
void time_loop(const char* name, int K, const void* W, float* out) {
  const int reps = 40, wgs = 256 * 2 * 8;
  const size_t lds = MODE == 0 ? (size_t)K * 64 * 4 : (size_t)3 * 64 * (2 * K + 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(loop_kernel<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  loop_kernel<MODE, NT><<<wgs, 256, lds>>>(W, out, K, 2);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  loop_kernel<MODE, NT><<<wgs, 256, lds>>>(W, out, K, reps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * (4.0 * NT * 32) * 64 * K * (double)reps * wgs;       // fp32-equivalent
  printf("%-34s K=%3d NT=%d lds=%6zu B: %8.3f ms  %7.1f TFLOP/s fp32-equivalent (%s)\n", name, K, NT, lds, ms, flops / ms / 1e9,
         hipGetErrorString(hipGetLastError()));
}

int main() {
  // ---- part A
  const int K = 512;
  std::vector<float> Wf(32 * K), Xf(32 * K);
  srand(1);
  auto rnd = []() { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return (float)(u - 6.0); };
  for (auto& v : Wf) v = rnd() * 0.06f;
  for (auto& v : Xf) v = fmaxf(rnd() * 3.f + 0.5f, 0.f);       // post-ReLU-like activations
  std::vector<u16> Wp(3 * 32 * K), Xp(3 * 32 * K);
  for (int i = 0; i < 32 * K; ++i) {
    split3(Wf[i], Wp[i], Wp[32 * K + i], Wp[2 * 32 * K + i]);
    split3(Xf[i], Xp[i], Xp[32 * K + i], Xp[2 * 32 * K + i]);
  }
  std::vector<double> ref(32 * 32);
  double scale = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)Wf[m * K + k] * (double)Xf[n * K + k];
      ref[m * 32 + n] = s;
      scale = fmax(scale, fabs(s));
    }
  u16 *dWp, *dXp; float *dWf, *dXf, *dC;
  hipMalloc(&dWp, Wp.size() * 2); hipMalloc(&dXp, Xp.size() * 2);
  hipMalloc(&dWf, Wf.size() * 4); hipMalloc(&dXf, Xf.size() * 4); hipMalloc(&dC, 32 * 32 * 4);
  hipMemcpy(dWp, Wp.data(), Wp.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dXp, Xp.data(), Xp.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dWf, Wf.data(), Wf.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dXf, Xf.data(), Xf.size() * 4, hipMemcpyHostToDevice);
  std::vector<float> C(32 * 32);
  auto report = [&](const char* name) {
    hipDeviceSynchronize();
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double mx = 0, rms = 0;
    for (int i = 0; i < 32 * 32; ++i) { const double e = fabs(C[i] - ref[i]); mx = fmax(mx, e); rms += e * e; }
    printf("accuracy %-22s max|err|/scale = %.3e   rms/scale = %.3e   (K = %d, scale %.3g)\n", name, mx / scale,
           sqrt(rms / 1024) / scale, K, scale);
  };
  acc_kernel<0><<<1, 64>>>(dWp, dXp, dWf, dXf, K, dC); report("fp32 mfma 32x32x2");
  acc_kernel<3><<<1, 64>>>(dWp, dXp, dWf, dXf, K, dC); report("bf16 split, 3 terms");
  acc_kernel<6><<<1, 64>>>(dWp, dXp, dWf, dXf, K, dC); report("bf16 split, 6 terms");
  acc_kernel<9><<<1, 64>>>(dWp, dXp, dWf, dXf, K, dC); report("bf16 split, 9 terms");
  // host fp32 fma chain for comparison
  {
    double mx = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        float s = 0;
        for (int k = 0; k < K; ++k) s = fmaf(Wf[m * K + k], Xf[n * K + k], s);
        mx = fmax(mx, fabs(s - ref[m * 32 + n]));
      }
    printf("accuracy %-22s max|err|/scale = %.3e\n", "host fp32 fma chain", mx / scale);
  }
  // ---- part B
  void* W; float* out;
  hipMalloc(&W, 64 << 20); hipMemset(W, 0x3c, 64 << 20);
  hipMalloc(&out, 256 * 2 * 8 * 256 * 4);
  for (int Kb : {128, 256}) {
    time_loop<0, 1>("fp32 mfma (shipping loop shape)", Kb, W, out);
    time_loop<0, 2>("fp32 mfma (shipping loop shape)", Kb, W, out);
    time_loop<6, 1>("bf16 split 6 terms", Kb, W, out);
    time_loop<6, 2>("bf16 split 6 terms", Kb, W, out);
    time_loop<9, 1>("bf16 split 9 terms", Kb, W, out);
    time_loop<9, 2>("bf16 split 9 terms", Kb, W, out);
  }
  return 0;
}
