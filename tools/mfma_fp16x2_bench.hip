// mfma_fp16x2_bench.hip -- would TWO fp16 pieces per fp32 operand (3 partial products on the fp16 matrix pipe) do what
// the THREE bf16 pieces (6 partial products) of csrc/sa_mlp_split.hip / split_gemm.hip do, at half the MFMA time?
//   x * s = h + l + e,  h = RN_fp16(x s), l = RN_fp16(x s - h), |e| <= 2^-22 |x s|   (s = a power of two that puts the
//   operand range inside fp16's; an fp16 x fp16 product is exact in fp32: 11 + 11 significant bits)
//   w.x ~ (wh.xh + wh.xl + wl.xh) / (s_w s_x): the dropped wl.xl is <= 2^-22 of the product.
// fp32's own unit roundoff is 2^-24; the fp32 FMA chain of K terms the reference runs is ~sqrt(K) 2^-24 away from exact.
// Part A: max / rms error of C = W (32 x K) . X (K x 32) against fp64 over many tiles, K = 512 and 1536, for the fp32
//         MFMA chain, bf16 x 3 (6 terms), fp16 x 2 (3 and 4 terms; RN and truncating splits; operand range scaled to
//         2^14 and -- a loose bound -- to 2^2 only).
// Part B: bare matrix-pipe time of the two schemes (register-resident fragments, 4 independent accumulators per wave).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_fp16x2_bench.hip -o tools/mfma_fp16x2_bench.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef unsigned short u16;

static inline u16 bf16_rne(float x) { unsigned u = __builtin_bit_cast(unsigned, x); u += 0x7fffu + ((u >> 16) & 1u); return (u16)(u >> 16); }
static inline float bf16_f(u16 h) { return __builtin_bit_cast(float, (unsigned)h << 16); }
static inline u16 f16_bits(_Float16 h) { return __builtin_bit_cast(u16, h); }
static inline _Float16 f16_trunc(float x) {       // round toward zero
  _Float16 h = (_Float16)x;
  if (fabsf((float)h) > fabsf(x)) { u16 b = f16_bits(h); b -= 1; h = __builtin_bit_cast(_Float16, b); }
  return h;
}

// pieces [p][row][k] as 16-bit words
template <int MODE>   // 0 fp32 mfma, 1 bf16x3 6 terms, 2 fp16x2 3 terms, 3 fp16x2 4 terms
__global__ void acc_kernel(const u16* __restrict__ Wp, const u16* __restrict__ Xp, const float* __restrict__ Wf,
                           const float* __restrict__ Xf, int K, float inv_scale, float* __restrict__ C) {
  const int tile = blockIdx.x;
  const int lane = threadIdx.x & 63, r = lane & 31, half = lane >> 5;
  const size_t plane = (size_t)gridDim.x * 32 * K, base = ((size_t)tile * 32 + r) * K;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (MODE == 0) {
    for (int k = 0; k < K; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Wf[base + k + half], Xf[base + k + half], acc, 0, 0, 0);
  } else if (MODE == 1) {
    for (int k = 0; k < K; k += 16) {
      bf16x8 a[3], b[3];
      for (int p = 0; p < 3; ++p) {
        a[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Wp + p * plane + base + k + 8 * half));
        b[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xp + p * plane + base + k + 8 * half));
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 16) {
      f16x8 a[2], b[2];
      for (int p = 0; p < 2; ++p) {
        a[p] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Wp + p * plane + base + k + 8 * half));
        b[p] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(Xp + p * plane + base + k + 8 * half));
      }
      if (MODE == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) acc[i] *= inv_scale;
  }
  for (int i = 0; i < 16; ++i) C[(size_t)tile * 1024 + ((i & 3) + 8 * (i >> 2) + 4 * half) * 32 + r] = acc[i];
}

template <int MODE>   // 1: 6 bf16 MFMAs per slab, 2: 3 fp16 MFMAs per slab; 4 accumulators (2 x 2 tiles) per wave
__global__ __launch_bounds__(256) void pipe_kernel(float* out, int slabs) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int lane = threadIdx.x & 63;
  bf16x8 ab[3], bb[3]; f16x8 ah[2], bh[2];
  for (int p = 0; p < 3; ++p) for (int i = 0; i < 8; ++i) { ab[p][i] = (__bf16)(0.01f * (lane + i + p)); bb[p][i] = (__bf16)(0.02f * (lane - i + p)); }
  for (int p = 0; p < 2; ++p) for (int i = 0; i < 8; ++i) { ah[p][i] = (_Float16)(0.01f * (lane + i + p)); bh[p][i] = (_Float16)(0.02f * (lane - i + p)); }
  for (int s = 0; s < slabs; ++s) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (MODE == 1) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[0], bb[2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[2], bb[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[1], bb[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[0], bb[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[1], bb[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[0], bb[0], acc[t], 0, 0, 0);
      } else {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][15];
  if (s == 1234.5f) out[threadIdx.x] = s;
}

int main() {
  const int TILES = 256;
  for (int K : {512, 1536}) {
    const size_t n = (size_t)TILES * 32 * K;
    std::vector<float> Wf(n), Xf(n);
    srand(K);
    auto rnd = []() { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return (float)(u - 6.0); };
    float wmax = 0, xmax = 0;
    for (auto& v : Wf) { v = rnd() * 0.06f; wmax = fmaxf(wmax, fabsf(v)); }
    for (auto& v : Xf) { v = fmaxf(rnd() * 3.f + 0.5f, 0.f); xmax = fmaxf(xmax, fabsf(v)); }       // post-ReLU-like activations
    std::vector<double> ref((size_t)TILES * 1024);
    double scale = 0;
    for (int t = 0; t < TILES; ++t)
      for (int m = 0; m < 32; ++m)
        for (int nn = 0; nn < 32; ++nn) {
          double s = 0;
          const float* w = &Wf[((size_t)t * 32 + m) * K]; const float* x = &Xf[((size_t)t * 32 + nn) * K];
          for (int k = 0; k < K; ++k) s += (double)w[k] * (double)x[k];
          ref[(size_t)t * 1024 + m * 32 + nn] = s;
          scale = fmax(scale, fabs(s));
        }
    u16 *dWp, *dXp; float *dWf, *dXf, *dC;
    hipMalloc(&dWp, 3 * n * 2); hipMalloc(&dXp, 3 * n * 2); hipMalloc(&dWf, n * 4); hipMalloc(&dXf, n * 4); hipMalloc(&dC, TILES * 1024 * 4);
    hipMemcpy(dWf, Wf.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dXf, Xf.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> C((size_t)TILES * 1024);
    auto report = [&](const char* name) {
      hipDeviceSynchronize();
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
      double mx = 0, rms = 0;
      for (size_t i = 0; i < C.size(); ++i) { const double e = fabs(C[i] - ref[i]); mx = fmax(mx, e); rms += e * e; }
      printf("K = %4d  %-52s max|err|/scale = %.3e   rms/scale = %.3e\n", K, name, mx / scale, sqrt(rms / C.size()) / scale);
    };
    std::vector<u16> Wp(3 * n), Xp(3 * n);
    acc_kernel<0><<<TILES, 64>>>(dWp, dXp, dWf, dXf, K, 1.f, dC); report("fp32 MFMA chain (v_mfma_f32_32x32x2_f32)");
    for (size_t i = 0; i < n; ++i) {
      u16 a = bf16_rne(Wf[i]); float r1 = Wf[i] - bf16_f(a); u16 b = bf16_rne(r1); u16 c = bf16_rne(r1 - bf16_f(b));
      Wp[i] = a; Wp[n + i] = b; Wp[2 * n + i] = c;
      // activations by truncation like the kernels
      unsigned u = __builtin_bit_cast(unsigned, Xf[i]); u16 h = (u16)(u >> 16); float q1 = Xf[i] - bf16_f(h);
      unsigned u2 = __builtin_bit_cast(unsigned, q1); u16 m = (u16)(u2 >> 16); float q2 = q1 - bf16_f(m);
      Xp[i] = h; Xp[n + i] = m; Xp[2 * n + i] = (u16)(__builtin_bit_cast(unsigned, q2) >> 16);
    }
    hipMemcpy(dWp, Wp.data(), 3 * n * 2, hipMemcpyHostToDevice); hipMemcpy(dXp, Xp.data(), 3 * n * 2, hipMemcpyHostToDevice);
    acc_kernel<1><<<TILES, 64>>>(dWp, dXp, dWf, dXf, K, 1.f, dC); report("bf16 x 3, 6 terms (the shipping split)");
    // fp16 x 2: operands scaled to [.., 2^top]
    for (int variant = 0; variant < 4; ++variant) {
      const bool trunc = variant == 1;
      const int top = variant == 2 ? 2 : 14;           // variant 2: a bound 2^12 too loose leaves the values near 2^2
      const float sw = exp2f((float)(top - (int)ceilf(log2f(wmax)))), sx = exp2f((float)(top - (int)ceilf(log2f(xmax))));
      for (size_t i = 0; i < n; ++i) {
        const float w = Wf[i] * sw, x = Xf[i] * sx;
        _Float16 wh = (_Float16)w, xh = trunc ? f16_trunc(x) : (_Float16)x;       // weights always RN (host)
        _Float16 wl = (_Float16)(w - (float)wh), xl = trunc ? f16_trunc(x - (float)xh) : (_Float16)(x - (float)xh);
        Wp[i] = f16_bits(wh); Wp[n + i] = f16_bits(wl); Xp[i] = f16_bits(xh); Xp[n + i] = f16_bits(xl);
      }
      hipMemcpy(dWp, Wp.data(), 2 * n * 2, hipMemcpyHostToDevice); hipMemcpy(dXp, Xp.data(), 2 * n * 2, hipMemcpyHostToDevice);
      const char* names[4] = {"fp16 x 2, 3 terms, RN split, range -> 2^14", "fp16 x 2, 3 terms, activations TRUNCATED, -> 2^14",
                              "fp16 x 2, 3 terms, RN split, range -> 2^2 (loose bound)", "fp16 x 2, 4 terms, RN split, range -> 2^14"};
      if (variant == 3) acc_kernel<3><<<TILES, 64>>>(dWp, dXp, dWf, dXf, K, 1.f / (sw * sx), dC);
      else acc_kernel<2><<<TILES, 64>>>(dWp, dXp, dWf, dXf, K, 1.f / (sw * sx), dC);
      report(names[variant]);
    }
    {
      double mx = 0, rms = 0;
      for (int t = 0; t < TILES; ++t)
        for (int m = 0; m < 32; ++m)
          for (int nn = 0; nn < 32; ++nn) {
            float s = 0;
            const float* w = &Wf[((size_t)t * 32 + m) * K]; const float* x = &Xf[((size_t)t * 32 + nn) * K];
            for (int k = 0; k < K; ++k) s = fmaf(w[k], x[k], s);
            const double e = fabs(s - ref[(size_t)t * 1024 + m * 32 + nn]); mx = fmax(mx, e); rms += e * e;
          }
      printf("K = %4d  %-52s max|err|/scale = %.3e   rms/scale = %.3e\n", K, "host fp32 FMA chain in k order (what cuDNN / MIOpen-class code does)",
             mx / scale, sqrt(rms / (TILES * 1024.0)) / scale);
    }
    hipFree(dWp); hipFree(dXp); hipFree(dWf); hipFree(dXf); hipFree(dC);
  }
  // ---- part B
  float* out; hipMalloc(&out, 4096);
  const int slabs = 4096, blocks = 256 * 8;
  for (int mode : {1, 2}) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 1) pipe_kernel<1><<<blocks, 256>>>(out, slabs); else pipe_kernel<2><<<blocks, 256>>>(out, slabs);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = 2.0 * 32 * 32 * 16 * 4.0 * slabs * blocks * 4;       // fp32-equivalent flops (one multiply-add per (m, n, k))
    printf("matrix pipe only: %-28s %7.3f ms   %7.1f TFLOP/s fp32-equivalent\n", mode == 1 ? "bf16 x 3, 6 MFMAs per slab" : "fp16 x 2, 3 MFMAs per slab",
           ms, fl / (ms * 1e-3) / 1e12);
  }
  return 0;
}
