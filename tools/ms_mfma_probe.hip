// ms_mfma_probe.hip -- would v_mfma_f32_4x4x1_16b_f32 pay as the accumulation step of the MeanShift pair loop?
// Per point and lane pair the loop is 3 v_pk_fma (exponent) + 2 v_exp + [v_pk_add + 3 v_pk_fma] (accumulate w, w a).
// Variant M replaces the bracket by two 4x4x1 MFMAs (A = w of the lane's seed, B = (x, y, z, 1)[lane & 3]).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ms_mfma_probe.hip -o tools/ms_mfma_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: VALU accumulate, 1: MFMA accumulate, 2: exponent part only, 3: MFMA part only
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float4 s[512];
  for (int i = threadIdx.x; i < 512; i += 256) s[i] = make_float4(i * 1e-3f, 0.5f, 0.25f, -1.f);
  __syncthreads();
  const f2 p2x = {threadIdx.x * 1e-3f, 0.3f}, p2y = {0.1f, 0.2f}, p2z = {0.05f, 0.07f};
  f2 Aw = {0, 0}, Ax = {0, 0}, Ay = {0, 0}, Az = {0, 0};
  f4 D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
  const float* sf = reinterpret_cast<const float*>(s);
  for (int i = 0; i < iters; ++i) {
#pragma unroll 4
    for (int q = 0; q < 512; ++q) {
      const float4 a = s[q];
      f2 w = {1.f, 1.f};
      if (MODE != 3) {
        const f2 e = __builtin_elementwise_fma(p2z, f2{a.z, a.z}, __builtin_elementwise_fma(p2y, f2{a.y, a.y}, __builtin_elementwise_fma(p2x, f2{a.x, a.x}, f2{a.w, a.w})));
        w = f2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
      }
      if (MODE == 0) {
        Aw += w;
        Ax = __builtin_elementwise_fma(w, f2{a.x, a.x}, Ax);
        Ay = __builtin_elementwise_fma(w, f2{a.y, a.y}, Ay);
        Az = __builtin_elementwise_fma(w, f2{a.z, a.z}, Az);
      } else if (MODE == 1 || MODE == 3) {
        const float b = sf[q * 4 + (threadIdx.x & 3)];
        D0 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.x, b, D0, 0, 0, 0);
        D1 = __builtin_amdgcn_mfma_f32_4x4x1f32(w.y, b, D1, 0, 0, 0);
      } else {
        Aw += w;
      }
    }
  }
  const float r = Aw.x + Aw.y + Ax.x + Ax.y + Ay.x + Ay.y + Az.x + Az.y + D0.x + D0.y + D0.z + D0.w + D1.x + D1.y + D1.z + D1.w;
  if (r == 12345.f) out[threadIdx.x] = r;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[4] = {"VALU accumulate (shipping loop)", "MFMA 4x4x1 accumulate", "exponent + exp only", "MFMA accumulate only"};
  for (int wgs : {256, 1024}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        const int iters = 40;
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, out, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, out, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(256), 0, 0, out, iters);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(wgs), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double pts = 40.0 * 512;       // points per wave
      printf("%4d workgroups (%d wave/SIMD)  %-32s %7.3f ms  %6.1f cycles per point per wave-slot @2.4GHz\n", wgs, wgs / 256, names[mode],
             best, best * 1e-3 * 2.4e9 / pts / (wgs / 256));
    }
  }
  return 0;
}
