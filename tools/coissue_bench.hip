// coissue_bench.hip -- can a VALU wave and an MFMA wave on the same SIMD make progress at the same time on gfx950?
// Two synthetic kernels, one wave per SIMD each (256 workgroups x 256 threads), launched on two streams:
//   M: fp32 MFMA only (v_mfma_f32_32x32x2_f32, four independent accumulators), V: fp32 VALU only (fma + exp),
//   B: bf16 MFMA only (v_mfma_f32_32x32x16_bf16).
// Build: hipcc --offload-arch=gfx950 -O3 tools/coissue_bench.hip -o tools/coissue_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_kernel(float* out, int iters) {
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  const float a = threadIdx.x * 1e-3f, b = 1.0f;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
  }
  const v16f s = c0 + c1 + c2 + c3;
  if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_bf16_kernel(float* out, int iters) {
  v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  bf16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(threadIdx.x * 1e-3f); b[k] = (__bf16)1.0f; }
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
  }
  const v16f s = c0 + c1 + c2 + c3;
  if (s[0] == 12345.f) out[threadIdx.x] = s[1];
}

template <int PRIO>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters) {
  if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
  float a = threadIdx.x * 1e-3f, b = 0.f, c = 0.f, d = 0.f, w = 0.f;
  for (int i = 0; i < iters; ++i) {
    const float e = fmaf(a, 0.5f, fmaf(b, 0.25f, fmaf(c, 0.125f, -1.f)));
    const float ww = __builtin_amdgcn_exp2f(e);
    w += ww; b = fmaf(ww, 0.5f, b); c = fmaf(ww, 0.25f, c); d = fmaf(ww, 0.125f, d);
    a = a * 0.999f + 1e-4f;
  }
  if (w == 12345.f) out[threadIdx.x] = b + c + d;
}

static float run(int which, hipStream_t s0, hipStream_t s1, float* out, int wgs, int im, int iv) {
  // 7: V(prio 3) || M, V enqueued first; 8: V || M, V enqueued first; 9: M || V(prio 3), M first
  // which: 0 M alone, 1 V alone, 2 M||V, 3 M||M, 4 V||V, 5 B alone (bf16 MFMA), 6 B||V
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipStreamWaitEvent(s0, e0, 0); hipStreamWaitEvent(s1, e0, 0);
    if (which == 7) hipLaunchKernelGGL(valu_kernel<3>, dim3(wgs), dim3(256), 0, s1, out, iv);
    if (which == 8) hipLaunchKernelGGL(valu_kernel<0>, dim3(wgs), dim3(256), 0, s1, out, iv);
    if (which == 0 || which == 2 || which == 3 || which >= 7) hipLaunchKernelGGL(mfma_kernel, dim3(wgs), dim3(256), 0, s0, out, im);
    if (which == 1 || which == 4) hipLaunchKernelGGL(valu_kernel<0>, dim3(wgs), dim3(256), 0, s0, out, iv);
    if (which == 5 || which == 6) hipLaunchKernelGGL(mfma_bf16_kernel, dim3(wgs), dim3(256), 0, s0, out, im * 2);
    if (which == 2 || which == 4 || which == 6) hipLaunchKernelGGL(valu_kernel<0>, dim3(wgs), dim3(256), 0, s1, out, iv);
    if (which == 9) hipLaunchKernelGGL(valu_kernel<3>, dim3(wgs), dim3(256), 0, s1, out, iv);
    if (which == 3) hipLaunchKernelGGL(mfma_kernel, dim3(wgs), dim3(256), 0, s1, out, im);
    hipEvent_t d0, d1; hipEventCreate(&d0); hipEventCreate(&d1);
    hipEventRecord(d0, s0); hipEventRecord(d1, s1);
    hipStreamWaitEvent(0, d0, 0); hipStreamWaitEvent(0, d1, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms);
    hipEventDestroy(d0); hipEventDestroy(d1);
  }
  return best;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  hipStream_t s0, s1; hipStreamCreateWithFlags(&s0, hipStreamNonBlocking); hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  const int im = 40000, iv = 120000;
  for (int wgs : {256, 512}) {
    const char* names[10] = {"M alone", "V alone", "M || V", "M || M", "V || V", "B alone", "B || V", "Vprio3 first || M", "V first || M", "M first || Vprio3"};
    for (int w = 0; w < 10; ++w) {
      printf("%4d workgroups (%d wave/SIMD per kernel)  %-18s %7.3f ms\n", wgs, wgs / 256, names[w], run(w, s0, s1, out, wgs, im, iv));
      fflush(stdout);
    }
  }
  return 0;
}
