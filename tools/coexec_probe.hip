// coexec_probe.hip -- do a VALU-only kernel and the fused MFMA MLP kernels co-execute? (tuning aid, not shipped)
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/coexec_probe.hip -o tools/libcoexec_probe.so
#include <hip/hip_runtime.h>

// MODE 0: pure VALU (fma + exp), no LDS, no memory.  MODE 1: the same plus one broadcast ds_read_b128 per step.
template <int MODE>
__global__ __launch_bounds__(256) void valu_kernel(float* out, int iters) {
  __shared__ float4 s[512];
  if (MODE == 1) { for (int i = threadIdx.x; i < 512; i += 256) s[i] = make_float4(i * 1e-3f, 0.5f, 0.25f, -1.f); __syncthreads(); }
  float a = threadIdx.x * 1e-3f, b = 0.f, c = 0.f, d = 0.f, w = 0.f;
  for (int i = 0; i < iters; ++i) {
    float4 p = MODE == 1 ? s[i & 511] : make_float4(a * 0.5f, 0.5f, 0.25f, -1.f);
    const float e = fmaf(a, p.x, fmaf(b, p.y, fmaf(c, p.z, p.w)));
    const float ww = __builtin_amdgcn_exp2f(e);
    w += ww; b = fmaf(ww, p.x, b); c = fmaf(ww, p.y, c); d = fmaf(ww, p.z, d);
    a = a * 0.999f + 1e-4f;
  }
  if (w == 12345.f) out[threadIdx.x] = b + c + d;
}

extern "C" int coexec_launch(int mode, int blocks, int iters, float* out, void* stream) {
  if (mode == 0) hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  else hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
