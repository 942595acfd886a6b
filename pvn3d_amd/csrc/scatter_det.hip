// scatter_det.hip -- order-independent (bit-reproducible) backward scatters.
//
// The reference's backward kernels (group_points_grad_kernel group_points_gpu.cu:43-64,
// gather_points_grad_kernel sampling_gpu.cu:34-47, three_interpolate_grad_kernel
// interpolate_gpu.cu:116-143) accumulate with float atomicAdd: the result depends on the order in
// which the atomics land, i.e. it changes from run to run in the last bits.  SURVEY.md 8(f)
// rank 2 asks for a deterministic alternative.  Sorting the contributions by target costs a
// segmented sort per call; instead the contributions are accumulated in 64-bit FIXED POINT:
//   1. amax = max |grad_out| (x max |weight|)            (atomicMax on the bit pattern)
//   2. every term t is converted to  q = llrint(t * 2^e),  e = 61 - ceil(log2(#terms)) - exponent(amax) - 1,
//      so that no sum of #terms values can overflow an int64, and added with an integer atomic
//      (integer addition is associative: any order gives the same bits);
//   3. grad = (float)(q_sum * 2^-e), one rounding.
// With e chosen this way the quantisation step is 2^-45 (or finer) relative to amax for up to
// 65536 terms per target: the absolute error of a sum is <= #terms * 2^-46 * amax, far below the
// fp32 rounding of any sum containing a term of that size (a target fed only by terms more than
// 2^20 times smaller than amax keeps this absolute -- not relative -- accuracy).  Non-finite inputs (inf / NaN anywhere in grad_out or weight)
// make the whole output NaN.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t count,
                                                     unsigned* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
    const float v = fabsf(x[i]);
    if (!(v <= m)) m = (v != v) ? __builtin_inff() : v;     // a NaN counts as +inf (fmaxf would drop it)
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));   // non-negative floats order like uints
}

// scale exponent from the bit patterns of the maxima (amax[1] = 1.0f's bits when no weights)
__device__ __forceinline__ int scale_exp(const unsigned* __restrict__ amax, int log2_terms) {
  const float a = __uint_as_float(amax[0]) * __uint_as_float(amax[1]);
  if (!(a > 0.f) || !(a < 3.0e38f)) return 0;
  const int ex = (int)((__float_as_uint(a) >> 23) & 0xff) - 127;     // a < 2^(ex+1)
  return 61 - log2_terms - (ex + 1);
}

// grid (ceil(P/256), c, b): q[b][l][idx[b][p]] += grad_out[b][l][p]
__global__ __launch_bounds__(256) void group_scatter_kernel(int c, int n, int P, int log2_terms,
                                                            const float* __restrict__ grad_out,
                                                            const int* __restrict__ idx,
                                                            const unsigned* __restrict__ amax,
                                                            unsigned long long* __restrict__ q) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  const int e = scale_exp(amax, log2_terms);
  const double t = (double)grad_out[((size_t)bi * c + l) * P + p];
  const long long v = __double2ll_rn(ldexp(t, e));
  atomicAdd(q + ((size_t)bi * c + l) * n + idx[(size_t)bi * P + p], (unsigned long long)v);
}

// grid (ceil(n/256), c, b): q[b][l][idx[b][j][k]] += grad_out[b][l][j] * weight[b][j][k]
__global__ __launch_bounds__(256) void interp_scatter_kernel(int c, int n, int m, int log2_terms,
                                                             const float* __restrict__ grad_out,
                                                             const int* __restrict__ idx,
                                                             const float* __restrict__ weight,
                                                             const unsigned* __restrict__ amax,
                                                             unsigned long long* __restrict__ q) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  const int e = scale_exp(amax, log2_terms);
  const float g = grad_out[((size_t)bi * c + l) * n + j];
  const int* ip = idx + ((size_t)bi * n + j) * 3;
  const float* wp = weight + ((size_t)bi * n + j) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float t = g * wp[k];          // the fp32 product of interpolate_gpu.cu:136-138
    atomicAdd(q + ((size_t)bi * c + l) * m + ip[k], (unsigned long long)__double2ll_rn(ldexp((double)t, e)));
  }
}

__global__ __launch_bounds__(256) void fixed_to_float_kernel(size_t count, int log2_terms,
                                                             const unsigned* __restrict__ amax,
                                                             const unsigned long long* __restrict__ q,
                                                             float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const float a = __uint_as_float(amax[0]) * __uint_as_float(amax[1]);
  if (!(a < 3.0e38f)) {       // an inf / NaN gradient or weight: fixed point cannot represent it --
    out[i] = __builtin_nanf("");   // poison the whole result rather than return finite garbage
    return;
  }
  const int e = scale_exp(amax, log2_terms);
  out[i] = (float)ldexp((double)(long long)q[i], -e);
}

int ceil_log2(long long v) {
  int l = 0;
  while ((1LL << l) < v) ++l;
  return l;
}

struct DetWs {
  unsigned* amax;             // [2]
  unsigned long long* q;      // [count]
};

size_t det_ws_bytes(size_t count) { return 256 + count * sizeof(unsigned long long); }

int det_prepare(void* ws, size_t ws_bytes, size_t count, const float* g, size_t g_count,
                const float* w, size_t w_count, DetWs* out, hipStream_t st) {
  if (!ws || ws_bytes < det_ws_bytes(count)) return (int)hipErrorInvalidValue;
  out->amax = (unsigned*)ws;
  out->q = (unsigned long long*)((char*)ws + 256);
  PVN3D_RETURN_IF_ERR(hipMemsetAsync(ws, 0, det_ws_bytes(count), st));
  const int blocks = (int)((g_count + 256 * 8 - 1) / (256 * 8)) < 1024 ? (int)((g_count + 256 * 8 - 1) / (256 * 8)) : 1024;
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, st, g, g_count, out->amax);
  if (w) {
    const int wb = (int)((w_count + 256 * 8 - 1) / (256 * 8)) < 1024 ? (int)((w_count + 256 * 8 - 1) / (256 * 8)) : 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3(wb > 0 ? wb : 1), dim3(256), 0, st, w, w_count, out->amax + 1);
  } else {
    // |w|max = 1.0f, written by the device (an async copy from a stack variable could outlive it)
    PVN3D_RETURN_IF_ERR(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(out->amax + 1), 0x3f800000, 1, st));
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" size_t pvn3d_scatter_det_workspace_bytes(int b, int c, int n) {
  return det_ws_bytes((size_t)(b > 0 ? b : 0) * (c > 0 ? c : 0) * (n > 0 ? n : 0));
}

extern "C" int pvn3d_group_points_grad_det(int b, int c, int n, int npoints, int nsample,
                                           const float* grad_out, const int* idx, float* grad_points,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int P = npoints * nsample;
  const size_t count = (size_t)b * c * n;
  if (P <= 0) return (int)hipMemsetAsync(grad_points, 0, sizeof(float) * count, st);
  if (!grad_out || !idx || !grad_points) return (int)hipErrorInvalidValue;
  DetWs ws;
  const int rc = det_prepare(workspace, workspace_bytes, count, grad_out, (size_t)b * c * P, nullptr, 0, &ws, st);
  if (rc) return rc;
  const int lt = ceil_log2(P);
  hipLaunchKernelGGL(group_scatter_kernel, dim3(pvn3d_ceil_div(P, 256), c, b), dim3(256), 0, st, c, n, P, lt,
                     grad_out, idx, ws.amax, ws.q);
  hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, count, lt,
                     ws.amax, ws.q, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_three_interpolate_grad_det(int b, int c, int n, int m, const float* grad_out,
                                                const int* idx, const float* weight,
                                                float* grad_points, void* workspace,
                                                size_t workspace_bytes, void* stream) {
  if (b <= 0 || c <= 0 || m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const size_t count = (size_t)b * c * m;
  if (n <= 0) return (int)hipMemsetAsync(grad_points, 0, sizeof(float) * count, st);
  if (!grad_out || !idx || !weight || !grad_points) return (int)hipErrorInvalidValue;
  DetWs ws;
  const int rc = det_prepare(workspace, workspace_bytes, count, grad_out, (size_t)b * c * n, weight,
                             (size_t)b * n * 3, &ws, st);
  if (rc) return rc;
  const int lt = ceil_log2((long long)n * 3);
  hipLaunchKernelGGL(interp_scatter_kernel, dim3(pvn3d_ceil_div(n, 256), c, b), dim3(256), 0, st, c, n, m, lt,
                     grad_out, idx, weight, ws.amax, ws.q);
  hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, count, lt,
                     ws.amax, ws.q, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
