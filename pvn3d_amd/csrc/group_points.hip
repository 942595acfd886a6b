// group_points.hip -- neighbourhood gather (the HBM-bound half of set abstraction), gfx950.
//
// Replaces group_points_kernel / group_points_grad_kernel,
// pvn3d/_ext-src/src/group_points_gpu.cu:8-28, 43-64 (reference), and fuses the caller's
// `grouped_xyz -= new_xyz` and `torch.cat` (pvn3d/lib/pointnet2_utils/pointnet2_utils.py:311-321)
// into the same pass by writing straight into the concatenated (b, 3+c, npoint, nsample) tensor.
//
// Traffic: the output (4*C*npoint*nsample bytes) dominates; inputs (4*C*n) are re-read from
// L2.  Layout: out[b][l][j][s] -- for a fixed channel l the (j,s) plane is contiguous, so a
// thread owns 4 consecutive (j,s) positions, keeps their 4 neighbour indices in registers for
// its whole channel loop and issues one 16-byte store per channel (1 KiB per wave instruction).
// The reference's thread writes with stride nsample and loops k serially.
#include "common.h"

namespace {

// grid: (ceil(P/1024), n_chunks, b); P = npoints*nsample, P % 4 == 0
__global__ __launch_bounds__(256) void group_points_vec4_kernel(
    int c, int n, int P, int cch, const float* __restrict__ points,
    const int* __restrict__ idx, float* __restrict__ out, size_t out_batch_stride) {
  const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= P) return;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * cch;
  const int c1 = min(c0 + cch, c);
  const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)bi * P + p0);
  const float* row = points + ((size_t)bi * c + c0) * n;
  float* o = out + (size_t)bi * out_batch_stride + (size_t)c0 * P + p0;
  int l = c0;
  for (; l + 4 <= c1; l += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* r = row + (size_t)u * n;
      v[u] = make_float4(r[id.x], r[id.y], r[id.z], r[id.w]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(o + (size_t)u * P) = v[u];
    row += (size_t)4 * n;
    o += (size_t)4 * P;
  }
  for (; l < c1; ++l) {
    *reinterpret_cast<float4*>(o) = make_float4(row[id.x], row[id.y], row[id.z], row[id.w]);
    row += n;
    o += P;
  }
}

// any-P fallback, one output element per thread.  grid: (ceil(P/256), c, b)
__global__ __launch_bounds__(256) void group_points_scalar_kernel(
    int c, int n, int P, const float* __restrict__ points, const int* __restrict__ idx,
    float* __restrict__ out, size_t out_batch_stride) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  out[(size_t)bi * out_batch_stride + (size_t)l * P + p] =
      points[((size_t)bi * c + l) * n + idx[(size_t)bi * P + p]];
}

// channels [0,3) of QueryAndGroup's output: xyz[idx] - new_xyz[j].  xyz is (b,n,3) AoS.
// grid: (ceil(P/256), 1, b)
__global__ __launch_bounds__(256) void group_xyz_rel_kernel(
    int n, int m, int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ idx, float* __restrict__ out, size_t out_batch_stride) {
  const int P = m * nsample;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int bi = blockIdx.z;
  const int j = p / nsample;
  const int k = idx[(size_t)bi * P + p];
  const float* q = xyz + ((size_t)bi * n + k) * 3;
  const float* cq = new_xyz + ((size_t)bi * m + j) * 3;
  float* o = out + (size_t)bi * out_batch_stride + p;
  o[0] = q[0] - cq[0];
  o[(size_t)P] = q[1] - cq[1];
  o[(size_t)2 * P] = q[2] - cq[2];
}

// grid: (ceil(P/256), c, b)
__global__ __launch_bounds__(256) void group_points_grad_kernel(
    int c, int n, int P, const float* __restrict__ grad_out, const int* __restrict__ idx,
    float* __restrict__ grad_points) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  atomicAdd(grad_points + ((size_t)bi * c + l) * n + idx[(size_t)bi * P + p],
            grad_out[((size_t)bi * c + l) * P + p]);
}

int launch_group(int b, int c, int n, int P, const float* points, const int* idx, float* out,
                 size_t out_batch_stride, hipStream_t st) {
  if (b <= 0 || c <= 0 || P <= 0) return 0;
  const bool aligned = (P % 4 == 0) && (out_batch_stride % 4 == 0) &&
                       (((uintptr_t)out & 15) == 0) && (((uintptr_t)idx & 15) == 0);
  if (aligned) {
    const int gx = pvn3d_ceil_div(P, 1024);
    // split channels until there are a few thousand workgroups (256 CUs x 8)
    int chunks = pvn3d_ceil_div(2048, gx * b);
    if (chunks < 1) chunks = 1;
    if (chunks > c) chunks = c;
    const int cch = pvn3d_ceil_div(c, chunks);
    chunks = pvn3d_ceil_div(c, cch);
    hipLaunchKernelGGL(group_points_vec4_kernel, dim3(gx, chunks, b), dim3(256), 0, st, c, n, P,
                       cch, points, idx, out, out_batch_stride);
  } else {
    hipLaunchKernelGGL(group_points_scalar_kernel, dim3(pvn3d_ceil_div(P, 256), c, b),
                       dim3(256), 0, st, c, n, P, points, idx, out, out_batch_stride);
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int pvn3d_group_points(int b, int c, int n, int npoints, int nsample,
                                  const float* points, const int* idx, float* out,
                                  void* stream) {
  const int P = npoints * nsample;
  return launch_group(b, c, n, P, points, idx, out, (size_t)c * P, (hipStream_t)stream);
}

extern "C" int pvn3d_group_xyz_features(int b, int n, int m, int c, int nsample, int use_xyz,
                                        const float* xyz, const float* new_xyz,
                                        const float* features, const int* idx, float* out,
                                        void* stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int P = m * nsample;
  const int c_out = (use_xyz ? 3 : 0) + (features ? c : 0);
  if (c_out == 0) return (int)hipErrorInvalidValue;
  const size_t bstride = (size_t)c_out * P;
  if (use_xyz) {
    hipLaunchKernelGGL(group_xyz_rel_kernel, dim3(pvn3d_ceil_div(P, 256), 1, b), dim3(256), 0, st,
                       n, m, nsample, xyz, new_xyz, idx, out, bstride);
    PVN3D_LAUNCH_CHECK();
  }
  if (features && c > 0)
    return launch_group(b, c, n, P, features, idx, out + (use_xyz ? (size_t)3 * P : 0), bstride,
                        st);
  return 0;
}

extern "C" int pvn3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                       const float* grad_out, const int* idx,
                                       float* grad_points, void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, st));
  const int P = npoints * nsample;
  if (P <= 0) return 0;
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(pvn3d_ceil_div(P, 256), c, b), dim3(256), 0,
                     st, c, n, P, grad_out, idx, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
