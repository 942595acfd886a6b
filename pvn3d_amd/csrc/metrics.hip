// metrics.hip -- ADD and ADD-S pose distances for a batch of (predicted, ground-truth) poses.
//
// Restates Basic_Utils.cal_add_cuda / cal_adds_cuda (pvn3d/lib/utils/basic_utils.py:617-635),
// called per object from eval_metric / eval_metric_lm (pvn3d_eval_utils.py:113-136, 204-221):
//   pd_i = R_p x_i + t_p,  gt_i = R_g x_i + t_g          (torch.mm(p3ds, R^T) + t)
//   ADD   = mean_i |pd_i - gt_i|
//   ADD-S = mean_i min_j |pd_j - gt_i|                    (dis[i][j] = |pd_j - gt_i|, min over j)
// The reference materialises (N,N,3) tensors per object (N = 2000-8000 mesh points) and syncs
// with .item(); here one launch serves every instance of the batch: workgroup (tile, instance)
// keeps 256 gt points in registers, streams the predicted points through LDS, and writes one
// partial sum per tile; a second tiny kernel adds the partials in a fixed order (deterministic,
// no float atomics).
#include "common.h"

namespace {

constexpr int MT_THREADS = 256;
constexpr int MT_CHUNK = 1024;

__device__ __forceinline__ float3 xform(const float* __restrict__ RT, float x, float y, float z) {
  // row-major (3,4): [R | t]; same product order as torch.mm(p3ds, R^T) + t (k ascending)
  float3 o;
  o.x = ((x * RT[0] + y * RT[1]) + z * RT[2]) + RT[3];
  o.y = ((x * RT[4] + y * RT[5]) + z * RT[6]) + RT[7];
  o.z = ((x * RT[8] + y * RT[9]) + z * RT[10]) + RT[11];
  return o;
}

// grid (tiles, n_inst); partial[inst][tile][2]
__global__ __launch_bounds__(MT_THREADS) void add_adds_partial_kernel(
    const float* __restrict__ pts, const int* __restrict__ pts_off, const float* __restrict__ pred_RT,
    const float* __restrict__ gt_RT, int max_tiles, float* __restrict__ partial) {
  __shared__ float4 s_pd[MT_CHUNK];
  __shared__ float s_red[2][MT_THREADS / 64];
  const int inst = blockIdx.y;
  const int base = pts_off[inst];
  const int n = pts_off[inst + 1] - base;
  const int tile0 = blockIdx.x * MT_THREADS;
  if (tile0 >= n) return;
  const int tid = threadIdx.x;
  const float* P = pred_RT + inst * 12;
  const float* G = gt_RT + inst * 12;
  const int i = tile0 + tid;
  float3 gt = make_float3(0.f, 0.f, 0.f);
  float add_d = 0.f;
  if (i < n) {
    const float* x = pts + (size_t)(base + i) * 3;
    gt = xform(G, x[0], x[1], x[2]);
    const float3 pd = xform(P, x[0], x[1], x[2]);
    const float dx = pd.x - gt.x, dy = pd.y - gt.y, dz = pd.z - gt.z;
    add_d = sqrtf((dx * dx + dy * dy) + dz * dz);
  }
  float best = 3.0e38f;
  for (int j0 = 0; j0 < n; j0 += MT_CHUNK) {
    const int cnt = min(MT_CHUNK, n - j0);
    __syncthreads();
    for (int q = tid; q < cnt; q += MT_THREADS) {
      const float* x = pts + (size_t)(base + j0 + q) * 3;
      const float3 pd = xform(P, x[0], x[1], x[2]);
      s_pd[q] = make_float4(pd.x, pd.y, pd.z, 0.f);
    }
    __syncthreads();
    for (int q = 0; q < cnt; ++q) {
      const float4 p = s_pd[q];
      const float dx = p.x - gt.x, dy = p.y - gt.y, dz = p.z - gt.z;
      best = fminf(best, (dx * dx + dy * dy) + dz * dz);
    }
  }
  float adds_d = (i < n) ? sqrtf(best) : 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    add_d += __shfl_xor(add_d, o, 64);
    adds_d += __shfl_xor(adds_d, o, 64);
  }
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = add_d; s_red[1][tid >> 6] = adds_d; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, s = 0.f;
    for (int w = 0; w < MT_THREADS / 64; ++w) { a += s_red[0][w]; s += s_red[1][w]; }
    partial[((size_t)inst * max_tiles + blockIdx.x) * 2 + 0] = a;
    partial[((size_t)inst * max_tiles + blockIdx.x) * 2 + 1] = s;
  }
}

__global__ void add_adds_final_kernel(const int* __restrict__ pts_off, int n_inst, int max_tiles,
                                      const float* __restrict__ partial, float* __restrict__ add_out,
                                      float* __restrict__ adds_out) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= n_inst) return;
  const int n = pts_off[inst + 1] - pts_off[inst];
  const int tiles = (n + MT_THREADS - 1) / MT_THREADS;
  float a = 0.f, s = 0.f;
  for (int t = 0; t < tiles; ++t) {
    a += partial[((size_t)inst * max_tiles + t) * 2 + 0];
    s += partial[((size_t)inst * max_tiles + t) * 2 + 1];
  }
  add_out[inst] = n > 0 ? a / (float)n : 0.f;
  adds_out[inst] = n > 0 ? s / (float)n : 0.f;
}

}  // namespace

extern "C" size_t pvn3d_add_adds_workspace_bytes(int n_inst, int max_pts) {
  return (size_t)n_inst * pvn3d_ceil_div(max_pts > 0 ? max_pts : 1, MT_THREADS) * 2 * sizeof(float);
}

extern "C" int pvn3d_add_adds_batch(int n_inst, int max_pts, const float* pts, const int* pts_off,
                                    const float* pred_RT, const float* gt_RT, void* workspace,
                                    size_t workspace_bytes, float* add_out, float* adds_out,
                                    void* stream) {
  if (n_inst <= 0) return 0;
  if (max_pts <= 0 || !pts || !pts_off || !pred_RT || !gt_RT || !workspace || !add_out || !adds_out ||
      workspace_bytes < pvn3d_add_adds_workspace_bytes(n_inst, max_pts))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int tiles = pvn3d_ceil_div(max_pts, MT_THREADS);
  hipLaunchKernelGGL(add_adds_partial_kernel, dim3(tiles, n_inst), dim3(MT_THREADS), 0, st, pts, pts_off,
                     pred_RT, gt_RT, tiles, (float*)workspace);
  PVN3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(add_adds_final_kernel, dim3(pvn3d_ceil_div(n_inst, 64)), dim3(64), 0, st, pts_off,
                     n_inst, tiles, (const float*)workspace, add_out, adds_out);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
