// relabel.hip -- YCB centre-cluster re-labelling of cal_frame_poses
// (pvn3d/lib/utils/pvn3d_eval_utils.py:58-72) for a batch of frames.
//
// Reference, per frame: ctr_dis[i][c] = |pred_ctr_i - ctr_c| over the classes present in the mask
// (ascending class id), (min_dis, min_idx) = torch.min(dim=1) (first minimum), closest = class of
// min_idx; every labelled point (mask > 0) whose min_dis < 0.8 * ycb_r_lst[closest-1] takes the
// label `closest` (:66-71).  The reference builds (n_pts, n_ctrs, 3) tensors and loops over the
// classes on the host; here one thread per point scans the <= 21 class centres of its frame
// (LDS) and the kernel also reports which classes survive in the new mask.
#include "common.h"

namespace {

// grid (ceil(n/256), n_frames)
__global__ __launch_bounds__(256) void relabel_kernel(int n, int n_cls_m1, const float* __restrict__ pcld,
                                                      const float* __restrict__ ctr_of,
                                                      const int* __restrict__ mask,
                                                      const float* __restrict__ ctrs,
                                                      const int* __restrict__ present,
                                                      const float* __restrict__ thr,
                                                      int* __restrict__ new_mask,
                                                      int* __restrict__ present_new) {
  __shared__ float s_c[64 * 3];
  __shared__ float s_thr[64];
  __shared__ int s_present[64];
  const int f = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < n_cls_m1) {
    s_c[tid * 3 + 0] = ctrs[((size_t)f * n_cls_m1 + tid) * 3 + 0];
    s_c[tid * 3 + 1] = ctrs[((size_t)f * n_cls_m1 + tid) * 3 + 1];
    s_c[tid * 3 + 2] = ctrs[((size_t)f * n_cls_m1 + tid) * 3 + 2];
    s_thr[tid] = thr[tid];
    s_present[tid] = present[(size_t)f * n_cls_m1 + tid];
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + tid;
  if (i >= n) return;
  const size_t r = (size_t)f * n + i;
  const int m = mask[r];
  // pred_ctr = pcld - ctr_of[0]  (:41)
  const float px = pcld[r * 3 + 0] - ctr_of[r * 3 + 0];
  const float py = pcld[r * 3 + 1] - ctr_of[r * 3 + 1];
  const float pz = pcld[r * 3 + 2] - ctr_of[r * 3 + 2];
  float best = 0.f;
  int best_c = -1;
  for (int c = 0; c < n_cls_m1; ++c) {
    if (!s_present[c]) continue;
    const float dx = px - s_c[c * 3 + 0], dy = py - s_c[c * 3 + 1], dz = pz - s_c[c * 3 + 2];
    const float d = sqrtf((dx * dx + dy * dy) + dz * dz);
    if (best_c < 0 || d < best) { best = d; best_c = c; }     // strict <: first minimum wins
  }
  int out = m;
  if (m > 0 && best_c >= 0 && best < s_thr[best_c]) out = best_c + 1;
  new_mask[r] = out;
  if (out > 0 && out <= n_cls_m1) present_new[(size_t)f * n_cls_m1 + out - 1] = 1;   // idempotent
}

}  // namespace

extern "C" int pvn3d_relabel_by_centre(int n_frames, int n_pts, int n_cls_m1, const float* pcld,
                                       const float* ctr_of, const int* mask, const float* ctrs,
                                       const int* present, const float* thr, int* new_mask,
                                       int* present_new, void* stream) {
  if (n_frames <= 0 || n_pts <= 0) return 0;
  if (n_cls_m1 <= 0 || n_cls_m1 > 64 || !pcld || !ctr_of || !mask || !ctrs || !present || !thr ||
      !new_mask || !present_new)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  pvn3d_fill_u32(present_new, 0u, (size_t)n_frames * n_cls_m1, st);
  PVN3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(relabel_kernel, dim3(pvn3d_ceil_div(n_pts, 256), n_frames), dim3(256), 0, st, n_pts,
                     n_cls_m1, pcld, ctr_of, mask, ctrs, present, thr, new_mask, present_new);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
