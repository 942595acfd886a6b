// vote_loss.hip -- the keypoint / centre offset ("vote") L1 loss of PVN3D, forward and backward.
//
// Restates of_l1_loss (pvn3d/lib/loss.py:45-73, called through OFLoss :76-90 from
// train/train_linemod_pvn3d.py:191-196) for normalize=True:
//   w_i      = labels[b,i] > 1e-8
//   loss[b,k] = sum_{i,c} w_i * |pred[b,k,i,c] - targ[b,i,k,c]|  /  (sum_i w_i + 1e-3)
// (the denominator sums w over the n_pts axis only: `w.view(bs, n_kpts, -1)` has n_pts entries).
// The reference builds the repeated mask, the permuted target and three (bs,K,N,3) temporaries
// with ~8 elementwise kernels; here one workgroup per (b,k) reads pred and targ once.
// backward: d pred[b,k,i,c] = g[b,k] * w_i * sign(pred - targ) / (sum_i w_i + 1e-3)
// (torch.abs has sign(0) = 0).  Summation order is fixed (thread-strided partial sums + a tree),
// so the loss is bit-reproducible run to run.
#include "common.h"

namespace {

constexpr int VL_THREADS = 256;

// grid (n_kpts, bs)
__global__ __launch_bounds__(VL_THREADS) void of_l1_loss_fwd_kernel(
    int n_kpts, int n_pts, const float* __restrict__ pred, const float* __restrict__ targ,
    const float* __restrict__ labels, float* __restrict__ loss, float* __restrict__ wsum) {
  __shared__ float s_a[VL_THREADS / 64], s_w[VL_THREADS / 64];
  const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const float* p = pred + ((size_t)b * n_kpts + k) * n_pts * 3;
  const float* t = targ + (size_t)b * n_pts * n_kpts * 3 + (size_t)k * 3;
  const float* l = labels + (size_t)b * n_pts;
  float acc = 0.f, wacc = 0.f;
  for (int i = tid; i < n_pts; i += VL_THREADS) {
    const float w = l[i] > 1e-8f ? 1.f : 0.f;
    const float* tt = t + (size_t)i * n_kpts * 3;
    const float d0 = fabsf(p[i * 3 + 0] - tt[0]), d1 = fabsf(p[i * 3 + 1] - tt[1]),
                d2 = fabsf(p[i * 3 + 2] - tt[2]);
    acc += (w * d0 + w * d1) + w * d2;
    wacc += w;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    acc += __shfl_xor(acc, o, 64);
    wacc += __shfl_xor(wacc, o, 64);
  }
  if ((tid & 63) == 0) { s_a[tid >> 6] = acc; s_w[tid >> 6] = wacc; }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, w = 0.f;
    for (int i = 0; i < VL_THREADS / 64; ++i) { a += s_a[i]; w += s_w[i]; }
    loss[b * n_kpts + k] = a / (w + 1e-3f);
    wsum[b * n_kpts + k] = w;
  }
}

// grid (ceil(n_pts/256), n_kpts, bs)
__global__ __launch_bounds__(VL_THREADS) void of_l1_loss_bwd_kernel(
    int n_kpts, int n_pts, const float* __restrict__ pred, const float* __restrict__ targ,
    const float* __restrict__ labels, const float* __restrict__ wsum,
    const float* __restrict__ grad_loss, float* __restrict__ grad_pred) {
  const int i = blockIdx.x * VL_THREADS + threadIdx.x;
  const int k = blockIdx.y, b = blockIdx.z;
  if (i >= n_pts) return;
  const float scale = grad_loss[b * n_kpts + k] / (wsum[b * n_kpts + k] + 1e-3f);
  const float w = labels[(size_t)b * n_pts + i] > 1e-8f ? scale : 0.f;
  const size_t po = (((size_t)b * n_kpts + k) * n_pts + i) * 3;
  const float* tt = targ + (((size_t)b * n_pts + i) * n_kpts + k) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float d = pred[po + c] - tt[c];
    grad_pred[po + c] = d > 0.f ? w : (d < 0.f ? -w : 0.f);
  }
}

}  // namespace

extern "C" int pvn3d_of_l1_loss(int bs, int n_kpts, int n_pts, const float* pred_ofsts,
                                const float* kp_targ_ofst, const float* labels, float* loss,
                                float* wsum, void* stream) {
  if (bs <= 0 || n_kpts <= 0) return 0;
  if (n_pts <= 0 || !pred_ofsts || !kp_targ_ofst || !labels || !loss || !wsum)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(of_l1_loss_fwd_kernel, dim3(n_kpts, bs), dim3(VL_THREADS), 0, (hipStream_t)stream,
                     n_kpts, n_pts, pred_ofsts, kp_targ_ofst, labels, loss, wsum);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_of_l1_loss_grad(int bs, int n_kpts, int n_pts, const float* pred_ofsts,
                                     const float* kp_targ_ofst, const float* labels,
                                     const float* wsum, const float* grad_loss, float* grad_pred,
                                     void* stream) {
  if (bs <= 0 || n_kpts <= 0 || n_pts <= 0) return 0;
  if (!pred_ofsts || !kp_targ_ofst || !labels || !wsum || !grad_loss || !grad_pred)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(of_l1_loss_bwd_kernel, dim3(pvn3d_ceil_div(n_pts, VL_THREADS), n_kpts, bs),
                     dim3(VL_THREADS), 0, (hipStream_t)stream, n_kpts, n_pts, pred_ofsts, kp_targ_ofst,
                     labels, wsum, grad_loss, grad_pred);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
