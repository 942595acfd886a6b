// pose.hip -- batched least-squares rigid fit (Kabsch) on the device, gfx950.
//
// Replaces best_fit_transform, pvn3d/lib/utils/basic_utils.py:47-80 (reference), which the
// reference runs on the host after a device->host copy of the voted keypoints
// (pvn3d/lib/utils/pvn3d_eval_utils.py:103-107, 196-199).  Keeping it on the device removes
// that per-object synchronisation; the work itself is tiny (9 points, one 3x3 SVD): one
// wave per point set, fp64, every lane carrying the same scalar computation.  SVD: one-sided Jacobi, singular values sorted
// descending like LAPACK so the reflection fix negates the same row of Vt (:70-72).
#include "common.h"

namespace {

__device__ void jacobi_svd3(const double H[9], double U[9], double S[3], double V[9]) {
  double G[9];
  for (int i = 0; i < 9; ++i) {
    G[i] = H[i];
    V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    bool rotated = false;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += G[r * 3 + p] * G[r * 3 + p];
          beta += G[r * 3 + q] * G[r * 3 + q];
          gamma += G[r * 3 + p] * G[r * 3 + q];
        }
        off += fabs(gamma);
        // columns orthogonal to 4 ulp: the usual one-sided Jacobi test.  (A threshold below the rounding unit, 1e-17
        // until round 4, was never met -- every rotation leaves ~1e-16 of noise -- so all 60 sweeps ran: 55 us.)
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 8.9e-16 * sqrt(alpha * beta)) continue;
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double gp = G[r * 3 + p], gq = G[r * 3 + q];
          G[r * 3 + p] = c * gp - s * gq;
          G[r * 3 + q] = s * gp + c * gq;
          const double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - s * vq;
          V[r * 3 + q] = s * vp + c * vq;
        }
      }
    // a sweep that rotated nothing left G and V as they were, so every later sweep would do the same: stopping here
    // returns the same bits (the absolute test alone ran all 60 sweeps on metre-scale data -- 85 us per launch)
    if (off < 1e-30 || !rotated) break;
  }
  double nrm[3];
  for (int c = 0; c < 3; ++c)
    nrm[c] = sqrt(G[c] * G[c] + G[3 + c] * G[3 + c] + G[6 + c] * G[6 + c]);
  int o0 = 0, o1 = 1, o2 = 2;
  if (nrm[o1] > nrm[o0]) { int t = o0; o0 = o1; o1 = t; }
  if (nrm[o2] > nrm[o0]) { int t = o0; o0 = o2; o2 = t; }
  if (nrm[o2] > nrm[o1]) { int t = o1; o1 = o2; o2 = t; }
  const int order[3] = {o0, o1, o2};
  double Vs[9];
  for (int c = 0; c < 3; ++c) {
    const int oc = order[c];
    S[c] = nrm[oc];
    for (int r = 0; r < 3; ++r) {
      Vs[r * 3 + c] = V[r * 3 + oc];
      U[r * 3 + c] = nrm[oc] > 0 ? G[r * 3 + oc] / nrm[oc] : 0.0;
    }
  }
  for (int i = 0; i < 9; ++i) V[i] = Vs[i];
  const double s0 = S[0] > 0 ? S[0] : 1.0;
  if (S[2] <= 1e-14 * s0) {  // rank-deficient: complete U to an orthonormal basis
    if (S[1] <= 1e-14 * s0) {
      double u0[3] = {U[0], U[3], U[6]};
      if (S[0] <= 0) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
      double a[3] = {0, 0, 0};
      const int mi = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2)
                                               : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
      a[mi] = 1.0;
      const double u1[3] = {u0[1] * a[2] - u0[2] * a[1], u0[2] * a[0] - u0[0] * a[2],
                            u0[0] * a[1] - u0[1] * a[0]};
      const double l = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      U[1] = u1[0] / l; U[4] = u1[1] / l; U[7] = u1[2] / l;
    }
    const double a0 = U[0], a1 = U[3], a2 = U[6], b0 = U[1], b1 = U[4], b2 = U[7];
    U[2] = a1 * b2 - a2 * b1; U[5] = a2 * b0 - a0 * b2; U[8] = a0 * b1 - a1 * b0;
  }
}

// One wave per point set.  The arithmetic is that of one lane walking the points in order (every lane carries the same
// sums, so the bits do not depend on the shape); the wave only fetches the points -- lane i loads point i, the values
// go round by readlane.  A single lane per set did 2 x npts dependent global round trips before its first flop,
// most of the kernel's 85 us (it runs once per call, alone on the device: its latency is the call's).
__global__ __launch_bounds__(64) void best_fit_transform_kernel(int n_sets, int npts, const float* __restrict__ A,
                                                                const float* __restrict__ B,
                                                                const int* __restrict__ valid,
                                                                double* __restrict__ T) {
  const int s = blockIdx.x;
  const int lane = threadIdx.x;
  if (s >= n_sets) return;
  double* To = T + (size_t)s * 12;
  if (valid && !valid[s]) {  // np.identity(4)[:3,:]  (pvn3d_eval_utils.py:172-173)
    if (lane < 12) To[lane] = (lane % 5 == 0) ? 1.0 : 0.0;
    return;
  }
  const float* a = A + (size_t)s * npts * 3;
  const float* b = B + (size_t)s * npts * 3;
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int i0 = 0; i0 < npts; i0 += 64) {
    const int i = min(i0 + lane, npts - 1);
    const float pa0 = a[i * 3 + 0], pa1 = a[i * 3 + 1], pa2 = a[i * 3 + 2];
    const float pb0 = b[i * 3 + 0], pb1 = b[i * 3 + 1], pb2 = b[i * 3 + 2];
    const int cnt = min(64, npts - i0);
    for (int j = 0; j < cnt; ++j) {
      ca[0] += __shfl(pa0, j, 64); ca[1] += __shfl(pa1, j, 64); ca[2] += __shfl(pa2, j, 64);
      cb[0] += __shfl(pb0, j, 64); cb[1] += __shfl(pb1, j, 64); cb[2] += __shfl(pb2, j, 64);
    }
  }
  for (int d = 0; d < 3; ++d) { ca[d] /= npts; cb[d] /= npts; }
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i0 = 0; i0 < npts; i0 += 64) {
    const int i = min(i0 + lane, npts - 1);
    const float pa[3] = {a[i * 3 + 0], a[i * 3 + 1], a[i * 3 + 2]};
    const float pb[3] = {b[i * 3 + 0], b[i * 3 + 1], b[i * 3 + 2]};
    const int cnt = min(64, npts - i0);
    for (int j = 0; j < cnt; ++j) {
      double da[3], db[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        da[d] = (double)__shfl(pa[d], j, 64) - ca[d];
        db[d] = (double)__shfl(pb[d], j, 64) - cb[d];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) H[r * 3 + c] += da[r] * db[c];
    }
  }
  double U[9], S[3], V[9], R[9];
  jacobi_svd3(H, U, S, V);
  for (int pass = 0; pass < 2; ++pass) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double acc = 0;
        for (int k = 0; k < 3; ++k) acc += V[r * 3 + k] * U[c * 3 + k];
        R[r * 3 + c] = acc;
      }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (pass == 1 || det >= 0) break;
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2];
  }
  if (lane != 0) return;
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) To[r * 4 + c] = R[r * 3 + c];
    To[r * 4 + 3] = cb[r] - (R[r * 3 + 0] * ca[0] + R[r * 3 + 1] * ca[1] + R[r * 3 + 2] * ca[2]);
  }
}

}  // namespace

extern "C" int pvn3d_best_fit_transform(int n_sets, int npts, const float* A, const float* B,
                                        const int* valid, double* T, void* stream) {
  if (n_sets <= 0) return 0;
  if (npts <= 0 || !A || !B || !T) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(best_fit_transform_kernel, dim3(n_sets), dim3(64), 0,
                     (hipStream_t)stream, n_sets, npts, A, B, valid, T);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
