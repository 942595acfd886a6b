// ball_query.hip -- ordered first-k ball query for gfx950.
//
// Replaces query_ball_point_kernel, pvn3d/_ext-src/src/ball_query_gpu.cu:9-44 (reference):
// for each centre, the first `nsample` indices k (ascending) with d2 < radius^2, padded with
// the first hit; all-zero row when there is no hit.
//
// Design: the reference gives each THREAD one centre and lets it walk all n points serially
// (one block per cloud).  Here each WAVE owns CPW centres and walks the cloud 64 points at a
// time: a lane holds one point, the centres are wave-uniform (SGPRs), `d2 < r2` lands in an
// SGPR-pair mask (v_cmp = ballot for free), the in-order slot of a hit is
// cnt + mbcnt(mask), and a centre stops being evaluated once its ball is full -- the ordered
// "first nsample" semantics are preserved because lanes are in index order and steps are
// in index order.  With PAIR the two radii of a multi-scale-grouping level share the distance
// evaluation.  grid = (ceil(m / (4*CPW)), b), 4 waves per workgroup.
// Arithmetic: -ffp-contract=off, d2 = ((dx*dx + dy*dy) + dz*dz), dx = centre - point.
#include "common.h"

namespace {

template <int CPW, bool PAIR>
__global__ __launch_bounds__(256) void ball_query_kernel(int n, int m, float r2a, int nsa,
                                                         float r2b, int nsb,
                                                         const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz,
                                                         int* __restrict__ idxa,
                                                         int* __restrict__ idxb) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y;
  const int j0 = (blockIdx.x * 4 + wave) * CPW;
  if (j0 >= m) return;
  xyz += (size_t)bi * n * 3;
  new_xyz += (size_t)bi * m * 3;
  idxa += (size_t)bi * m * nsa;
  if (PAIR) idxb += (size_t)bi * m * nsb;

  float cx[CPW], cy[CPW], cz[CPW];
  int cnta[CPW], firsta[CPW], cntb[CPW], firstb[CPW];
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    const bool ok = j < m;
    const int jj = ok ? j : j0;
    cx[c] = new_xyz[jj * 3 + 0];
    cy[c] = new_xyz[jj * 3 + 1];
    cz[c] = new_xyz[jj * 3 + 2];
    cnta[c] = ok ? 0 : nsa;
    cntb[c] = ok ? 0 : nsb;
    firsta[c] = 0;
    firstb[c] = 0;
  }

  for (int k0 = 0; k0 < n; k0 += 64) {
    bool all_done = true;
#pragma unroll
    for (int c = 0; c < CPW; ++c)
      all_done = all_done && (cnta[c] >= nsa) && (!PAIR || cntb[c] >= nsb);
    if (all_done) break;
    const int k = k0 + lane;
    float x = __builtin_inff(), y = 0.f, z = 0.f;
    if (k < n) {
      x = xyz[k * 3 + 0];
      y = xyz[k * 3 + 1];
      z = xyz[k * 3 + 2];
    }
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const bool da = cnta[c] >= nsa;
      const bool db = PAIR ? (cntb[c] >= nsb) : true;
      if (da && db) continue;
      const float dx = cx[c] - x, dy = cy[c] - y, dz = cz[c] - z;
      const float d2 = dx * dx + dy * dy + dz * dz;
      const int j = j0 + c;
      if (!da) {
        const bool hit = d2 < r2a;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          const int slot = cnta[c] + pvn3d_mbcnt(mask);
          if (hit && slot < nsa) idxa[(size_t)j * nsa + slot] = k;
          if (cnta[c] == 0) firsta[c] = k0 + __builtin_ctzll(mask);
          cnta[c] += __builtin_popcountll(mask);
        }
      }
      if (PAIR && !db) {
        const bool hit = d2 < r2b;
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          const int slot = cntb[c] + pvn3d_mbcnt(mask);
          if (hit && slot < nsb) idxb[(size_t)j * nsb + slot] = k;
          if (cntb[c] == 0) firstb[c] = k0 + __builtin_ctzll(mask);
          cntb[c] += __builtin_popcountll(mask);
        }
      }
    }
  }
  // pad: slots [cnt, nsample) repeat the first hit (ball_query_gpu.cu:33-37); no hit -> zeros
#pragma unroll
  for (int c = 0; c < CPW; ++c) {
    const int j = j0 + c;
    if (j >= m) break;
    for (int l = lane; l < nsa; l += 64)
      if (l >= cnta[c]) idxa[(size_t)j * nsa + l] = firsta[c];
    if (PAIR)
      for (int l = lane; l < nsb; l += 64)
        if (l >= cntb[c]) idxb[(size_t)j * nsb + l] = firstb[c];
  }
}

template <bool PAIR>
int launch_ball_query(int b, int n, int m, float r2a, int nsa, float r2b, int nsb,
                      const float* new_xyz, const float* xyz, int* idxa, int* idxb,
                      hipStream_t st) {
  // enough waves to cover 256 CUs x 4 SIMDs a few times over; more centres per wave = fewer
  // passes over the cloud.
  const long long centres = (long long)b * m;
  int cpw = 8;
  while (cpw > 1 && centres / cpw < 4096) cpw >>= 1;
#define BQ_LAUNCH(CPW)                                                                     \
  hipLaunchKernelGGL((ball_query_kernel<CPW, PAIR>), dim3(pvn3d_ceil_div(m, 4 * CPW), b),  \
                     dim3(256), 0, st, n, m, r2a, nsa, r2b, nsb, new_xyz, xyz, idxa, idxb)
  switch (cpw) {
    case 8: BQ_LAUNCH(8); break;
    case 4: BQ_LAUNCH(4); break;
    case 2: BQ_LAUNCH(2); break;
    default: BQ_LAUNCH(1); break;
  }
#undef BQ_LAUNCH
  PVN3D_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int pvn3d_ball_query(int b, int n, int m, float radius, int nsample,
                                const float* new_xyz, const float* xyz, int* idx,
                                void* stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  if (n < 0 || !new_xyz || !xyz || !idx) return (int)hipErrorInvalidValue;
  const float r2 = radius * radius;  // fp32, ball_query_gpu.cu:22
  return launch_ball_query<false>(b, n, m, r2, nsample, 0.f, 0, new_xyz, xyz, idx, nullptr,
                                  (hipStream_t)stream);
}

extern "C" int pvn3d_ball_query_pair(int b, int n, int m, float radius0, int nsample0,
                                     float radius1, int nsample1, const float* new_xyz,
                                     const float* xyz, int* idx0, int* idx1, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (nsample0 <= 0 || nsample1 <= 0 || n < 0 || !new_xyz || !xyz || !idx0 || !idx1)
    return (int)hipErrorInvalidValue;
  return launch_ball_query<true>(b, n, m, radius0 * radius0, nsample0, radius1 * radius1,
                                 nsample1, new_xyz, xyz, idx0, idx1, (hipStream_t)stream);
}
