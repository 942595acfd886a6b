// sampling.hip -- furthest point sampling + gather_points for gfx950.
//
// Replaces pvn3d/_ext-src/src/sampling_gpu.cu (reference): furthest_point_sampling_kernel
// (:69-173), gather_points_kernel (:8-20), gather_points_grad_kernel (:34-47).
//
// FPS design (one workgroup per cloud, the m-1 rounds are inherently serial):
//   * every point and its running min-distance live in VGPRs for the whole kernel
//     (PPT points per thread); global memory is read once.  The reference re-reads xyz and
//     temp from global memory every round.
//   * one wave per SIMD (256 threads, up to 64 points per lane): the round is VALU-issue bound
//     (~12 instructions per point), so fewer, fatter waves minimise the per-round reduction
//     tail: DPP wave reductions (no LDS), one 8-byte LDS slot per wave, ONE barrier per round
//     (double-buffered slots), winner coordinates from an LDS copy of the cloud.  The
//     reference uses 9 __syncthreads per round.
//   * the reference's tie-break is reproduced exactly without emulating its block: its
//     512-thread strided scan + halving tree picks, among equal distances, the smallest
//     bit-reversed (k mod bs), then the smallest k (sampling_gpu.cu:59-65,96-168,
//     bs = opt_n_threads(n)).  With prio(k) = bitrev(k mod bs) * ceil(n/bs) + k / bs the winner
//     is (max d2, then min prio) under ANY reduction order; points are laid out so that a
//     thread's slots ascend in prio and a strict '>' scan suffices inside a thread.
// Arithmetic: this TU is compiled with -ffp-contract=off; d is evaluated as
// ((dx*dx + dy*dy) + dz*dz) with one rounding per operation (oracle/pvn3d_oracle.c).
#include "common.h"

namespace {

// ---- wave64 reductions on the DPP path (no LDS traffic): quad swaps, half-row / row mirrors,
// then row_bcast15 / row_bcast31; the full result lands in lane 63 and is broadcast as a scalar.
#define PVN3D_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rmask), 0xf, false)

// Distances are >= 0 (or a negative "invalid" marker), so their bit patterns order like signed
// integers: v_max_i32 needs no NaN canonicalisation and fuses with the DPP operand.
__device__ __forceinline__ int wave_max_i32(int v) {
  v = max(v, PVN3D_DPP(v, 0xB1, 0xf));   // quad_perm [1,0,3,2]
  v = max(v, PVN3D_DPP(v, 0x4E, 0xf));   // quad_perm [2,3,0,1]
  v = max(v, PVN3D_DPP(v, 0x141, 0xf));  // row_half_mirror
  v = max(v, PVN3D_DPP(v, 0x140, 0xf));  // row_mirror
  v = max(v, PVN3D_DPP(v, 0x142, 0xa));  // row_bcast:15
  v = max(v, PVN3D_DPP(v, 0x143, 0xc));  // row_bcast:31
  return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)PVN3D_DPP((int)v, 0xB1, 0xf));
  v = min(v, (unsigned)PVN3D_DPP((int)v, 0x4E, 0xf));
  v = min(v, (unsigned)PVN3D_DPP((int)v, 0x141, 0xf));
  v = min(v, (unsigned)PVN3D_DPP((int)v, 0x140, 0xf));
  v = min(v, (unsigned)PVN3D_DPP((int)v, 0x142, 0xa));
  v = min(v, (unsigned)PVN3D_DPP((int)v, 0x143, 0xc));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ long long shfl_xor_i64(long long v, int mask) {
  int lo = __shfl_xor((int)(v & 0xffffffffLL), mask, 64);
  int hi = __shfl_xor((int)(v >> 32), mask, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    long long o = shfl_xor_i64(v, s);
    v = o > v ? o : v;
  }
  return v;
}

// Tie-break priority of point k in the reference block (bs = opt_n_threads(n) threads,
// L = log2 bs, Q = ceil(n/bs)): smaller = wins ties.  A bijection between k and prio.
__device__ __forceinline__ unsigned fps_prio(int k, int L, int Q) {
  unsigned r = L ? (__brev((unsigned)k & ((1u << L) - 1u)) >> (32 - L)) : 0u;
  return r * (unsigned)Q + ((unsigned)k >> L);
}

__device__ __forceinline__ int fps_prio_to_k(unsigned p, int L, int Q) {
  unsigned r = p / (unsigned)Q, q = p % (unsigned)Q;
  unsigned t = L ? (__brev(r) >> (32 - L)) : 0u;
  return (int)((q << L) | t);
}

// mag <= 1e-3 with mag fp32 and the literal double (sampling_gpu.cu:100-101)
__device__ __forceinline__ bool fps_skipped(float x, float y, float z) {
  float mag = (x * x) + (y * y) + (z * z);
  return (double)mag <= 1e-3;
}

// Register-resident FPS.  THREADS in {64,256} (one wave per SIMD at most); PPT points per
// thread.  Slot (thread t, i) holds the point whose tie-break priority is i*THREADS + t, so
// within a thread priorities ascend with i; the per-thread arg-max is an order-preserving
// pairwise TREE (ties keep the left = lower-priority-number operand) instead of the
// reference's serial scan -- same winner, log-depth dependency chain.  Across threads the
// winner is (max d2, then min priority).
// lds_xyz != 0: dynamic LDS holds an SoA copy of the cloud in PRIORITY order (3*slots floats),
// so the winner's coordinates are fetched by priority and the priority->index conversion
// (an integer division) happens once, after the serial loop.
// PVN3D_FPS_PROBE (tools/fps_probe.hip only): per-phase s_memtime accounting of the round.
#ifdef PVN3D_FPS_PROBE
#define FPS_PROBE_ARG , long long* __restrict__ dbg
#define FPS_PROBE_NULL , (long long*)nullptr
#define FPS_T(i) { const long long t_ = __builtin_readcyclecounter(); acc_[i] += t_ - last_; last_ = t_; }
#else
#define FPS_PROBE_ARG
#define FPS_PROBE_NULL
#define FPS_T(i)
#endif

// Nesting (see fps_nest_verify_kernel): `dmax` != nullptr -> the winning distance of every round is
// written to dmax[cloud][j] (bit pattern; < 0 = no valid point); `nest` != nullptr -> nest[cloud][nest_level]
// = R >= 1 says that the cloud is in FPS order for the first R picks: rounds 1 .. R-1 of this run are known
// to select 1 .. R-1, so they are not run (R >= m: no round at all).
struct FpsNest {
  int* dmax;           // [b][m] or nullptr
  const int* nest;     // [b][FPS_NEST_LEVELS] first round that has to be run, or nullptr
  int nest_level;
};
constexpr int FPS_NEST_LEVELS = 3;

__device__ __forceinline__ int fps_first_round(const FpsNest& nz, int m) {
  if (!nz.nest) return 1;
  const int r = nz.nest[(size_t)blockIdx.x * FPS_NEST_LEVELS + nz.nest_level];
  return r < 1 ? 1 : (r > m ? m : r);
}

template <int THREADS, int PPT, bool lds_xyz>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(int n, int m, int L, int Q,
                                                          const float* __restrict__ dataset,
                                                          int* __restrict__ idxs, FpsNest nz FPS_PROBE_ARG) {
  constexpr int NW = THREADS / 64;
  constexpr int SLOTS = THREADS * PPT;
  extern __shared__ float s_dyn[];
  __shared__ unsigned long long s_slot[2][NW];
  if (m <= 0) return;
  const int tid = threadIdx.x;
  dataset += (size_t)blockIdx.x * n * 3;
  idxs += (size_t)blockIdx.x * m;
  const int first = fps_first_round(nz, m);          // (workgroup-uniform)
  if (first >= m) {
    for (int j = tid; j < m; j += THREADS) idxs[j] = j;
    return;
  }
  int* const dmax = nz.dmax ? nz.dmax + (size_t)blockIdx.x * m : nullptr;
  const unsigned n_prio = (unsigned)Q << L;  // priorities in use: [0, bs*Q)

  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const unsigned p = (unsigned)(i * THREADS + tid);
    const int k = p < n_prio ? fps_prio_to_k(p, L, Q) : n;
    if (k < n) {
      px[i] = dataset[k * 3 + 0];
      py[i] = dataset[k * 3 + 1];
      pz[i] = dataset[k * 3 + 2];
      // skipped points never update temp and never win: temp = -inf keeps d2 = -inf (negative
      // as an integer too), below every real distance.
      tmp[i] = fps_skipped(px[i], py[i], pz[i]) ? -__builtin_inff() : 1e10f;
    } else {
      px[i] = py[i] = pz[i] = 0.f;
      tmp[i] = -__builtin_inff();
    }
    if (lds_xyz) {
      s_dyn[p] = px[i];
      s_dyn[SLOTS + p] = py[i];
      s_dyn[2 * SLOTS + p] = pz[i];
    }
  }
  const unsigned prio0 = fps_prio(0, L, Q);  // == 0
  // picks 0 .. first-1 are points 0 .. first-1 (first == 1: only the seed)
  for (int j = tid; j < first; j += THREADS) idxs[j] = lds_xyz ? (int)fps_prio(j, L, Q) : j;
  if (first > 1) {
    // the distance state those rounds would have left: the same min() chain in the same order, without the
    // per-round arg-max and exchange
    for (int q = 0; q + 1 < first; ++q) {
      const float xq = dataset[q * 3], yq = dataset[q * 3 + 1], zq = dataset[q * 3 + 2];   // (uniform)
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const float dx = px[i] - xq, dy = py[i] - yq, dz = pz[i] - zq;
        const float d = dx * dx + dy * dy + dz * dz;
        tmp[i] = __builtin_fminf(d, tmp[i]);
      }
    }
  }
  float x1 = dataset[(first - 1) * 3], y1 = dataset[(first - 1) * 3 + 1], z1 = dataset[(first - 1) * 3 + 2];
  if (NW > 1 || lds_xyz) __syncthreads();

  // In the loop only LDS traffic is ever waited for: the per-round index store stays in
  // flight (a __syncthreads() would drain it -- its release fence waits vmcnt(0), ~1 us per
  // round), so the cross-wave exchange uses a raw s_barrier behind an LDS-only wait.
#ifdef PVN3D_FPS_PROBE
  long long acc_[6] = {0, 0, 0, 0, 0, 0};
  long long last_ = __builtin_readcyclecounter();
  const long long w0_ = wall_clock64();
#endif
  for (int j = first; j < m; ++j) {
    int v[PPT], ix[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float d2 = __builtin_fminf(d, tmp[i]);
      tmp[i] = d2;
      v[i] = __float_as_int(d2);
      ix[i] = i;
    }
    FPS_T(0)
#pragma unroll
    for (int s = 1; s < PPT; s *= 2) {
#pragma unroll
      for (int i = 0; i + s < PPT; i += 2 * s) {
        const bool gt = v[i + s] > v[i];   // strict: ties keep the lower slot
        v[i] = gt ? v[i + s] : v[i];
        ix[i] = gt ? ix[i + s] : ix[i];
      }
    }
    const int best = v[0];
    FPS_T(1)
    const int wmax = wave_max_i32(best);
    const unsigned cand = (best == wmax) ? (unsigned)(ix[0] * THREADS + tid) : 0xffffffffu;
    unsigned wprio = wave_min_u32(cand);
    int gmax = wmax;
    FPS_T(2)
    if (NW > 1) {
      const int buf = j & 1;
      if ((tid & 63) == 0)
        s_slot[buf][tid >> 6] = ((unsigned long long)(unsigned)wmax << 32) | wprio;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const unsigned long long o = s_slot[buf][w];
        const int om = (int)(unsigned)(o >> 32);
        const unsigned op = (unsigned)(o & 0xffffffffu);
        const bool take = (om > gmax) || (om == gmax && op < wprio);
        gmax = take ? om : gmax;
        wprio = take ? op : wprio;
      }
    }
    // no valid point at all (every distance is the negative marker): the reference's threads
    // all keep (best=-1, besti=0) and its tree returns index 0
    unsigned win = (gmax < 0) ? prio0 : wprio;
    win = (unsigned)__builtin_amdgcn_readfirstlane((int)win);
    if (dmax && tid == 0) dmax[j] = gmax;
    FPS_T(3)
    if (lds_xyz) {
      if (tid == 0) idxs[j] = (int)win;   // priority for now; converted after the loop
      x1 = s_dyn[win];
      y1 = s_dyn[SLOTS + win];
      z1 = s_dyn[2 * SLOTS + win];
    } else {
      const int old = fps_prio_to_k(win, L, Q);
      if (tid == 0) idxs[j] = old;
      x1 = dataset[old * 3 + 0];
      y1 = dataset[old * 3 + 1];
      z1 = dataset[old * 3 + 2];
    }
#ifdef PVN3D_FPS_PROBE
    asm volatile("" :: "v"(x1), "v"(y1), "v"(z1));
    FPS_T(4)
#endif
  }
#ifdef PVN3D_FPS_PROBE
  if (tid == 0 && blockIdx.x == 0 && dbg) {
    for (int i = 0; i < 5; ++i) dbg[i] = acc_[i];
    dbg[5] = wall_clock64() - w0_;
  }
#endif
  if (lds_xyz) {
    __syncthreads();  // tid 0's stores are visible to the block after the barrier
    for (int j = tid; j < m; j += THREADS) idxs[j] = fps_prio_to_k((unsigned)idxs[j], L, Q);
  }
}

// Any-n fallback: distances in the caller's `temp` scratch (global memory), 1024 threads.
__global__ __launch_bounds__(1024) void fps_global_kernel(int n, int m, int L, int Q,
                                                          const float* __restrict__ dataset,
                                                          float* __restrict__ temp,
                                                          int* __restrict__ idxs, FpsNest nz) {
  __shared__ long long s_slot[2][16];
  if (m <= 0) return;
  const int tid = threadIdx.x;
  dataset += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * m;
  if (fps_first_round(nz, m) >= m) {      // (large clouds: all or nothing)
    for (int j = tid; j < m; j += 1024) idxs[j] = j;
    return;
  }
  int* const dmax = nz.dmax ? nz.dmax + (size_t)blockIdx.x * m : nullptr;
  for (int k = tid; k < n; k += 1024)
    temp[k] = fps_skipped(dataset[k * 3], dataset[k * 3 + 1], dataset[k * 3 + 2])
                  ? -__builtin_inff() : 1e10f;
  int old = 0;
  if (tid == 0) idxs[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
    long long best = -1LL;
    for (int k = tid; k < n; k += 1024) {
      const float dx = dataset[k * 3 + 0] - x1, dy = dataset[k * 3 + 1] - y1,
                  dz = dataset[k * 3 + 2] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      // signed key: skipped points (d2 = -inf) sort below the "none" key -1
      const long long key =
          ((long long)__float_as_int(d2) << 32) | (long long)(~fps_prio(k, L, Q));
      best = key > best ? key : best;
    }
    best = wave_max_i64(best);
    const int buf = j & 1;
    if ((tid & 63) == 0) s_slot[buf][tid >> 6] = best;
    __syncthreads();
    long long b2 = s_slot[buf][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
      long long o = s_slot[buf][w];
      b2 = o > b2 ? o : b2;
    }
    old = (b2 == -1LL) ? 0 : fps_prio_to_k(~(unsigned)(b2 & 0xffffffffLL), L, Q);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idxs[j] = old;
    if (dmax && tid == 0) dmax[j] = (b2 == -1LL) ? -1 : (int)(b2 >> 32);
  }
}

// ---- nested sampling --------------------------------------------------------------------------------
// PointNet++ samples a pyramid: level l+1 runs FPS on the m_l points level l selected, in the order it
// selected them.  FPS is greedy, so the first m' picks of a run ARE the run for m' samples: on the cloud
// S = (p_0 .. p_{n'-1}) of level-l picks, round t of the next level looks for the point of S farthest from
// {p_0 .. p_{t-1}} -- and p_t is the farthest point of the WHOLE cloud, computed with the same arithmetic.
// The next level therefore selects t in round t unless a tie is broken differently: some k > t in S with
// exactly the same distance D_t and a smaller tie-break priority under the NEXT level's block shape (it
// happens: about one 12288-point cloud in 64 has such a pair among its first 1024 picks), or a degenerate
// round (D_t <= 0).  The first such round R of every cloud and level is found here for all (k, t) at once,
// without any per-round synchronisation: thread k keeps its running minimum distance to the picks while t
// advances.  The FPS kernels of the later levels take R (FpsNest): picks below R are the identity, the
// distance state of round R is rebuilt by the same min() chain, and only rounds >= R are run -- index-exact
// in every case, and no round at all for the clouds without such a tie.
struct NestLevels {
  int n[FPS_NEST_LEVELS], m[FPS_NEST_LEVELS], L[FPS_NEST_LEVELS], Q[FPS_NEST_LEVELS];
  int count, tmax;
};

// grid (ceil(n0/64), b), block 1024: lane = candidate k of the FPS-ordered cloud `pts` (b, n0, 3), wave = one of 16
// segments of the rounds; dmax (b, n0) = the winning distances of the run that produced the order.  The running
// minimum of a candidate over the rounds is a prefix minimum, and min() is exact under any association: every wave
// first reduces its own segment of picks, then takes the minimum of the earlier segments as its start value and
// walks its segment again with the checks -- 2 x (rounds / 16) serial steps instead of `rounds` (the kernel sits
// on the critical path of a single-frame forward: 141 -> ~20 us).  dynamic LDS: tmax x (x, y, z, D).
constexpr int NEST_SEGS = 16;
__global__ __launch_bounds__(1024) void fps_nest_verify_kernel(int n0, NestLevels lv, const float* __restrict__ pts,
                                                               const int* __restrict__ dmax, int* __restrict__ first_bad) {
  extern __shared__ float4 s_pick[];      // [tmax]: pick t and (bits of) the winning distance of round t
  __shared__ float s_segmin[NEST_SEGS][64];
  const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  pts += (size_t)blockIdx.y * n0 * 3;
  dmax += (size_t)blockIdx.y * n0;
  first_bad += (size_t)blockIdx.y * FPS_NEST_LEVELS;
  for (int t = threadIdx.x; t < lv.tmax; t += 1024)
    s_pick[t] = make_float4(pts[t * 3], pts[t * 3 + 1], pts[t * 3 + 2], __int_as_float(dmax[t]));
  __syncthreads();
  const int kc = min(k, n0 - 1);
  const float x = pts[kc * 3], y = pts[kc * 3 + 1], z = pts[kc * 3 + 2];
  const bool candidate = k < n0 && !fps_skipped(x, y, z);
  // a violation needs k > t: the workgroup stops at its largest k
  const int t_end = min(lv.tmax, (int)(blockIdx.x * 64 + 63) + 1);
  // rounds 1 .. t_end-1 in NEST_SEGS segments; round t uses picks 0 .. t-1
  const int per = (t_end - 1 + NEST_SEGS - 1) / NEST_SEGS;
  const int ta = 1 + seg * per, tb = min(t_end, ta + per);
  float m = 1e10f;
  for (int t = ta; t < tb; ++t) {
    const float4 prev = s_pick[t - 1];
    const float dx = x - prev.x, dy = y - prev.y, dz = z - prev.z;
    const float d = dx * dx + dy * dy + dz * dz;
    m = __builtin_fminf(d, m);
  }
  s_segmin[seg][lane] = m;
  __syncthreads();
  float run = 1e10f;
  for (int s = 0; s < seg; ++s) run = __builtin_fminf(s_segmin[s][lane], run);
  unsigned pk[FPS_NEST_LEVELS];
#pragma unroll
  for (int l = 0; l < FPS_NEST_LEVELS; ++l) pk[l] = l < lv.count ? fps_prio(kc, lv.L[l], lv.Q[l]) : 0u;
  int bad[FPS_NEST_LEVELS];
#pragma unroll
  for (int l = 0; l < FPS_NEST_LEVELS; ++l) bad[l] = 0x7fffffff;
  for (int t = ta; t < tb; ++t) {
    const float4 prev = s_pick[t - 1];
    const int dt = __float_as_int(s_pick[t].w);
    const float dx = x - prev.x, dy = y - prev.y, dz = z - prev.z;
    const float d = dx * dx + dy * dy + dz * dz;
    run = __builtin_fminf(d, run);
    const int rb = __float_as_int(run);
    if (dt <= 0 || (candidate && k > t && rb >= dt)) {      // (rare: the priorities are only computed here)
#pragma unroll
      for (int l = 0; l < FPS_NEST_LEVELS; ++l) {
        if (l < lv.count && t < lv.m[l]) {
          const bool tie_lost = rb == dt && pk[l] < fps_prio(t, lv.L[l], lv.Q[l]);
          if (dt <= 0 || (k < lv.n[l] && (rb > dt || tie_lost))) bad[l] = min(bad[l], t);
        }
      }
    }
  }
#pragma unroll
  for (int l = 0; l < FPS_NEST_LEVELS; ++l)
    if (bad[l] != 0x7fffffff) atomicMin(&first_bad[l], bad[l]);
}

// A level whose predecessor did not come out as the identity samples a different cloud: it runs in full.
__global__ void fps_nest_finalize_kernel(int b, NestLevels lv, int* __restrict__ first_bad) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= b) return;
  int* f = first_bad + (size_t)c * FPS_NEST_LEVELS;
  bool prefix = true;
  for (int l = 0; l < FPS_NEST_LEVELS; ++l) {
    if (l >= lv.count || !prefix) { f[l] = 1; continue; }
    if (f[l] < lv.m[l]) prefix = false;
  }
}

template <int THREADS, int PPT>
int launch_fps_reg(int b, int n, int m, int L, int Q, const float* dataset, int* idxs, const FpsNest& nz,
                   hipStream_t st) {
  size_t lds = (size_t)THREADS * PPT * 3 * sizeof(float);
  int use_lds = lds + 1024 <= 160 * 1024;
  if (!use_lds) lds = 0;
  if (use_lds) {
    auto kern = fps_reg_kernel<THREADS, PPT, true>;
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(kern));
    hipLaunchKernelGGL(kern, dim3(b), dim3(THREADS), lds, st, n, m, L, Q, dataset,
                       idxs, nz FPS_PROBE_NULL);
  } else {
    hipLaunchKernelGGL((fps_reg_kernel<THREADS, PPT, false>), dim3(b), dim3(THREADS), 0, st, n, m,
                       L, Q, dataset, idxs, nz FPS_PROBE_NULL);
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}

__global__ void gather_points_kernel(int c, int n, int m, const float* __restrict__ points,
                                     const int* __restrict__ idx, float* __restrict__ out) {
  // grid: (ceil(m/256), c, b)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y, i = blockIdx.z;
  if (j < m) {
    const int a = idx[(size_t)i * m + j];
    out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
  }
}

__global__ void gather_points_grad_kernel(int c, int n, int m,
                                          const float* __restrict__ grad_out,
                                          const int* __restrict__ idx,
                                          float* __restrict__ grad_points) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y, i = blockIdx.z;
  if (j < m) {
    const int a = idx[(size_t)i * m + j];
    atomicAdd(grad_points + ((size_t)i * c + l) * n + a, grad_out[((size_t)i * c + l) * m + j]);
  }
}

}  // namespace

extern "C" int pvn3d_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

extern "C" int pvn3d_abi_version(void) { return PVN3D_ABI_VERSION; }

static void fps_block_shape(int n, int* L, int* Q) {
  const int bs = pvn3d_opt_n_threads(n);
  *L = 0;
  while ((1 << *L) < bs) ++*L;
  *Q = (n + bs - 1) / bs;
}

// fps_cells.hip: exact spatial culling, one to three waves per cloud (64 < n <= 12288, needs a workspace)
int pvn3d_fps_cells_ws_words(int n);
int pvn3d_fps_cells_launch(int b, int n, int m, int L, int Q, const float* dataset, int* ws, int* idxs, int* dmax,
                           int waves, hipStream_t st);
// below this size the register-resident kernel (one round = a few hundred instructions) is at least as fast
constexpr int FPS_CELLS_MIN_N = 4097;

static int fps_launch(int b, int n, int m, const float* dataset, float* temp, int* cells_ws, int* idxs,
                      const FpsNest& nz, hipStream_t st, int waves = 0) {
  int L, Q;
  fps_block_shape(n, &L, &Q);
  const int slots = (1 << L) * Q;  // priority slots to cover (>= n)
  // the culled kernel runs whole runs only (no nest.first > 1 restart)
  if (cells_ws && !nz.nest && n >= FPS_CELLS_MIN_N && pvn3d_fps_cells_ws_words(n) > 0)
    return pvn3d_fps_cells_launch(b, n, m, L, Q, dataset, cells_ws, idxs, nz.dmax, waves, st);
  if (slots <= 64) return launch_fps_reg<64, 1>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 128) return launch_fps_reg<64, 2>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 256) return launch_fps_reg<64, 4>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 512) return launch_fps_reg<64, 8>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 1024) return launch_fps_reg<256, 4>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 2048) return launch_fps_reg<256, 8>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 4096) return launch_fps_reg<256, 16>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 8192) return launch_fps_reg<256, 32>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 12288) return launch_fps_reg<256, 48>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (slots <= 16384) return launch_fps_reg<256, 64>(b, n, m, L, Q, dataset, idxs, nz, st);
  if (!temp) return (int)hipErrorInvalidValue;  // large clouds need the caller's scratch
  hipLaunchKernelGGL(fps_global_kernel, dim3(b), dim3(1024), 0, st, n, m, L, Q, dataset, temp,
                     idxs, nz);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_furthest_point_sampling(int b, int n, int m, const float* dataset,
                                             float* temp, int* idxs, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (n <= 0 || !dataset || !idxs) return (int)hipErrorInvalidValue;
  return fps_launch(b, n, m, dataset, temp, nullptr, idxs, FpsNest{nullptr, nullptr, 0}, (hipStream_t)stream);
}

extern "C" int pvn3d_fps_ws_words(int n) {
  const int w = n >= FPS_CELLS_MIN_N ? pvn3d_fps_cells_ws_words(n) : 0;
  return w > 0 ? w : (n > 16384 ? n : 0);
}

static int fps_ws_entry(int b, int n, int m, const float* dataset, void* ws, int* idxs, int* dmax_out,
                        const int* nest_flags, int nest_level, int waves, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (n <= 0 || !dataset || !idxs || nest_level < 0 || nest_level >= FPS_NEST_LEVELS || waves < 0)
    return (int)hipErrorInvalidValue;
  if (pvn3d_fps_ws_words(n) > 0 && !ws) return (int)hipErrorInvalidValue;
  const bool cells = n >= FPS_CELLS_MIN_N && pvn3d_fps_cells_ws_words(n) > 0;
  return fps_launch(b, n, m, dataset, cells ? nullptr : (float*)ws, cells ? (int*)ws : nullptr, idxs,
                    FpsNest{dmax_out, nest_flags, nest_level}, (hipStream_t)stream, waves);
}
extern "C" int pvn3d_furthest_point_sampling_ws(int b, int n, int m, const float* dataset, void* ws, int* idxs,
                                                int* dmax_out, const int* nest_flags, int nest_level,
                                                void* stream) {
  return fps_ws_entry(b, n, m, dataset, ws, idxs, dmax_out, nest_flags, nest_level, 0, stream);
}
extern "C" int pvn3d_furthest_point_sampling_ws_waves(int b, int n, int m, const float* dataset, void* ws, int* idxs,
                                                      int* dmax_out, const int* nest_flags, int nest_level,
                                                      int waves_per_cloud, void* stream) {
  return fps_ws_entry(b, n, m, dataset, ws, idxs, dmax_out, nest_flags, nest_level, waves_per_cloud, stream);
}

extern "C" int pvn3d_furthest_point_sampling_nested(int b, int n, int m, const float* dataset, float* temp,
                                                    int* idxs, int* dmax_out, const int* nest_flags,
                                                    int nest_level, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (n <= 0 || !dataset || !idxs || nest_level < 0 || nest_level >= FPS_NEST_LEVELS)
    return (int)hipErrorInvalidValue;
  return fps_launch(b, n, m, dataset, temp, nullptr, idxs, FpsNest{dmax_out, nest_flags, nest_level},
                    (hipStream_t)stream);
}

extern "C" int pvn3d_fps_nest_verify(int b, int n0, int n_levels, const int* m_levels, const float* ordered_xyz,
                                     const int* dmax, int* flags, void* stream) {
  if (b <= 0) return 0;
  if (n0 <= 0 || n_levels < 1 || n_levels > FPS_NEST_LEVELS || !m_levels || !ordered_xyz || !dmax || !flags)
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  NestLevels lv;
  lv.count = n_levels;
  int n = n0;
  for (int l = 0; l < FPS_NEST_LEVELS; ++l) {
    lv.n[l] = lv.m[l] = lv.L[l] = 0;
    lv.Q[l] = 1;
    if (l < n_levels) {
      if (m_levels[l] <= 0 || m_levels[l] > n) return (int)hipErrorInvalidValue;
      lv.n[l] = n;
      lv.m[l] = m_levels[l];
      fps_block_shape(n, &lv.L[l], &lv.Q[l]);
      n = m_levels[l];
    }
  }
  lv.tmax = 0;
  for (int l = 0; l < n_levels; ++l) lv.tmax = lv.m[l] > lv.tmax ? lv.m[l] : lv.tmax;
  if ((size_t)lv.tmax * sizeof(float4) > 128 * 1024) return (int)hipErrorInvalidValue;
  // 0x7f7f7f7f = "no round has to be run"
  pvn3d_fill_u32(flags, 0x7f7f7f7fu, (size_t)b * FPS_NEST_LEVELS, st);
  PVN3D_LAUNCH_CHECK();
  PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(fps_nest_verify_kernel));
  hipLaunchKernelGGL(fps_nest_verify_kernel, dim3(pvn3d_ceil_div(n0, 64), b), dim3(1024),
                     (size_t)lv.tmax * sizeof(float4), st, n0, lv, ordered_xyz, dmax, flags);
  PVN3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(fps_nest_finalize_kernel, dim3(pvn3d_ceil_div(b, 64)), dim3(64), 0, st, b, lv, flags);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_gather_points(int b, int c, int n, int npoints, const float* points,
                                   const int* idx, float* out, void* stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  hipLaunchKernelGGL(gather_points_kernel, dim3(pvn3d_ceil_div(npoints, 256), c, b), dim3(256),
                     0, (hipStream_t)stream, c, n, npoints, points, idx, out);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_gather_points_grad(int b, int c, int n, int npoints,
                                        const float* grad_out, const int* idx,
                                        float* grad_points, void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, st));
  if (npoints <= 0) return 0;
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(pvn3d_ceil_div(npoints, 256), c, b),
                     dim3(256), 0, st, c, n, npoints, grad_out, idx, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
