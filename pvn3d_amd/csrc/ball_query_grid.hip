// ball_query_grid.hip -- exact ball query through a toroidal uniform grid, for large clouds.
//
// Same results, bit for bit, as the brute-force scan in ball_query.hip (and therefore as
// query_ball_point_kernel, pvn3d/_ext-src/src/ball_query_gpu.cu:9-44): for every centre the
// first `nsample` indices k in ascending order with d2 < r^2, padded with the first hit.
// Only the set of (centre, point) pairs whose distance is EVALUATED shrinks:
//
//   build  (one workgroup per cloud): cell = floor(p / h) per axis with h = 1.001 * r_max,
//          bucket = (cx & 31) | (cy & 31) << 5 | (cz & 31) << 10   (32^3 toroidal grid);
//          LDS histogram -> exclusive scan -> scatter of (x, y, z, k) into bucket order.
//          The 27 neighbour cells of any cell map to 27 DISTINCT buckets (32 >= 3), and any
//          point within r_max of a centre lies in one of the centre's 27 neighbour cells, so
//          the candidate set is a superset of the ball; aliased far cells only add candidates
//          that the distance test rejects.
//   query  (one wave per centre): the 27 bucket ranges are flattened, lanes evaluate
//          candidates with EXACTLY the brute-force arithmetic
//          (d2 = ((dx*dx + dy*dy) + dz*dz), dx = centre - point, -ffp-contract=off), and a hit
//          sets bit k of a per-wave LDS bitmap over the index space.  A hit's rank in ascending k is
//          the number of set bits below bit k (prefix popcounts) -- no sort, any hit count.
//
// With the PVN3D level-0 shapes (n = 12288, m = 2048) ~100 candidates per centre are examined
// instead of 12288.  Scratch (bucket offsets + the bucket-ordered copy of the cloud) is
// passed in by the caller; nothing is allocated here.
#include "common.h"

namespace {

// toroidal extent per axis is 2^AL: 32 (32768 buckets) for large clouds, 16 (4096 buckets) for
// small ones, where zero-filling and scanning the bucket table would dominate the build
constexpr int GRID_T_MAX = 32 * 32 * 32;
constexpr int BQG_MAX_N = 32768;                // bitmap capacity (bits) per wave
__host__ __device__ constexpr int grid_t(int AL) { return 1 << (3 * AL); }

struct GridWs {
  int* cell_start;    // [b][T + 1]
  float4* sorted;     // [b][n]  (x, y, z, bits(k))
};

inline int grid_al_for(int n) { return n >= 8192 ? 5 : 4; }

inline size_t grid_ws_layout(int b, int n, char* base, GridWs* ws) {
  const size_t cs = ((size_t)b * (GRID_T_MAX + 1) * sizeof(int) + 255) / 256 * 256;
  const size_t so = (size_t)b * n * sizeof(float4);
  if (ws) {
    ws->cell_start = (int*)base;
    ws->sorted = (float4*)(base + cs);
  }
  return cs + so;
}

template <int AL>
__device__ __forceinline__ int grid_bucket_c(int cx, int cy, int cz) {
  constexpr int M = (1 << AL) - 1;
  return (cx & M) | ((cy & M) << AL) | ((cz & M) << (2 * AL));
}

// Cell coordinates are taken relative to the cloud's first point: floor((p - origin) / h) is then exact
// to ~6e-8 * (extent / h) cells, far inside the 0.1 % margin of h = 1.001 r for any cloud whose extent
// is below ~10^4 cells (with absolute coordinates a cloud given in millimetres, or far from the
// origin, could put a point with d2 < r^2 outside the 27 cells that are searched).
template <int AL>
__device__ __forceinline__ int grid_bucket(float x, float y, float z, float ox, float oy, float oz, float inv_h) {
  return grid_bucket_c<AL>((int)floorf((x - ox) * inv_h), (int)floorf((y - oy) * inv_h), (int)floorf((z - oz) * inv_h));
}

// LDS index of bucket c, padded by one word per 32 so that a thread scanning its own 32 (or 4)
// consecutive buckets does not fight its neighbours for one bank
__device__ __forceinline__ int pad32(int c) { return c + (c >> 5); }

// one workgroup (1024 threads) per cloud; dynamic LDS = padded bucket table (132 KiB at AL=5)
// PPT > 0: n <= 1024 * PPT and every thread keeps its PPT points (and their buckets) in registers between
// the histogram and the scatter pass -- all loads of a pass are then in flight together (with a
// run-time trip count each of the two passes was a chain of n/1024 dependent global round trips:
// 29 us for n = 12288); PPT == 0: any n, two passes over global memory.
// VEC (PPT % 4 == 0, n % 4 == 0, 16-byte aligned clouds): a thread owns groups of four consecutive points
// and reads each group with three 16-byte loads (the 4-byte loads at a 12-byte stride touch every cache
// line three times).
template <int AL, int PPT, bool VEC>
__global__ __launch_bounds__(1024) void grid_build_kernel(int n, float inv_h,
                                                          const float* __restrict__ xyz,
                                                          int* __restrict__ cell_start,
                                                          float4* __restrict__ sorted) {
  constexpr int T = grid_t(AL);
  constexpr int PER = T / 1024;
  extern __shared__ int s_cnt[];  // [pad32(T)]
  __shared__ int s_part[1024];
  const int tid = threadIdx.x;
  xyz += (size_t)blockIdx.x * n * 3;
  cell_start += (size_t)blockIdx.x * (T + 1);
  sorted += (size_t)blockIdx.x * n;
  for (int i = tid; i < T + (T >> 5); i += 1024) s_cnt[i] = 0;
  const float ox = xyz[0], oy = xyz[1], oz = xyz[2];
  __syncthreads();
  float px[PPT > 0 ? PPT : 1], py[PPT > 0 ? PPT : 1], pz[PPT > 0 ? PPT : 1];
  int pb[PPT > 0 ? PPT : 1];
  auto point_of = [&](int i) { return VEC ? 4 * (tid + 1024 * (i >> 2)) + (i & 3) : tid + 1024 * i; };
  if (PPT > 0) {
    if constexpr (VEC) {
#pragma unroll
      for (int g = 0; g < PPT / 4; ++g) {
        const int k0 = min(4 * (tid + 1024 * g), n - 4);
        const float4* q = reinterpret_cast<const float4*>(xyz + (size_t)k0 * 3);
        const float4 a = q[0], b = q[1], c = q[2];
        px[4 * g] = a.x; py[4 * g] = a.y; pz[4 * g] = a.z;
        px[4 * g + 1] = a.w; py[4 * g + 1] = b.x; pz[4 * g + 1] = b.y;
        px[4 * g + 2] = b.z; py[4 * g + 2] = b.w; pz[4 * g + 2] = c.x;
        px[4 * g + 3] = c.y; py[4 * g + 3] = c.z; pz[4 * g + 3] = c.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PPT; ++i) {
        const int k = min(tid + 1024 * i, n - 1);
        px[i] = xyz[k * 3]; py[i] = xyz[k * 3 + 1]; pz[i] = xyz[k * 3 + 2];
      }
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      pb[i] = pad32(grid_bucket<AL>(px[i], py[i], pz[i], ox, oy, oz, inv_h));
      if (point_of(i) < n) atomicAdd(&s_cnt[pb[i]], 1);
    }
  } else {
    for (int k = tid; k < n; k += 1024)
      atomicAdd(&s_cnt[pad32(grid_bucket<AL>(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2], ox, oy, oz, inv_h))], 1);
  }
  __syncthreads();
  // exclusive scan: each thread owns PER consecutive buckets
  int local = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) local += s_cnt[pad32(tid * PER + i)];
  // block scan of the 1024 partials: wave scan, then the 16 wave totals
  const int lane = tid & 63, wv = tid >> 6;
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_part[wv] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wv; ++w) wave_off += s_part[w];
  int run = wave_off + incl - local;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = s_cnt[pad32(tid * PER + i)];
    s_cnt[pad32(tid * PER + i)] = run;  // becomes the scatter cursor
    run += c;
  }
  if (tid == 1023) cell_start[T] = run;
  __syncthreads();
  // the table leaves in bucket order, one 256-byte burst per wave and store (written from the scan loop,
  // bucket tid * PER + i, every store instruction touched 64 different cache lines: 14 of the kernel's 22 us)
#pragma unroll
  for (int i = 0; i < PER; ++i) cell_start[tid + 1024 * i] = s_cnt[pad32(tid + 1024 * i)];
  __syncthreads();
  if (PPT > 0) {
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int k = point_of(i);
      if (k < n) {
        const int pos = atomicAdd(&s_cnt[pb[i]], 1);
        sorted[pos] = make_float4(px[i], py[i], pz[i], __int_as_float(k));
      }
    }
  } else {
    for (int k = tid; k < n; k += 1024) {
      const float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
      const int pos = atomicAdd(&s_cnt[pad32(grid_bucket<AL>(x, y, z, ox, oy, oz, inv_h))], 1);
      sorted[pos] = make_float4(x, y, z, __int_as_float(k));
    }
  }
}

// ---- query ---------------------------------------------------------------------------------------------
// One wave per centre at a time.  Per centre: (1) lanes 0..26 fetch the 27 bucket ranges (one centre ahead),
// a wave scan flattens them; (2) distance pass: candidates 128 at a time (two per lane, both loads in flight
// together) with EXACTLY the brute-force arithmetic, a hit sets bit k of the wave's LDS bitmap over the index
// space and the first 128 candidates are parked as (k | flags); (3) rank pass: every lane counts the bits of
// its WPL consecutive bitmap words and one wave scan turns that into pre[w] = number of hits below word w; the
// candidates are walked again (parked ones from LDS, the rest evaluated again) and a hit writes itself to
// slot pre[k >> 5] + popcount(bm[k >> 5] & bits below k) = its rank in ascending index order -- no sort, no
// per-lane loop, any hit count.
// The kernel is latency-bound (a chain of dependent LDS and L2 round trips per centre, LDS-limited occupancy),
// which is why the batches are paired; measured alternatives: pulling the bits out of the bitmap with per-lane
// loops (round 1: ~250 mostly scalar exec-mask instructions per bitmap, 170 us at level 0), half a wave per
// centre (fewer instructions, twice the LDS per wave: 120 us against 78 us for the unpaired full wave).
constexpr int BQG_CAND = 128;         // parked candidates per centre (one pair of batches)
constexpr int BQG_STAGE = 64;         // rows up to this length are assembled in LDS and leave coalesced
constexpr int BQG_FLAG_A = 1 << 30, BQG_FLAG_B = 1 << 31, BQG_KMASK = BQG_FLAG_A - 1;

// exclusive prefix sum over the wave on the DPP path (row shifts, then row_bcast 15 / 31): six VALU
// instructions, no LDS traffic (a __shfl_up ladder is six dependent ds_bpermute round trips, and a
// centre needs three scans)
__device__ __forceinline__ int wave_excl_scan_add(int v, int* total) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);   // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);   // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);   // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);   // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  *total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}

// Candidates f and f + 64 of the centre (total >= 1): owner cell q = largest q with first[q] <= f (27
// non-decreasing entries, padded to 32 with INT_MAX), then the brute-force distance test(s).  Both searches
// and both loads are independent, so their LDS / L2 round trips overlap; lanes past the end evaluate the last
// candidate and drop the result.  Returns k | flags (0 past the end).
template <bool PAIR>
__device__ __forceinline__ void eval_pair(int f, int total, const int2* __restrict__ cell,
                                          const float4* __restrict__ sorted, float cx, float cy, float cz,
                                          float r2a, float r2b, int (&parked)[2]) {
  int fc[2], q[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) { fc[x] = min(f + 64 * x, total - 1); q[x] = 0; }
#pragma unroll
  for (int s = 16; s >= 1; s >>= 1) {
#pragma unroll
    for (int x = 0; x < 2; ++x) q[x] = (cell[q[x] + s].x <= fc[x]) ? q[x] + s : q[x];
  }
  float4 p[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int2 cb = cell[q[x]];
    p[x] = sorted[cb.y + (fc[x] - cb.x)];
  }
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const float dx = cx - p[x].x, dy = cy - p[x].y, dz = cz - p[x].z;
    const float d2 = dx * dx + dy * dy + dz * dz;
    const int c = __float_as_int(p[x].w) | (d2 < r2a ? BQG_FLAG_A : 0) | ((PAIR && d2 < r2b) ? BQG_FLAG_B : 0);
    parked[x] = (f + 64 * x < total) ? c : 0;
  }
}

// (the prefix counts are stored packed two per word and read back as 16-bit values, the bitmap is read in
// 8- / 16-byte pieces: may_alias types, so that type-based alias analysis never reorders those accesses)
typedef unsigned short __attribute__((may_alias)) bq_u16;
typedef unsigned __attribute__((may_alias)) bq_u32;
typedef unsigned bq_u32x4 __attribute__((ext_vector_type(4), may_alias));
typedef unsigned bq_u32x2 __attribute__((ext_vector_type(2), may_alias));

template <int WPL>
__device__ __forceinline__ int rank_prefix(const unsigned* bm, bq_u16* pre, int lane) {
  const int w0 = lane * WPL;
  unsigned w[WPL];
  if constexpr (WPL % 4 == 0) {
#pragma unroll
    for (int g = 0; g < WPL / 4; ++g) {
      const bq_u32x4 v = reinterpret_cast<const bq_u32x4*>(bm + w0)[g];
      w[4 * g] = v.x; w[4 * g + 1] = v.y; w[4 * g + 2] = v.z; w[4 * g + 3] = v.w;
    }
  } else if constexpr (WPL % 2 == 0) {
#pragma unroll
    for (int g = 0; g < WPL / 2; ++g) {
      const bq_u32x2 v = reinterpret_cast<const bq_u32x2*>(bm + w0)[g];
      w[2 * g] = v.x; w[2 * g + 1] = v.y;
    }
  } else {
#pragma unroll
    for (int i = 0; i < WPL; ++i) w[i] = bm[w0 + i];
  }
  int below[WPL];
  int run = 0;
#pragma unroll
  for (int i = 0; i < WPL; ++i) {
    below[i] = run;
    run += __builtin_popcount(w[i]);
  }
  int total;
  const int base = wave_excl_scan_add(run, &total);
  if constexpr (WPL % 2 == 0) {        // counts <= n <= 32768 fit 16 bits
#pragma unroll
    for (int g = 0; g < WPL / 2; ++g)
      reinterpret_cast<bq_u32*>(pre + w0)[g] = (unsigned)(base + below[2 * g]) | ((unsigned)(base + below[2 * g + 1]) << 16);
  } else {
#pragma unroll
    for (int i = 0; i < WPL; ++i) pre[w0 + i] = (unsigned short)(base + below[i]);
  }
  return total;
}

// slots [0, min(total, nsample)) hold the hits; pad with the first hit (slot BQG_STAGE), no hit -> zeros
__device__ __forceinline__ void emit_row(const int* __restrict__ stage, int total, int nsample,
                                         int* __restrict__ out, int lane) {
  const int filled = total < nsample ? total : nsample;
  if (nsample <= BQG_STAGE) {
    if (lane < nsample) out[lane] = total ? stage[lane < filled ? lane : BQG_STAGE] : 0;
  } else {      // the hits went straight to `out`
    const int first = total ? stage[BQG_STAGE] : 0;
    for (int l = filled + lane; l < nsample; l += 64) out[l] = first;
  }
}

// 4 waves per workgroup, each wave works through centres j = blockIdx.x*4 + wave, + 4*gridDim.x, ... (its
// bitmaps are cleared once: every centre leaves them clean).  WPL = bitmap words per lane (64 * WPL * 32 >= n).
// dynamic LDS: per wave bitmaps [NB][64*WPL] u32, then (after all waves) prefix counts [NB][64*WPL] u16.
template <bool PAIR, int AL, int WPL>
__global__ __launch_bounds__(256) void ball_query_grid_kernel(
    int n, int m, float inv_h, float r2a, int nsa, float r2b, int nsb,
    const float* __restrict__ new_xyz_all, const float* __restrict__ xyz_all, const int* __restrict__ cell_start,
    const float4* __restrict__ sorted, int* __restrict__ idxa, int* __restrict__ idxb) {
  constexpr int WORDS = 64 * WPL;
  constexpr int NB = PAIR ? 2 : 1;
  extern __shared__ __align__(16) unsigned s_bm[];
  __shared__ int2 s_cell[4][32];      // {first flat candidate of the cell, its offset in `sorted`}
  __shared__ int s_cand[4][BQG_CAND];
  __shared__ int s_stage[4][NB][BQG_STAGE + 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // every workgroup of a cloud on one XCD: `sorted` / `cell_start` of the cloud are then fetched from HBM once, not
  // once per XCD (PMC: 3.3x the algorithmic traffic with the plain mapping)
  int bi, bx;
  pvn3d_xcd_frame_map(bi, bx);
  unsigned* const bma = s_bm + (size_t)wave * NB * WORDS;
  unsigned* const bmb = bma + (NB - 1) * WORDS;
  bq_u16* const prea = reinterpret_cast<bq_u16*>(s_bm + (size_t)4 * NB * WORDS) + (size_t)wave * NB * WORDS;
  bq_u16* const preb = prea + (NB - 1) * WORDS;
  const int2* const cell = s_cell[wave];
  int* const cand = s_cand[wave];
  int* const stga = s_stage[wave][0];
  int* const stgb = s_stage[wave][NB - 1];
#pragma unroll
  for (int i = 0; i < NB * WPL; ++i) bma[i * 64 + lane] = 0u;
  cell_start += (size_t)bi * (grid_t(AL) + 1);
  sorted += (size_t)bi * n;
  const float ox = xyz_all[(size_t)bi * n * 3], oy = xyz_all[(size_t)bi * n * 3 + 1], oz = xyz_all[(size_t)bi * n * 3 + 2];
  const int dcx = lane % 3 - 1, dcy = (lane / 3) % 3 - 1, dcz = lane / 9 - 1;   // lanes 0..26: one neighbour cell each
  // the centre's coordinates and its 27 bucket ranges are fetched one centre ahead (two dependent
  // global round trips that would otherwise head every centre's latency chain)
  auto fetch_cells = [&](int j, float& cx, float& cy, float& cz, int& beg, int& cnt) {
    const float* c = new_xyz_all + ((size_t)bi * m + min(j, m - 1)) * 3;
    cx = c[0]; cy = c[1]; cz = c[2];
    const int gx = (int)floorf((cx - ox) * inv_h), gy = (int)floorf((cy - oy) * inv_h), gz = (int)floorf((cz - oz) * inv_h);
    beg = 0; cnt = 0;
    if (lane < 27) {
      const int bucket = grid_bucket_c<AL>(gx + dcx, gy + dcy, gz + dcz);
      beg = cell_start[bucket];
      cnt = cell_start[bucket + 1] - beg;
    }
  };
  const int stride = 4 * gridDim.x;
  float ncx, ncy, ncz;
  int nbeg, ncnt;
  fetch_cells(bx * 4 + wave, ncx, ncy, ncz, nbeg, ncnt);
  for (int j = bx * 4 + wave; j < m; j += stride) {
    const float cx = ncx, cy = ncy, cz = ncz;
    const int beg = nbeg, cnt = ncnt;
    fetch_cells(j + stride, ncx, ncy, ncz, nbeg, ncnt);
    int total;
    const int pref = wave_excl_scan_add(cnt, &total);
    if (lane < 32) s_cell[wave][lane] = make_int2(lane < 27 ? pref : 0x7fffffff, beg);
    __builtin_amdgcn_wave_barrier();   // (single wave: its LDS operations complete in program order)
    // ---- distance pass
    for (int f0 = 0; f0 < total; f0 += 128) {
      int parked[2];
      eval_pair<PAIR>(f0 + lane, total, cell, sorted, cx, cy, cz, r2a, r2b, parked);
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int k = parked[x] & BQG_KMASK;
        if (parked[x] & BQG_FLAG_A) atomicOr(&bma[k >> 5], 1u << (k & 31));
        if (PAIR && (parked[x] & BQG_FLAG_B)) atomicOr(&bmb[k >> 5], 1u << (k & 31));
      }
      if (f0 == 0) { cand[lane] = parked[0]; cand[64 + lane] = parked[1]; }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- rank pass
    const int ha_total = rank_prefix<WPL>(bma, prea, lane);
    const int hb_total = PAIR ? rank_prefix<WPL>(bmb, preb, lane) : 0;
    __builtin_amdgcn_wave_barrier();
    int* const outa = idxa + ((size_t)bi * m + j) * nsa;
    int* const outb = PAIR ? idxb + ((size_t)bi * m + j) * nsb : nullptr;
    int* const da = nsa <= BQG_STAGE ? stga : outa;
    int* const db = nsb <= BQG_STAGE ? stgb : outb;
    for (int f0 = 0; f0 < total; f0 += 128) {
      int c[2];
      if (f0 == 0) { c[0] = cand[lane]; c[1] = cand[64 + lane]; }
      else eval_pair<PAIR>(f0 + lane, total, cell, sorted, cx, cy, cz, r2a, r2b, c);
      int k[2], ra[2], rb[2];
#pragma unroll
      for (int x = 0; x < 2; ++x) {     // (all reads first: the two candidates' LDS round trips overlap)
        k[x] = c[x] & BQG_KMASK;
        const int wd = k[x] >> 5;
        const unsigned lower = (1u << (k[x] & 31)) - 1u;
        ra[x] = (c[x] & BQG_FLAG_A) ? (int)prea[wd] + __builtin_popcount(bma[wd] & lower) : 0x7fffffff;
        rb[x] = (PAIR && (c[x] & BQG_FLAG_B)) ? (int)preb[wd] + __builtin_popcount(bmb[wd] & lower) : 0x7fffffff;
      }
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        if (ra[x] < nsa) da[ra[x]] = k[x];
        if (ra[x] == 0) stga[BQG_STAGE] = k[x];
        if (PAIR) {
          if (rb[x] < nsb) db[rb[x]] = k[x];
          if (rb[x] == 0) stgb[BQG_STAGE] = k[x];
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < NB * WPL; ++i) bma[i * 64 + lane] = 0u;   // leave the bitmaps clean for the next centre
    emit_row(stga, ha_total, nsa, outa, lane);
    if (PAIR) emit_row(stgb, hb_total, nsb, outb, lane);
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace

extern "C" size_t pvn3d_ball_query_grid_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  return grid_ws_layout(b, n, nullptr, nullptr);
}

extern "C" int pvn3d_ball_query_pair_grid(int b, int n, int m, float radius0, int nsample0,
                                          float radius1, int nsample1, const float* new_xyz,
                                          const float* xyz, int* idx0, int* idx1,
                                          void* workspace, size_t workspace_bytes,
                                          void* stream) {
  if (b <= 0 || m <= 0) return 0;
  const bool pair = nsample1 > 0;
  if (nsample0 <= 0 || n <= 0 || n > BQG_MAX_N || !new_xyz || !xyz || !idx0 || (pair && !idx1) ||
      !workspace || !(radius0 > 0.f) || (pair && !(radius1 > 0.f)))
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < pvn3d_ball_query_grid_workspace_bytes(b, n))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  GridWs ws;
  grid_ws_layout(b, n, (char*)workspace, &ws);
  const float rmax = pair ? fmaxf(radius0, radius1) : radius0;
  const float inv_h = 1.0f / (rmax * 1.001f);
  // bitmap words per lane: 64 * wpl * 32 bits >= n, rounded up to an instantiated size
  const int wpl_need = (n + 2047) / 2048;
  const int al = grid_al_for(n);
  const int wpl = al == 5 ? (wpl_need <= 4 ? 4 : wpl_need <= 6 ? 6 : wpl_need <= 8 ? 8 : 16)
                          : (wpl_need <= 1 ? 1 : wpl_need <= 2 ? 2 : 4);
  const size_t qlds = (size_t)4 * (pair ? 2 : 1) * 64 * wpl * (sizeof(unsigned) + sizeof(unsigned short));
  // enough workgroups to fill the chip a few times over, then several centres per wave
  int qx = pvn3d_ceil_div(m, 4);
  while (qx > 8 && (long long)qx * b > 8192) qx = (qx + 1) / 2;
  const dim3 qgrid(qx, b);
  const float r2a = radius0 * radius0, r2b = radius1 * radius1;
#define BQG_QUERY(AL, WPL)                                                                        \
  do {                                                                                            \
    if (pair) {                                                                                   \
      auto qk = ball_query_grid_kernel<true, AL, WPL>;                                            \
      if (qlds > 48 * 1024) PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(qk));             \
      hipLaunchKernelGGL(qk, qgrid, dim3(256), qlds, st, n, m, inv_h, r2a, nsample0, r2b,         \
                         nsample1, new_xyz, xyz, ws.cell_start, ws.sorted, idx0, idx1);           \
    } else {                                                                                      \
      auto qk = ball_query_grid_kernel<false, AL, WPL>;                                           \
      if (qlds > 48 * 1024) PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(qk));             \
      hipLaunchKernelGGL(qk, qgrid, dim3(256), qlds, st, n, m, inv_h, r2a, nsample0, 0.f, 0,      \
                         new_xyz, xyz, ws.cell_start, ws.sorted, idx0, nullptr);                  \
    }                                                                                             \
  } while (0)
#define BQG_BUILD(AL, PPT, VEC)                                                                   \
  do {                                                                                            \
    auto bk = grid_build_kernel<AL, PPT, VEC>;                                                    \
    const size_t blds = (size_t)(grid_t(AL) + (grid_t(AL) >> 5)) * sizeof(int);                   \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(bk));                                     \
    hipLaunchKernelGGL(bk, dim3(b), dim3(1024), blds, st, n, inv_h, xyz, ws.cell_start,           \
                       ws.sorted);                                                                \
    PVN3D_LAUNCH_CHECK();                                                                         \
  } while (0)
  if (al == 5) {
    const bool vec = (n % 4 == 0) && (((uintptr_t)xyz & 15) == 0);
    if (n <= 12288 && vec) BQG_BUILD(5, 12, true); else if (n <= 12288) BQG_BUILD(5, 12, false); else BQG_BUILD(5, 0, false);
    if (wpl == 4) BQG_QUERY(5, 4); else if (wpl == 6) BQG_QUERY(5, 6); else if (wpl == 8) BQG_QUERY(5, 8); else BQG_QUERY(5, 16);
  } else {
    if (n <= 2048) BQG_BUILD(4, 2, false); else BQG_BUILD(4, 8, false);
    if (wpl == 1) BQG_QUERY(4, 1); else if (wpl == 2) BQG_QUERY(4, 2); else BQG_QUERY(4, 4);
  }
#undef BQG_QUERY
#undef BQG_BUILD
  PVN3D_LAUNCH_CHECK();
  return 0;
}
