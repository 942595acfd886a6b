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
//          candidates 64 at a time with EXACTLY the brute-force arithmetic
//          (d2 = ((dx*dx + dy*dy) + dz*dz), dx = centre - point, -ffp-contract=off), and a hit
//          sets bit k of a per-wave LDS bitmap over the index space.  Reading the bitmap back
//          in word order yields the hits in ascending k -- no sort, any hit count.
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
template <int AL>
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
  for (int k = tid; k < n; k += 1024)
    atomicAdd(&s_cnt[pad32(grid_bucket<AL>(xyz[k * 3], xyz[k * 3 + 1], xyz[k * 3 + 2], ox, oy, oz, inv_h))], 1);
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
    cell_start[tid * PER + i] = run;
    run += c;
  }
  if (tid == 1023) cell_start[T] = run;
  __syncthreads();
  for (int k = tid; k < n; k += 1024) {
    const float x = xyz[k * 3], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const int pos = atomicAdd(&s_cnt[pad32(grid_bucket<AL>(x, y, z, ox, oy, oz, inv_h))], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(k));
  }
}

// exclusive prefix sum over the wave on the DPP path (row shifts, then row_bcast 15 / 31): six VALU
// instructions, no LDS traffic (a __shfl_up ladder is six dependent ds_bpermute round trips, and a
// centre needs three scans)
__device__ __forceinline__ int wave_excl_scan_add(int v, int lane, int* total) {
  (void)lane;
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

// Pull the hits of one bitmap out in ascending index order.  words = ceil(n/32) <= 1024.
// The row is assembled in a wave-private LDS staging line `stage` (BQG_STAGE ints) and leaves with one
// coalesced store per 64 slots: the lanes find their hits at different times, and 4-byte stores issued
// one lane at a time cost a memory transaction each.
constexpr int BQG_STAGE = 64;
__device__ __forceinline__ void emit_from_bitmap(unsigned* bm, int words, int nsample,
                                                 int* __restrict__ out, int* __restrict__ stage, int lane) {
  // lane owns WPL consecutive words
  const int wpl = (words + 63) >> 6;
  const int w0 = lane * wpl;
  int mine = 0;
  for (int i = 0; i < wpl; ++i) {
    const int w = w0 + i;
    if (w < words) mine += __builtin_popcount(bm[w]);
  }
  int total;
  int rank = wave_excl_scan_add(mine, lane, &total);
  // first hit (smallest index) = first set bit overall
  const unsigned long long has = __ballot(mine > 0);
  int first = 0;
  if (has) {
    const int fl = __builtin_ctzll(has);
    int f = 0;
    if (lane == fl) {
      for (int i = 0; i < wpl; ++i) {
        const unsigned v = bm[w0 + i];
        if (v) { f = (w0 + i) * 32 + __builtin_ctz(v); break; }
      }
    }
    first = __builtin_amdgcn_readlane(f, fl);
  }
  const bool staged = nsample <= BQG_STAGE;
  int* dst = staged ? stage : out;
  if (mine > 0) {
    for (int i = 0; i < wpl; ++i) {
      const int w = w0 + i;
      if (w >= words) break;
      unsigned v = bm[w];
      if (!v) continue;
      bm[w] = 0u;  // leave the bitmap clean for the next centre
      while (v && rank < nsample) {
        const int bit = __builtin_ctz(v);
        v &= v - 1;
        dst[rank++] = w * 32 + bit;
      }
      if (v) rank += __builtin_popcount(v);
    }
  }
  // pad slots [min(total, nsample), nsample) with the first hit; no hit -> zeros
  const int filled = total < nsample ? total : nsample;
  for (int l = filled + lane; l < nsample; l += 64) dst[l] = first;
  if (staged && lane < nsample) out[lane] = stage[lane];     // (same wave: LDS ops complete in order)
}

// ---- rank pass (the common case: <= BQG_CAND candidates) --------------------------------------------
// The owner-lane extraction above is a nest of per-lane loops: at PVN3D's densities (10-30 hits in a
// 12288-bit map) it costs ~250 mostly scalar (exec-mask) instructions per bitmap.  The hit-driven form has
// no divergent loop: (1) every lane counts the bits of its WPL consecutive words and one wave scan turns
// that into pre[w] = number of hits below word w; (2) the candidates, parked as (k | flags) during the
// distance pass, are walked again 64 at a time and a hit writes itself to slot
// pre[k >> 5] + popcount(bm[k >> 5] & bits below k) = its rank in ascending index order.
constexpr int BQG_CAND = 256;
constexpr int BQG_FLAG_A = 1 << 30, BQG_FLAG_B = 1 << 31, BQG_KMASK = BQG_FLAG_A - 1;

template <int WPL>
__device__ __forceinline__ int rank_prefix(unsigned* __restrict__ bm, unsigned short* __restrict__ pre, int lane) {
  const int w0 = lane * WPL;
  int below[WPL];
  int run = 0;
#pragma unroll
  for (int i = 0; i < WPL; ++i) {
    below[i] = run;
    run += __builtin_popcount(bm[w0 + i]);
  }
  int total;
  const int base = wave_excl_scan_add(run, lane, &total);
#pragma unroll
  for (int i = 0; i < WPL; ++i) pre[w0 + i] = (unsigned short)(base + below[i]);   // <= n <= 32768
  return total;
}

// slots [0, min(total, nsample)) hold the hits; pad with the first hit (slot BQG_STAGE), no hit -> zeros
__device__ __forceinline__ void emit_row(const int* __restrict__ stage, int total, int nsample,
                                         int* __restrict__ out, int lane) {
  const int filled = total < nsample ? total : nsample;
  if (nsample <= BQG_STAGE) {
    if (lane < nsample) out[lane] = total ? stage[lane < filled ? lane : BQG_STAGE] : 0;
  } else {      // hits went straight to `out`
    const int first = total ? stage[BQG_STAGE] : 0;
    for (int l = filled + lane; l < nsample; l += 64) out[l] = first;
  }
}

// one wave per centre at a time, 4 waves per workgroup, each wave works through centres
// j = blockIdx.x*4 + wave, + 4*gridDim.x, ... (its bitmaps are cleared once: the emission leaves them
// clean).  WPL = bitmap words per lane (64 * WPL * 32 >= n).
// dynamic LDS, per wave: bitmaps [NB][64*WPL] u32 | prefix counts [NB][64*WPL] u16.
template <bool PAIR, int AL, int WPL>
__global__ __launch_bounds__(256) void ball_query_grid_kernel(
    int n, int m, float inv_h, float r2a, int nsa, float r2b, int nsb,
    const float* __restrict__ new_xyz_all, const float* __restrict__ xyz_all, const int* __restrict__ cell_start,
    const float4* __restrict__ sorted, int* __restrict__ idxa, int* __restrict__ idxb) {
  constexpr int WORDS = 64 * WPL;
  constexpr int NB = PAIR ? 2 : 1;
  extern __shared__ unsigned s_bm[];  // [4 waves][NB][WORDS] u32, then [4 waves][NB][WORDS] u16
  __shared__ int2 s_cell[4][32];      // {first flat candidate of the cell, its offset in `sorted`}
  __shared__ int s_cand[4][BQG_CAND];
  __shared__ int s_stage[4][NB][BQG_STAGE + 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bi = blockIdx.y;
  unsigned* bma = s_bm + (size_t)wave * NB * WORDS;
  unsigned* bmb = bma + WORDS;
  unsigned short* prea = reinterpret_cast<unsigned short*>(s_bm + (size_t)4 * NB * WORDS) + (size_t)wave * NB * WORDS;
  unsigned short* preb = prea + WORDS;
  int* stga = s_stage[wave][0];
  int* stgb = s_stage[wave][NB - 1];
  for (int i = lane; i < NB * WORDS; i += 64) bma[i] = 0u;
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
  float ncx, ncy, ncz;
  int nbeg, ncnt;
  fetch_cells(blockIdx.x * 4 + wave, ncx, ncy, ncz, nbeg, ncnt);
  for (int j = blockIdx.x * 4 + wave; j < m; j += 4 * gridDim.x) {
    const float cx = ncx, cy = ncy, cz = ncz;
    const int beg = nbeg, cnt = ncnt;
    fetch_cells(j + 4 * gridDim.x, ncx, ncy, ncz, nbeg, ncnt);
    int total;
    const int pref = wave_excl_scan_add(cnt, lane, &total);
    if (lane < 32) s_cell[wave][lane] = make_int2(lane < 27 ? pref : 0x7fffffff, beg);
    __builtin_amdgcn_wave_barrier();   // (single wave: its LDS operations complete in program order)
    // ---- distance pass: candidates 64 at a time, hits set their bit; (k | flags) parked for the rank pass
    for (int f0 = 0; f0 < total; f0 += 64) {
      const int f = f0 + lane;
      int parked = 0;
      if (f < total) {
        // owner cell q: largest q with first[q] <= f (27 non-decreasing entries, padded to 32 with INT_MAX)
        int q = 0;
#pragma unroll
        for (int s = 16; s >= 1; s >>= 1) q = (s_cell[wave][q + s].x <= f) ? q + s : q;
        const int2 cb = s_cell[wave][q];
        const float4 p = sorted[cb.y + (f - cb.x)];
        const float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const int k = __float_as_int(p.w);
        const bool ha = d2 < r2a, hb = PAIR && d2 < r2b;
        if (ha) atomicOr(&bma[k >> 5], 1u << (k & 31));
        if (hb) atomicOr(&bmb[k >> 5], 1u << (k & 31));
        parked = k | (ha ? BQG_FLAG_A : 0) | (hb ? BQG_FLAG_B : 0);
      }
      if (f0 < BQG_CAND) s_cand[wave][f] = parked;
    }
    __builtin_amdgcn_wave_barrier();
    int* const outa = idxa + ((size_t)bi * m + j) * nsa;
    int* const outb = PAIR ? idxb + ((size_t)bi * m + j) * nsb : nullptr;
    if (total <= BQG_CAND) {
      const int ha_total = rank_prefix<WPL>(bma, prea, lane);
      const int hb_total = PAIR ? rank_prefix<WPL>(bmb, preb, lane) : 0;
      __builtin_amdgcn_wave_barrier();
      int* const da = nsa <= BQG_STAGE ? stga : outa;
      int* const db = nsb <= BQG_STAGE ? stgb : outb;
      for (int f0 = 0; f0 < total; f0 += 64) {
        const int c = s_cand[wave][f0 + lane];
        const int k = c & BQG_KMASK;
        const int wd = k >> 5;
        const unsigned lower = (1u << (k & 31)) - 1u;
        if (c & BQG_FLAG_A) {
          const int r = prea[wd] + __builtin_popcount(bma[wd] & lower);
          if (r < nsa) da[r] = k;
          if (r == 0) stga[BQG_STAGE] = k;
        }
        if (PAIR && (c & BQG_FLAG_B)) {
          const int r = preb[wd] + __builtin_popcount(bmb[wd] & lower);
          if (r < nsb) db[r] = k;
          if (r == 0) stgb[BQG_STAGE] = k;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NB * WPL; ++i) bma[i * 64 + lane] = 0u;   // leave the bitmaps clean for the next centre
      emit_row(stga, ha_total, nsa, outa, lane);
      if (PAIR) emit_row(stgb, hb_total, nsb, outb, lane);
      __builtin_amdgcn_wave_barrier();
    } else {
      emit_from_bitmap(bma, WORDS, nsa, outa, stga, lane);
      if (PAIR) emit_from_bitmap(bmb, WORDS, nsb, outb, stgb, lane);
    }
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
    if (pair)                                                                                     \
      hipLaunchKernelGGL((ball_query_grid_kernel<true, AL, WPL>), qgrid, dim3(256), qlds, st, n,  \
                         m, inv_h, r2a, nsample0, r2b, nsample1, new_xyz, xyz, ws.cell_start,     \
                         ws.sorted, idx0, idx1);                                                  \
    else                                                                                          \
      hipLaunchKernelGGL((ball_query_grid_kernel<false, AL, WPL>), qgrid, dim3(256), qlds, st, n, \
                         m, inv_h, r2a, nsample0, 0.f, 0, new_xyz, xyz, ws.cell_start, ws.sorted, \
                         idx0, nullptr);                                                          \
  } while (0)
#define BQG_BUILD(AL)                                                                             \
  do {                                                                                            \
    auto bk = grid_build_kernel<AL>;                                                              \
    const size_t blds = (size_t)(grid_t(AL) + (grid_t(AL) >> 5)) * sizeof(int);                   \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(bk));                                     \
    hipLaunchKernelGGL(bk, dim3(b), dim3(1024), blds, st, n, inv_h, xyz, ws.cell_start,           \
                       ws.sorted);                                                                \
    PVN3D_LAUNCH_CHECK();                                                                         \
  } while (0)
  if (al == 5) {
    BQG_BUILD(5);
    if (wpl == 4) BQG_QUERY(5, 4); else if (wpl == 6) BQG_QUERY(5, 6); else if (wpl == 8) BQG_QUERY(5, 8); else BQG_QUERY(5, 16);
  } else {
    BQG_BUILD(4);
    if (wpl == 1) BQG_QUERY(4, 1); else if (wpl == 2) BQG_QUERY(4, 2); else BQG_QUERY(4, 4);
  }
#undef BQG_QUERY
#undef BQG_BUILD
  PVN3D_LAUNCH_CHECK();
  return 0;
}
