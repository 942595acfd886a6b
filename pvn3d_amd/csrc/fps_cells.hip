// fps_cells.hip -- furthest point sampling with exact spatial culling, one WAVE per cloud (gfx950).
//
// Replaces, for 4096 < n <= 12288, the serial scan of furthest_point_sampling_kernel
// (pvn3d/_ext-src/src/sampling_gpu.cu:69-173): every round of the reference (and of
// sampling.hip::fps_reg_kernel) recomputes the distance of ALL n points to the newest sample,
// although a sample only lowers the running minimum of the points near it.  Here the cloud is
// cut into 64 spatially compact cells of <= 192 points (8 slabs along the widest axis x 8 along
// the second widest, equal counts: two counting sorts in LDS), and a round
//   1. tests all 64 cells at once, one lane per cell: LB = squared distance from the sample to the
//      cell's bounding box, evaluated with the SAME unfused fp32 expression as the point
//      distance; rounding is monotone, so LB <= d(k) holds exactly for every point k of the cell,
//      and LB > max_k temp[k] proves that min(d, temp) leaves the whole cell unchanged;
//   2. updates only the touched cells (2.15 of 64 on average on the BASELINE clouds): coordinates and
//      running minima from LDS.  A second lane-parallel test -- does the sample lower the cell's CACHED
//      arg-max point? -- decides whether the cell's cached (max, arg-max) can change at all (values never
//      grow, the arg-max keeps its value and its rank among equals): only then (1.3 cells per round, the
//      sample's own cell included) a DPP wave reduction and a locate step refresh the cache;
//   3. takes the arg-max over the 64 cached cell maxima (one more DPP reduction), no barrier, no
//      cross-wave exchange: the serial chain lives in ONE wave, whose cost is the NUMBER of instructions on
//      its path (a wave alone on its SIMD issues every 5 cycles at best, every 8 when dependent -- SALU and
//      branches included, tools/issue_bench.hip), so the round is written for few instructions: reductions
//      as single-instruction DPP steps with the per-lane selects in their wait states, v_writelane cache
//      updates, the sample's coordinates carried in the cell caches (no memory access between rounds),
//      data of the sample's own cell and of the next two touched cells requested ahead.
// Ties (equal maxima inside a cell or across cells) are detected with ballots and resolved with
// the reference block's order -- smallest bit-reversed (k mod bs), then smallest k, i.e. the
// smallest fps_prio(k) of sampling.hip -- by looking the candidates' indices up in the scratch
// table; rare on real clouds, every round on exhausted / duplicated clouds (still exact).
// Non-finite input: a non-finite sample changes nothing in the reference either (every d is inf or
// NaN, fminf keeps temp), so such a round touches no cell; non-finite points only widen a box.
//
// Arithmetic: -ffp-contract=off; d = ((dx*dx + dy*dy) + dz*dz), one rounding per operation.
#include "common.h"

namespace {


// wave64 reductions on the DPP path, one instruction per step (v_max_i32 with the DPP modifier on src0; the
// compiler's own lowering of update_dpp is v_mov + s_nop + v_mov_dpp + v_max per step, and in a wave that
// runs alone on its SIMD every dependent instruction costs 8-10 cycles).  s_nop 1 = the two wait states a DPP
// read needs after the VALU write of its source; rows outside row_mask keep their value; lane 63 ends up
// with the result.  The trailing s_nop covers "VALU wrote an SGPR" for whatever the compiler puts next.
__device__ __forceinline__ int fc_wave_max_i32(int v) {
  int r;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_i32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %1, %0, 63\n\t"
      "s_nop 3"
      : "+v"(v), "=s"(r));
  return r;
}

__device__ __forceinline__ unsigned fc_wave_min_u32(unsigned v) {
  unsigned r;
  asm volatile(
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %1, %0, 63\n\t"
      "s_nop 3"
      : "+v"(v), "=s"(r));
  return r;
}

// order-preserving float <-> int (for LDS atomicMin / atomicMax on coordinates)
__device__ __forceinline__ int fc_ord(float f) {
  const int i = __float_as_int(f);
  return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float fc_unord(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

// tie-break priority of point k in the reference block (see sampling.hip::fps_prio)
__device__ __forceinline__ unsigned fc_prio(int k, int L, int Q) {
  const unsigned r = L ? (__brev((unsigned)k & ((1u << L) - 1u)) >> (32 - L)) : 0u;
  return r * (unsigned)Q + ((unsigned)k >> L);
}

// mag <= 1e-3 with mag fp32 and the literal double (sampling_gpu.cu:100-101)
__device__ __forceinline__ bool fc_skipped(float x, float y, float z) {
  const float mag = (x * x) + (y * y) + (z * z);
  return (double)mag <= 1e-3;
}

__device__ __forceinline__ int fc_wave_incl_scan(int v, int lane, int width) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d, 64);
    if ((lane & (width - 1)) >= d) v += o;
    if (d * 2 >= width) break;
  }
  return v;
}

constexpr int FC_BINS1 = 2048;   // level-1 histogram (whole cloud, widest axis)
constexpr int FC_BINS2 = 256;    // level-2 histogram per slab (second widest axis)
constexpr int FC_AUX_INTS = 4096;  // hist (2048) + misc (2048)

// misc region (ints), after the histogram
constexpr int MI_BBOX = 0;       // [6]  ordered ints: min x,y,z, max x,y,z
constexpr int MI_SLABR = 8;      // [8][2] ordered ints: min / max of the level-2 coordinate per slab
constexpr int MI_WSCAN = 32;     // [8]  wave totals of the scans
constexpr int MI_SLABLO = 48;    // [8]  float: level-2 origin per slab
constexpr int MI_SLABSC = 56;    // [8]  float: level-2 scale per slab
constexpr int MI_CBOX = 64;      // [64][6] floats: cell boxes

// One cloud after the build phase: where its arrays live, the slab boundaries (ranks along the widest axis).
template <int SPC>
struct FcCloud {
  float *X, *Y, *T, *Zl, *Zg;      // by position (cell * CELL + q); Zl: LDS (SPC < 3), Zg: workspace
  int *hist, *misc, *sorted_k;     // aux region of LDS (histogram | misc), point index by rank (workspace)
  float* miscf;
  int bnd[9];
  // cell (s, t): ranks [bnd[s] + t*len/8, bnd[s] + (t+1)*len/8), positions cell*CELL + [0, cnt)
  __device__ __forceinline__ int cell_start(int c) const {
    const int s = c >> 3, t = c & 7;
    int b0 = 0, b1 = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { if (s == q) { b0 = bnd[q]; b1 = bnd[q + 1]; } }
    return b0 + ((t * (b1 - b0)) >> 3);
  }
  __device__ __forceinline__ int cell_cnt(int c) const {
    const int s = c >> 3, t = c & 7;
    int b0 = 0, b1 = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) { if (s == q) { b0 = bnd[q]; b1 = bnd[q + 1]; } }
    const int sl = b1 - b0;
    return (((t + 1) * sl) >> 3) - ((t * sl) >> 3);
  }
};

// Build phase, all 256 threads of the workgroup (ends with a barrier): the cloud cut into 64 equal-count cells; x, y and the
// running minima by position in LDS, z in LDS or in the workspace, the point index by rank in the workspace, the cell
// boxes in the misc region.  `dataset` / `ws` already point at this workgroup's cloud.  xinit: words at the start of the
// histogram region that are set to `xinit_val` before the last barrier (the multi-wave rounds' exchange area).
template <int SPC>
__device__ __forceinline__ void fc_build(FcCloud<SPC>& cl, int n, const float* __restrict__ dataset, int* __restrict__ ws,
                                         float* s_dyn, int xinit, int xinit_val) {
  constexpr int CELL = SPC * 64;      // positions per cell
  constexpr int NPOS = 64 * CELL;     // positions in LDS
  constexpr int PPT = 16 * SPC;       // points per thread during the build
  float* const X = s_dyn;          // x, y by position (cell * CELL + q)
  float* const Y = X + NPOS;
  float* const T = Y + NPOS;       // running minimum distance ("temp" of the reference) by position
  // z: in LDS while four arrays fit (SPC < 3); at SPC = 3 (12288 points x 16 B = 192 KiB > 160 KiB) it stays in
  // the workspace (48 KiB, L2-resident, requested a whole cull test ahead of its use).  Measured: keeping it on
  // chip instead (slot 2 in the histogram region, slots 0 / 1 in VGPR vectors read through s_set_gpr_idx)
  // changes nothing, 1.585 vs 1.536 ms -- the round is bound by dependent instruction issue, not by latency.
  constexpr bool Z_IN_LDS = SPC < 3;
  float* const Zl = T + NPOS;                                   // (SPC < 3) z by position
  int* const hist = reinterpret_cast<int*>(T + NPOS + (Z_IN_LDS ? NPOS : 0));
  int* const misc = hist + FC_BINS1;
  float* const miscf = reinterpret_cast<float*>(misc);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // workspace of the cloud: z by position (NPOS floats; LDS holds x, y and temp), then the point index by
  // rank (n ints; rank = position with the pads of the cells squeezed out)
  float* const Zg = reinterpret_cast<float*>(ws);
  int* const sorted_k = ws + NPOS;

  // ------------------------------------------------------------------ build: points in registers
  float px[PPT], py[PPT], pz[PPT];
  float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
  float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = i * 256 + tid;
    if (k < n) {
      px[i] = dataset[k * 3 + 0];
      py[i] = dataset[k * 3 + 1];
      pz[i] = dataset[k * 3 + 2];
      lo[0] = __builtin_fminf(lo[0], px[i]); hi[0] = __builtin_fmaxf(hi[0], px[i]);
      lo[1] = __builtin_fminf(lo[1], py[i]); hi[1] = __builtin_fmaxf(hi[1], py[i]);
      lo[2] = __builtin_fminf(lo[2], pz[i]); hi[2] = __builtin_fmaxf(hi[2], pz[i]);
    } else {
      px[i] = py[i] = pz[i] = 0.f;
    }
  }
  for (int i = tid; i < FC_BINS1; i += 256) hist[i] = 0;
  if (tid < 3) { misc[MI_BBOX + tid] = 0x7fffffff; misc[MI_BBOX + 3 + tid] = (int)0x80000000; }
  if (tid < 8) { misc[MI_SLABR + tid * 2] = 0x7fffffff; misc[MI_SLABR + tid * 2 + 1] = (int)0x80000000; }
  // pads (and skipped points) never update temp and never win: temp = -inf keeps d2 = -inf whatever
  // coordinates the pad position holds
  if (n < NPOS)
    for (int i = tid; i < NPOS; i += 256) T[i] = -__builtin_inff();
  __syncthreads();
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
      lo[a] = __builtin_fminf(lo[a], __shfl_xor(lo[a], s, 64));
      hi[a] = __builtin_fmaxf(hi[a], __shfl_xor(hi[a], s, 64));
    }
    if (lane == 0) {
      atomicMin(&misc[MI_BBOX + a], fc_ord(lo[a]));
      atomicMax(&misc[MI_BBOX + 3 + a], fc_ord(hi[a]));
    }
  }
  __syncthreads();
  // widest (a1) and second widest (a2) axis of the cloud's box; scale = bins / extent (0 when degenerate)
  int a1, a2;
  float lo1, sc1;
  {
    float e[3], l[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      l[a] = fc_unord(misc[MI_BBOX + a]);
      e[a] = fc_unord(misc[MI_BBOX + 3 + a]) - l[a];
      if (!(e[a] >= 0.f && e[a] <= 3.0e38f)) e[a] = 0.f;   // empty / infinite / NaN extent: one bin
    }
    a1 = (e[0] >= e[1] && e[0] >= e[2]) ? 0 : (e[1] >= e[2] ? 1 : 2);
    const int b = a1 == 0 ? 1 : 0, c = a1 == 2 ? 1 : 2;    // the two other axes
    a2 = e[b] >= e[c] ? b : c;
    lo1 = l[a1];
    sc1 = e[a1] > 0.f ? (float)FC_BINS1 / e[a1] : 0.f;
  }
  // slab boundaries (ranks along a1): slab s = [s*n/8, (s+1)*n/8)
  int (&bnd)[9] = cl.bnd;
#pragma unroll
  for (int s = 0; s <= 8; ++s) bnd[s] = (int)(((long long)s * n) >> 3);

  // level 1: counting sort along a1
#define FC_COORD(a, i) ((a) == 0 ? +px[i] : ((a) == 1 ? +py[i] : +pz[i]))
#define FC_BIN1(i) min(max((int)((FC_COORD(a1, i) - lo1) * sc1), 0), FC_BINS1 - 1) /* NaN -> 0 */
#pragma unroll
  for (int i = 0; i < PPT; ++i)
    if (i * 256 + tid < n) atomicAdd(&hist[FC_BIN1(i)], 1);
  __syncthreads();
  {
    int v[8], run = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = run; run += hist[tid * 8 + i]; }
    const int incl = fc_wave_incl_scan(run, lane, 64);
    if (lane == 63) misc[MI_WSCAN + wave] = incl;
    __syncthreads();
    int base = incl - run;
    for (int w = 0; w < wave; ++w) base += misc[MI_WSCAN + w];
#pragma unroll
    for (int i = 0; i < 8; ++i) hist[tid * 8 + i] = base + v[i];
  }
  __syncthreads();
  int slab[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    slab[i] = 0;
    if (i * 256 + tid < n) {
      const int r1 = atomicAdd(&hist[FC_BIN1(i)], 1);
      int s = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) s += (r1 >= bnd[q]) ? 1 : 0;
      slab[i] = s;
      const float c2 = FC_COORD(a2, i);
      if (c2 == c2) {
        atomicMin(&misc[MI_SLABR + s * 2], fc_ord(c2));
        atomicMax(&misc[MI_SLABR + s * 2 + 1], fc_ord(c2));
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < FC_BINS1; i += 256) hist[i] = 0;
  if (tid < 8) {
    const float l = fc_unord(misc[MI_SLABR + tid * 2]);
    float e = fc_unord(misc[MI_SLABR + tid * 2 + 1]) - l;
    if (!(e >= 0.f && e <= 3.0e38f)) e = 0.f;
    miscf[MI_SLABLO + tid] = l;
    miscf[MI_SLABSC + tid] = e > 0.f ? (float)FC_BINS2 / e : 0.f;
  }
  __syncthreads();
  // level 2: counting sort along a2 inside every slab
#define FC_BIN2(i)                                                                                        \
  (slab[i] * FC_BINS2 +                                                                                   \
   min(max((int)((FC_COORD(a2, i) - miscf[MI_SLABLO + slab[i]]) * miscf[MI_SLABSC + slab[i]]), 0), FC_BINS2 - 1))
#pragma unroll
  for (int i = 0; i < PPT; ++i)
    if (i * 256 + tid < n) atomicAdd(&hist[FC_BIN2(i)], 1);
  __syncthreads();
  {
    // thread t: slab t/32, bins (t%32)*8 .. +7; 32-lane exclusive scan per slab
    int v[8], run = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = run; run += hist[tid * 8 + i]; }
    const int incl = fc_wave_incl_scan(run, lane, 32);
    const int base = incl - run;
#pragma unroll
    for (int i = 0; i < 8; ++i) hist[tid * 8 + i] = base + v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = i * 256 + tid;
    if (k < n) {
      const int s = slab[i];
      const int r2 = atomicAdd(&hist[FC_BIN2(i)], 1);      // rank inside the slab
      int b0 = 0, b1 = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) { if (s == q) { b0 = bnd[q]; b1 = bnd[q + 1]; } }
      const int sl = b1 - b0;
      int t = 0;
#pragma unroll
      for (int q = 1; q < 8; ++q) t += (r2 >= ((q * sl) >> 3)) ? 1 : 0;
      const int qq = r2 - ((t * sl) >> 3);
      const int pos = (s * 8 + t) * CELL + qq;
      X[pos] = px[i];
      Y[pos] = py[i];
      if (Z_IN_LDS) Zl[pos] = pz[i];
      else Zg[pos] = pz[i];
      T[pos] = fc_skipped(px[i], py[i], pz[i]) ? -__builtin_inff() : 1e10f;
      sorted_k[b0 + r2] = k;
    }
  }
  __syncthreads();
  cl.X = X; cl.Y = Y; cl.T = T; cl.Zl = Zl; cl.Zg = Zg;
  cl.hist = hist; cl.misc = misc; cl.miscf = miscf; cl.sorted_k = sorted_k;
  for (int i = tid; i < xinit; i += 256) hist[i] = xinit_val;      // (the histogram is dead from here on)
  // cell boxes: wave w reduces cells w*16 .. w*16+15
  for (int cc = 0; cc < 16; ++cc) {
    const int c = wave * 16 + cc;
    const int cnt = cl.cell_cnt(c);
    float bl[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float bh[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
    for (int sl = 0; sl < SPC; ++sl) {
      const int q = sl * 64 + lane;
      if (q < cnt) {
        const float x = X[c * CELL + q], y = Y[c * CELL + q], z = Z_IN_LDS ? Zl[c * CELL + q] : Zg[c * CELL + q];
        bl[0] = __builtin_fminf(bl[0], x); bh[0] = __builtin_fmaxf(bh[0], x);
        bl[1] = __builtin_fminf(bl[1], y); bh[1] = __builtin_fmaxf(bh[1], y);
        bl[2] = __builtin_fminf(bl[2], z); bh[2] = __builtin_fmaxf(bh[2], z);
      }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
      for (int s = 32; s >= 1; s >>= 1) {
        bl[a] = __builtin_fminf(bl[a], __shfl_xor(bl[a], s, 64));
        bh[a] = __builtin_fmaxf(bh[a], __shfl_xor(bh[a], s, 64));
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) { miscf[MI_CBOX + c * 6 + a] = bl[a]; miscf[MI_CBOX + c * 6 + 3 + a] = bh[a]; }
    }
  }
  __syncthreads();
}

template <int SPC>
__global__ __launch_bounds__(256) void fps_cells_kernel(int n, int m, int L, int Q,
                                                        const float* __restrict__ dataset,
                                                        int* __restrict__ ws, int* __restrict__ idxs,
                                                        int* __restrict__ dmax) {
  constexpr int CELL = SPC * 64;      // positions per cell
  constexpr int NPOS = 64 * CELL;     // positions in LDS
  constexpr bool Z_IN_LDS = SPC < 3;
  extern __shared__ float s_dyn[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  dataset += (size_t)blockIdx.x * n * 3;
  ws += (size_t)blockIdx.x * (NPOS + n);
  idxs += (size_t)blockIdx.x * m;
  if (dmax) dmax += (size_t)blockIdx.x * m;
  FcCloud<SPC> cl;
  fc_build<SPC>(cl, n, dataset, ws, s_dyn, 0, 0);
  if (wave != 0) return;
  float* const X = cl.X;
  float* const Y = cl.Y;
  float* const T = cl.T;
  float* const Zl = cl.Zl;
  float* const Zg = cl.Zg;
  const float* const miscf = cl.miscf;
  const int* const sorted_k = cl.sorted_k;
  auto cell_start = [&](int c) { return cl.cell_start(c); };

  // ------------------------------------------------------------------ the serial rounds: wave 0 alone
  // A wave that runs alone on its SIMD issues one instruction every 5 cycles at best and every 8 when it
  // depends on the previous one, SALU and branches included (tools/issue_bench.hip): the round is written
  // for the smallest instruction count on its path, not for arithmetic throughput.
  // lane c = cell c
  const float lox = miscf[MI_CBOX + lane * 6 + 0], loy = miscf[MI_CBOX + lane * 6 + 1],
              loz = miscf[MI_CBOX + lane * 6 + 2];
  const float hix = miscf[MI_CBOX + lane * 6 + 3], hiy = miscf[MI_CBOX + lane * 6 + 4],
              hiz = miscf[MI_CBOX + lane * 6 + 5];
  const int cstart = cell_start(lane);      // rank of the cell's position 0
  int cmax = __float_as_int(1e10f);         // cached cell maximum (bit pattern; < 0: no valid point)
  int cq = 0;                               // in-cell position q of its arg-max under the tie order
  float cwx = 0.f, cwy = 0.f, cwz = 0.f;    // ... and that point's coordinates

  const float p0x = dataset[0], p0y = dataset[1], p0z = dataset[2];   // (uniform) point 0 = the seed
  float sx = p0x, sy = p0y, sz = p0z;
  int res = -1;          // lane (j & 63): rank of pick j, -1 = "index 0" (the seed / no valid point)
  int resd = 0;          // lane (j & 63): winning distance of round j

  struct CellRegs { float x[SPC], y[SPC], z[SPC], t[SPC]; };
  auto load_cell = [&](int c, CellRegs& r) {
    const int p = c * CELL + lane;
#pragma unroll
    for (int sl = 0; sl < SPC; ++sl) {
      r.x[sl] = X[p + sl * 64];
      r.y[sl] = Y[p + sl * 64];
      r.t[sl] = T[p + sl * 64];
      r.z[sl] = Z_IN_LDS ? Zl[p + sl * 64] : Zg[p + sl * 64];
    }
  };
  // running minima of one cell against the current sample; returns the new values' bit patterns
  auto update_cell = [&](int c, const CellRegs& r, int (&vi)[SPC]) {
#pragma unroll
    for (int sl = 0; sl < SPC; ++sl) {
      const float dx = r.x[sl] - sx, dy = r.y[sl] - sy, dz = r.z[sl] - sz;
      const float d = dx * dx + dy * dy + dz * dz;
      // fminf without the compiler's canonicalising v_max(t, t): t is never NaN (1e10, -inf or an earlier
      // minimum), and v_min_f32 returns the other operand for a NaN d
      float d2;
      asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(r.t[sl]));
      T[c * CELL + sl * 64 + lane] = d2;
      vi[sl] = __float_as_int(d2);
    }
  };
  // A cell whose cached arg-max may have been lowered: new maximum (DPP wave reduction) and arg-max.  The
  // per-lane "which slot holds my largest value, and that point's coordinates" selects fill the wait states
  // of the reduction steps (a DPP read needs two after the write of its source).
  auto refresh_cell = [&](int c, const CellRegs& r, const int (&vi)[SPC]) {
    int M, ls = 0, mloc = vi[0];
    float lx = r.x[0], ly = r.y[0], lz = r.z[0];
    unsigned long long ct = 0;      // lanes where a second slot holds the lane's largest value
    if constexpr (SPC == 3) {
      mloc = max(max(vi[0], vi[1]), vi[2]);
      int red, md;
      unsigned long long c0, c1;
      asm volatile(
          "v_mov_b32 %[red], %[mloc]\n\t"
          "v_cmp_eq_u32 %[c1], %[v1], %[mloc]\n\t"
          "v_cmp_eq_u32 %[c0], %[v0], %[mloc]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "v_med3_i32 %[md], %[v0], %[v1], %[v2]\n\t"
          "v_cndmask_b32_e64 %[ls], 2, 1, %[c1]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_e64 %[lx], %[x2], %[x1], %[c1]\n\t"
          "v_cndmask_b32_e64 %[ly], %[y2], %[y1], %[c1]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_e64 %[lz], %[z2], %[z1], %[c1]\n\t"
          "v_cndmask_b32_e64 %[ls], %[ls], 0, %[c0]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_mirror row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_e64 %[lx], %[lx], %[x0], %[c0]\n\t"
          "v_cndmask_b32_e64 %[ly], %[ly], %[y0], %[c0]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
          "v_cndmask_b32_e64 %[lz], %[lz], %[z0], %[c0]\n\t"
          "v_cmp_eq_u32 %[ct], %[md], %[mloc]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_readlane_b32 %[M], %[red], 63\n\t"
          "s_nop 1"
          : [M] "=s"(M), [red] "=&v"(red), [md] "=&v"(md), [ls] "=&v"(ls), [lx] "=&v"(lx), [ly] "=&v"(ly),
            [lz] "=&v"(lz), [c0] "=&s"(c0), [c1] "=&s"(c1), [ct] "=&s"(ct)
          : [mloc] "v"(mloc), [v0] "v"(vi[0]), [v1] "v"(vi[1]), [v2] "v"(vi[2]), [x0] "v"(r.x[0]), [x1] "v"(r.x[1]),
            [x2] "v"(r.x[2]), [y0] "v"(r.y[0]), [y1] "v"(r.y[1]), [y2] "v"(r.y[2]), [z0] "v"(r.z[0]),
            [z1] "v"(r.z[1]), [z2] "v"(r.z[2]));
    } else if constexpr (SPC == 2) {
      mloc = max(vi[0], vi[1]);
      int red;
      unsigned long long c0;
      asm volatile(
          "v_mov_b32 %[red], %[mloc]\n\t"
          "v_cmp_eq_u32 %[c0], %[v0], %[mloc]\n\t"
          "v_cmp_eq_u32 %[ct], %[v0], %[v1]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_e64 %[ls], 1, 0, %[c0]\n\t"
          "v_cndmask_b32_e64 %[lx], %[x1], %[x0], %[c0]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
          "v_cndmask_b32_e64 %[ly], %[y1], %[y0], %[c0]\n\t"
          "v_cndmask_b32_e64 %[lz], %[z1], %[z0], %[c0]\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_mirror row_mask:0xf bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_max_i32_dpp %[red], %[red], %[red] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
          "s_nop 1\n\t"
          "v_readlane_b32 %[M], %[red], 63\n\t"
          "s_nop 1"
          : [M] "=s"(M), [red] "=&v"(red), [ls] "=&v"(ls), [lx] "=&v"(lx), [ly] "=&v"(ly), [lz] "=&v"(lz),
            [c0] "=&s"(c0), [ct] "=&s"(ct)
          : [mloc] "v"(mloc), [v0] "v"(vi[0]), [v1] "v"(vi[1]), [x0] "v"(r.x[0]), [x1] "v"(r.x[1]),
            [y0] "v"(r.y[0]), [y1] "v"(r.y[1]), [z0] "v"(r.z[0]), [z1] "v"(r.z[1]));
    } else {
      M = fc_wave_max_i32(mloc);
    }
    int q = 0;
    float wx = 0.f, wy = 0.f, wz = 0.f;
    const unsigned long long e = __ballot(mloc == M);
    if (__builtin_expect(M >= 0 && !(e & (e - 1)) && !(e & ct), 1)) {
      const int wl = __builtin_ctzll(e);
      q = __builtin_amdgcn_readlane(ls, wl) * 64 + wl;
      wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lx), wl));
      wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ly), wl));
      wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lz), wl));
    } else if (M >= 0) {
      // equal maxima inside the cell: the reference's order = smallest priority
      const int cs = __builtin_amdgcn_readlane(cstart, c);
      unsigned key = 0xffffffffu;
#pragma unroll
      for (int sl = 0; sl < SPC; ++sl) {
        if (vi[sl] == M) {
          const int qq = sl * 64 + lane;
          const unsigned p = fc_prio(sorted_k[cs + qq], L, Q);
          key = min(key, (p << 8) | (unsigned)qq);
        }
      }
      key = fc_wave_min_u32(key);
      q = (int)(key & 255u);
      const int wl = q & 63, ws = q >> 6;      // (uniform)
      float tx = r.x[0], ty = r.y[0], tz = r.z[0];
#pragma unroll
      for (int sl = 1; sl < SPC; ++sl) {
        tx = ws == sl ? r.x[sl] : tx;
        ty = ws == sl ? r.y[sl] : ty;
        tz = ws == sl ? r.z[sl] : tz;
      }
      wx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tx), wl));
      wy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ty), wl));
      wz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tz), wl));
    }
    // the arg-max's coordinates ride in the cell's cache (lane c): no memory access between two rounds
    // (two SGPR operands exceed the constant bus: the lane select goes through M0, like the compiler's own
    // v_writelane lowering)
    asm volatile(
        "s_mov_b32 m0, %[c]\n\t"
        "s_nop 1\n\t"
        "v_writelane_b32 %[cmax], %[M], m0\n\t"
        "v_writelane_b32 %[cq], %[q], m0\n\t"
        "v_writelane_b32 %[cwx], %[wx], m0\n\t"
        "v_writelane_b32 %[cwy], %[wy], m0\n\t"
        "v_writelane_b32 %[cwz], %[wz], m0"
        : [cmax] "+v"(cmax), [cq] "+v"(cq), [cwx] "+v"(cwx), [cwy] "+v"(cwy), [cwz] "+v"(cwz)
        : [M] "s"(M), [q] "s"(q), [wx] "s"(wx), [wy] "s"(wy), [wz] "s"(wz), [c] "s"(c)
        : "m0");
  };

  int cw = -1;           // the previous round's winning cell: always touched
  unsigned long long force = ~0ull;      // first round: the caches are not valid yet, every cell is refreshed
  CellRegs ra, rb, rc;
  for (int j = 1; j < m; ++j) {
    // the sample's own cell is certainly touched: request its data before anything else (unconditionally --
    // cell 0 when there is none -- so that the registers are not merged with last round's under a wait)
    load_cell(cw < 0 ? 0 : cw, ra);
    // 1. one lane per cell: (a) cull test; (b) does the sample lower the cell's cached arg-max?  Only then
    // can the cell's cached (max, arg-max) change: values never grow, and the arg-max keeps its value and its
    // rank among equals.  Same unfused expression as the point distance, so both tests are exact.
    const float ax = __builtin_fmaxf(__builtin_fmaxf(lox - sx, sx - hix), 0.f);
    const float ay = __builtin_fmaxf(__builtin_fmaxf(loy - sy, sy - hiy), 0.f);
    const float az = __builtin_fmaxf(__builtin_fmaxf(loz - sz, sz - hiz), 0.f);
    const float lb = ax * ax + ay * ay + az * az;
    const float ex = cwx - sx, ey = cwy - sy, ez = cwz - sz;
    const float dw = ex * ex + ey * ey + ez * ez;
    // lb is NaN only for a sample with a NaN coordinate, and such a sample changes no running minimum
    // (every d is NaN, fminf keeps temp): "<=" leaves every cell alone then.  An infinite coordinate gives
    // lb = inf; a cell without valid points has cmax = -inf.
    unsigned long long mask = __ballot(lb <= __int_as_float(cmax));
    unsigned long long fullm = __ballot(dw < __int_as_float(cmax));
    // the own cell is handled apart (cw = -1 clears bit 63: only when no cell is touched, or before `force`)
    // (readfirstlane: the inline-asm "s" operands below do not stop the compiler from keeping the wave-uniform cw in a VGPR)
    const int cws = __builtin_amdgcn_readfirstlane(cw);
    asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(cws));
    mask |= force;
    fullm = (fullm | force) & mask;
    force = 0;
    // 2. the first two other touched cells: data requested now, used after the sample's own cell
    int c1 = -1, c2 = -1;
    if (mask) {
      c1 = __builtin_ctzll(mask);
      mask &= mask - 1;
      load_cell(c1, rb);
      if (mask) {
        c2 = __builtin_ctzll(mask);
        mask &= mask - 1;
        load_cell(c2, rc);
      }
    }
    // the sample's own cell: its data was requested as soon as the arg-max of the previous round was known
    if (__builtin_expect(cws >= 0, 1)) {
      int vi[SPC];
      update_cell(cws, ra, vi);
      refresh_cell(cws, ra, vi);
    }
    if (c1 >= 0) {
      int vi[SPC];
      update_cell(c1, rb, vi);
      if (c2 >= 0) update_cell(c2, rc, vi);
    }
    // the rest: further touched cells, and the cells above whose arg-max was lowered
    mask |= fullm;
    while (mask) {
      const int c = __builtin_ctzll(mask);
      mask &= mask - 1;
      load_cell(c, rb);
      int vi[SPC];
      update_cell(c, rb, vi);          // (idempotent for a cell that was updated above)
      if ((fullm >> c) & 1) refresh_cell(c, rb, vi);
    }
    // 3. arg-max over the cells
    const int G = fc_wave_max_i32(cmax);
    int wrank;
    if (__builtin_expect(G < 0, 0)) {
      // no valid point at all: the reference's threads keep (best = -1, besti = 0) -> index 0
      wrank = -1;
      cw = -1;
      sx = p0x; sy = p0y; sz = p0z;
    } else {
      const unsigned long long g = __ballot(cmax == G);
      cw = __builtin_ctzll(g);
      if (__builtin_expect((g & (g - 1)) != 0, 0)) {
        // tie across cells
        unsigned key = 0xffffffffu;
        if (cmax == G) key = (fc_prio(sorted_k[cstart + cq], L, Q) << 6) | (unsigned)lane;
        key = fc_wave_min_u32(key);
        cw = (int)(key & 63u);
      }
      wrank = __builtin_amdgcn_readlane(cstart, cw) + __builtin_amdgcn_readlane(cq, cw);
      sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwx), cw));
      sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwy), cw));
      sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwz), cw));
    }
    // 4. results: 64 rounds per coalesced store
    {
      const int jl = j & 63;
      asm volatile("s_mov_b32 m0, %4\n\ts_nop 1\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                   : "+v"(res), "+v"(resd) : "s"(wrank), "s"(G), "s"(jl) : "m0");
    }
    if ((j & 63) == 63 || j == m - 1) {
      const int jj = (j & ~63) + lane;
      if (jj <= j) {
        idxs[jj] = res < 0 ? 0 : sorted_k[res];
        if (dmax && jj >= 1) dmax[jj] = resd;
      }
    }
  }
  if (m == 1 && lane == 0) idxs[0] = 0;
}

// ======================================================================================================================
// Multi-wave rounds (4096 < n <= 12288): SPC waves per cloud on SPC SIMDs of one CU, wave w owns SLOT w -- positions
// 64 w .. 64 w + 63 -- of EVERY cell.  (sampling_gpu.cu:69-173 gives a cloud a 512-thread block; the one-wave kernel
// above spends ~250 dependent-issue slots per round, most of them on the three slots of the touched cells.)
//   * every wave keeps the complete per-cell caches (max, arg-max position and coordinates: lane = cell) and runs the cull
//     test and the arg-max over the cells redundantly -- same inputs, same instructions, same results: the next sample
//     needs no exchange;
//   * a touched cell costs a wave one slot instead of three; a cell whose cache must be refreshed (the sample's own cell
//     and the cells whose cached arg-max the sample lowers: 1.3 per round) is reduced over the wave's 64 points, the lane
//     holding the wave's maximum writes a 20-byte entry (max, position, coordinates) into an LDS exchange area, the wave
//     raises its sequence word, and -- after updating the cells that need no refresh, which covers the round trip -- reads
//     the other waves' entries: the cell's new cache is the largest entry;
//   * no barrier: sequence words are monotone counters in LDS, polled with one ds_read_b128; the LDS executes a wave's
//     instructions in order, so an entry written before the counter is visible to whoever has seen the counter.  Entries
//     are double-buffered by exchange parity (a wave can be at most one exchange ahead of the slowest);
//   * ties -- more than one wave, or more than one lane of the winning wave, at the maximum -- are resolved like in the
//     one-wave kernel with the reference block's order (smallest fc_prio) through a second exchange of keys; every wave
//     takes the same decisions from the same published words, so the waves never disagree on the sequence of exchanges.
// Index-exact with the one-wave kernel by construction: the running minima are the same values in the same positions,
// a cell's cache is the same (max, smallest-priority arg-max), the cross-cell arg-max is the same code.
//
// MEASURED (round 6, profiles/r06_fps_multiwave.txt): index-exact on every FPS test, and SLOWER than one wave -- 12288 ->
// 2048: 2.21 ms (1.08 us per round) against 1.53 ms (0.75).  Cycles per round and wave (-DPVN3D_FC_PROF): cull test 315 +
// arg-max over the cells 390-470 are the same work as in the one-wave kernel and stay serial; what the three waves share
// (one slot instead of three per touched cell) is replaced by slot partials + publish 745, the LDS round trip 350-430
// (no failed poll: that is ONE ds_read_b128 of the sequence words behind the wave's own queued LDS traffic) and the
// combination of the entries into the caches 680 -- 2 040 cycles where the one-wave kernel spends ~1 090 on its three
// slots.  A wave alone on its SIMD pays 8-16 cycles per instruction whatever the instruction does, so fewer points per
// wave buy nothing and every exchange step is ~25 more instructions on the serial path.  The kernel stays selectable
// (waves_per_cloud >= 2) as an independently written cross-check of the one-wave kernel; it is not the default.
constexpr int FX_B = 8;                                   // refreshed cells per exchange
constexpr int FX_SEQ = 0;                                 // [4] exchange counters, one per wave
constexpr int FX_ENT = 16;                                // entries [2 parities][FX_B cells][4 waves][8 words]
constexpr int FX_TIE = FX_ENT + 2 * FX_B * 4 * 8;         // tie entries [2 parities][4 waves][8 words]
constexpr int FX_WORDS = FX_TIE + 2 * 4 * 8;
constexpr int FX_NONE = (int)0x80000000;                  // initial value of every word: below every counter / entry

// Tuning build only (-DPVN3D_FC_PROF, tools/build_probe_lib.sh, tools/fps_prof.py): cycles per phase of a round, summed
// over the rounds, written by every wave over the first words of the cloud's workspace when the run is over.
#ifdef PVN3D_FC_PROF
#define FC_PROF_DECL unsigned fc_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long fc_t = __builtin_readcyclecounter()
#define FC_PROF(P)                                                  \
  do {                                                              \
    const unsigned long long t__ = __builtin_readcyclecounter();    \
    fc_acc[P] += (unsigned)(t__ - fc_t);                            \
    fc_t = t__;                                                     \
  } while (0)
#define FC_PROF_CNT(P, V) fc_acc[P] += (unsigned)(V)
#else
#define FC_PROF_DECL do { } while (0)
#define FC_PROF(P) do { } while (0)
#define FC_PROF_CNT(P, V) do { } while (0)
#endif

template <int SPC>
__global__ __launch_bounds__(256) void fps_cells_mw_kernel(int n, int m, int L, int Q,
                                                           const float* __restrict__ dataset,
                                                           int* __restrict__ ws, int* __restrict__ idxs,
                                                           int* __restrict__ dmax) {
  static_assert(SPC == 2 || SPC == 3, "one wave per 64-point slot of a cell");
  constexpr int CELL = SPC * 64;
  constexpr int NPOS = 64 * CELL;
  constexpr bool Z_IN_LDS = SPC < 3;
  extern __shared__ float s_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  dataset += (size_t)blockIdx.x * n * 3;
  ws += (size_t)blockIdx.x * (NPOS + n);
  idxs += (size_t)blockIdx.x * m;
  if (dmax) dmax += (size_t)blockIdx.x * m;
  FcCloud<SPC> cl;
  fc_build<SPC>(cl, n, dataset, ws, s_dyn, FX_WORDS, FX_NONE);
  if (wave >= SPC) return;
  float* const X = cl.X;
  float* const Y = cl.Y;
  float* const T = cl.T;
  float* const Zl = cl.Zl;
  float* const Zg = cl.Zg;
  const float* const miscf = cl.miscf;
  const int* const sorted_k = cl.sorted_k;
  int* const xw = cl.hist;                        // the exchange area (the histogram is dead)
  const int soff = wave * 64 + lane;              // this lane's position inside every cell

  // lane c = cell c (every wave holds all of it)
  const float lox = miscf[MI_CBOX + lane * 6 + 0], loy = miscf[MI_CBOX + lane * 6 + 1],
              loz = miscf[MI_CBOX + lane * 6 + 2];
  const float hix = miscf[MI_CBOX + lane * 6 + 3], hiy = miscf[MI_CBOX + lane * 6 + 4],
              hiz = miscf[MI_CBOX + lane * 6 + 5];
  const int cstart = cl.cell_start(lane);
  int cmax = __float_as_int(1e10f);
  int cq = 0;
  float cwx = 0.f, cwy = 0.f, cwz = 0.f;

  const float p0x = dataset[0], p0y = dataset[1], p0z = dataset[2];
  float sx = p0x, sy = p0y, sz = p0z;
  int res = -1, resd = 0;                         // wave 0 only
  FC_PROF_DECL;

  struct Slot { float x, y, z, t; };
  auto load_slot = [&](int c, Slot& r) {
    const int p = c * CELL + soff;
    r.x = X[p];
    r.y = Y[p];
    r.t = T[p];
    r.z = Z_IN_LDS ? Zl[p] : Zg[p];
  };
  // this wave's points of cell c against the current sample; returns the new running minimum's bit pattern
  auto update_slot = [&](int c, const Slot& r) -> int {
    const float dx = r.x - sx, dy = r.y - sy, dz = r.z - sz;
    const float d = dx * dx + dy * dy + dz * dz;
    float d2;
    asm("v_min_f32 %0, %1, %2" : "=v"(d2) : "v"(d), "v"(r.t));       // (see update_cell above)
    T[c * CELL + soff] = d2;
    return __float_as_int(d2);
  };

  int xc = 0;          // exchanges published so far (every wave counts alike)
  int nx = 0, tx = 0;  // ... of which batches of refreshed cells / tie resolutions: parity of the two entry areas
  const unsigned seq_addr = (unsigned)(uintptr_t)(xw + FX_SEQ);
  auto publish = [&]() {
    ++xc;
    asm volatile("" ::: "memory");                 // (entries first: the LDS executes this wave's writes in order)
    if (lane == 0) __hip_atomic_store(&xw[FX_SEQ + wave], xc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
  };
  auto wait_all = [&]() {
    // watchdog: a protocol error must end as a kernel fault, never as a hung GPU (the longest legitimate wait is another
    // wave's share of a round, a few microseconds)
    unsigned spins = 0;
    for (;;) {
      int4 s;
      asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(s) : "v"(seq_addr) : "memory");
      int mn = min(s.x, s.y);
      if (SPC == 3) mn = min(mn, s.z);
      if (__builtin_amdgcn_readfirstlane(mn) >= xc) break;
      if (++spins > (1u << 22)) __builtin_trap();
    }
    FC_PROF_CNT(6, spins);
  };
  // the lane `wl` of this wave writes (key, q, coordinates of its point) as this wave's entry
  auto put_entry = [&](int* e, int wl, int key, int q, const Slot& r) {
    if (lane == wl) {
      *reinterpret_cast<int4*>(e) = make_int4(key, q, __float_as_int(r.x), __float_as_int(r.y));
      e[4] = __float_as_int(r.z);
    }
  };
  // the four entries at `base` (unused waves: FX_NONE): largest key, whether it is unique, and its entry's words
  auto get_entries = [&](const int* base, int& key, bool& unique, unsigned& winners, int& q, float& x, float& y, float& z,
                         int4& mine) {
    const int* e = base + (lane & 3) * 8;
    mine = *reinterpret_cast<const int4*>(e);
    const int zb = e[4];
    int red = mine.x;
    asm volatile(
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(red));
    key = __builtin_amdgcn_readfirstlane(red);
    winners = (unsigned)__ballot(mine.x == key) & 0xfu;
    unique = (winners & (winners - 1u)) == 0u;
    const int wl = __builtin_ctz(winners);
    q = __builtin_amdgcn_readlane(mine.y, wl);
    x = __int_as_float(__builtin_amdgcn_readlane(mine.z, wl));
    y = __int_as_float(__builtin_amdgcn_readlane(mine.w, wl));
    z = __int_as_float(__builtin_amdgcn_readlane(zb, wl));
  };

  int cw = -1;
  unsigned long long force = ~0ull;
  Slot ra, rb;
  for (int j = 1; j < m; ++j) {
    const int cws = __builtin_amdgcn_readfirstlane(cw);
    load_slot(cws < 0 ? 0 : cws, ra);
    // 1. cull test and "is the cached arg-max lowered" test, one lane per cell (as in the one-wave kernel)
    const float ax = __builtin_fmaxf(__builtin_fmaxf(lox - sx, sx - hix), 0.f);
    const float ay = __builtin_fmaxf(__builtin_fmaxf(loy - sy, sy - hiy), 0.f);
    const float az = __builtin_fmaxf(__builtin_fmaxf(loz - sz, sz - hiz), 0.f);
    const float lb = ax * ax + ay * ay + az * az;
    const float ex = cwx - sx, ey = cwy - sy, ez = cwz - sz;
    const float dw = ex * ex + ey * ey + ez * ez;
    unsigned long long mask = __ballot(lb <= __int_as_float(cmax));
    unsigned long long fullm = __ballot(dw < __int_as_float(cmax));
    asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(cws));
    mask |= force;
    fullm = (fullm | force) & mask;
    force = 0;
    unsigned long long uset = mask & ~fullm;       // touched, cache unchanged: this wave's slot is updated, nothing else
    unsigned long long rset = fullm;               // cache refreshed (+ the sample's own cell)
    bool own = cws >= 0;
    bool first = true;
    FC_PROF(0);
    do {
      // 2. a batch of <= FX_B refreshed cells: update this wave's slot, publish the slot's (max, arg-max)
      int* const ebase = xw + FX_ENT + ((nx + 1) & 1) * (FX_B * 4 * 8) + wave * 8;
      unsigned long long blist = 0;
      int nb = 0;
      auto part = [&](int c, const Slot& r) {
        const int vi = update_slot(c, r);
        const int Mw = fc_wave_max_i32(vi);
        const unsigned long long e = __ballot(vi == Mw);
        const int wl = __builtin_ctzll(e);
        const int tie = (Mw >= 0 && (e & (e - 1))) ? 256 : 0;
        put_entry(ebase + nb * 32, wl, Mw, soff - lane + wl + tie, r);
        blist |= (unsigned long long)c << (8 * nb);
        ++nb;
      };
      if (own) {
        part(cws, ra);
        own = false;
      }
      while (rset && nb < FX_B) {
        const int c = __builtin_ctzll(rset);
        rset &= rset - 1;
        load_slot(c, rb);
        part(c, rb);
      }
      if (nb) {
        ++nx;
        publish();
      }
      FC_PROF(1);
      // 3. (under the exchange's round trip) the touched cells whose cache stays
      if (first) {
        first = false;
        while (uset) {
          const int c = __builtin_ctzll(uset);
          uset &= uset - 1;
          load_slot(c, rb);
          (void)update_slot(c, rb);
        }
      }
      FC_PROF(2);
      // 4. the cells' new caches from all waves' entries
      if (nb) {
        wait_all();
        FC_PROF(3);
        const int* const rbase = xw + FX_ENT + (nx & 1) * (FX_B * 4 * 8);
        for (int b = 0; b < nb; ++b) {
          const int c = (int)((blist >> (8 * b)) & 255u);
          int M, q;
          float wx, wy, wz;
          bool unique;
          unsigned winners;
          int4 mine;
          get_entries(rbase + b * 32, M, unique, winners, q, wx, wy, wz, mine);
          const bool lane_tie = ((unsigned)__ballot((mine.y & 256) != 0) & winners) != 0u;
          if (__builtin_expect(M >= 0 && (!unique || lane_tie), 0)) {
            // equal maxima: the reference's order = smallest priority, over every wave's points at the maximum
            const int cs = __builtin_amdgcn_readlane(cstart, c);
            Slot r;
            load_slot(c, r);
            unsigned key = 0xffffffffu;
            if (__float_as_int(r.t) == M) key = (fc_prio(sorted_k[cs + soff], L, Q) << 8) | (unsigned)soff;
            const unsigned kmin = fc_wave_min_u32(key);
            const int wl = __builtin_ctzll(__ballot(key == kmin));
            ++tx;
            put_entry(xw + FX_TIE + ((tx & 1) * 4 + wave) * 8, wl, kmin == 0xffffffffu ? FX_NONE + 1 : 0x7fffffff - (int)kmin,
                      (int)(kmin & 255u), r);
            publish();
            wait_all();
            int tk;
            get_entries(xw + FX_TIE + (tx & 1) * 4 * 8, tk, unique, winners, q, wx, wy, wz, mine);
          }
          q &= 255;
          if (M < 0) { q = 0; wx = 0.f; wy = 0.f; wz = 0.f; }
          // (readfirstlane: a no-op for values the compiler already keeps in SGPRs, and the guarantee the "s" operands need)
          M = __builtin_amdgcn_readfirstlane(M);
          q = __builtin_amdgcn_readfirstlane(q);
          wx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wx)));
          wy = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wy)));
          wz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wz)));
          asm volatile(
              "s_mov_b32 m0, %[c]\n\t"
              "s_nop 1\n\t"
              "v_writelane_b32 %[cmax], %[M], m0\n\t"
              "v_writelane_b32 %[cq], %[q], m0\n\t"
              "v_writelane_b32 %[cwx], %[wx], m0\n\t"
              "v_writelane_b32 %[cwy], %[wy], m0\n\t"
              "v_writelane_b32 %[cwz], %[wz], m0"
              : [cmax] "+v"(cmax), [cq] "+v"(cq), [cwx] "+v"(cwx), [cwy] "+v"(cwy), [cwz] "+v"(cwz)
              : [M] "s"(M), [q] "s"(q), [wx] "s"(wx), [wy] "s"(wy), [wz] "s"(wz), [c] "s"(c)
              : "m0");
        }
      }
      FC_PROF(4);
    } while (rset);
    // 5. arg-max over the cells (every wave, identically)
    const int G = fc_wave_max_i32(cmax);
    int wrank;
    if (__builtin_expect(G < 0, 0)) {
      wrank = -1;
      cw = -1;
      sx = p0x; sy = p0y; sz = p0z;
    } else {
      const unsigned long long g = __ballot(cmax == G);
      cw = __builtin_ctzll(g);
      if (__builtin_expect((g & (g - 1)) != 0, 0)) {
        unsigned key = 0xffffffffu;
        if (cmax == G) key = (fc_prio(sorted_k[cstart + cq], L, Q) << 6) | (unsigned)lane;
        key = fc_wave_min_u32(key);
        cw = (int)(key & 63u);
      }
      wrank = __builtin_amdgcn_readlane(cstart, cw) + __builtin_amdgcn_readlane(cq, cw);
      sx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwx), cw));
      sy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwy), cw));
      sz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cwz), cw));
    }
    // 6. results: wave 0, 64 rounds per coalesced store
    if (wave == 0) {
      const int jl = j & 63;
      asm volatile("s_mov_b32 m0, %4\n\ts_nop 1\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                   : "+v"(res), "+v"(resd) : "s"(wrank), "s"(G), "s"(jl) : "m0");
      if ((j & 63) == 63 || j == m - 1) {
        const int jj = (j & ~63) + lane;
        if (jj <= j) {
          idxs[jj] = res < 0 ? 0 : sorted_k[res];
          if (dmax && jj >= 1) dmax[jj] = resd;
        }
      }
    }
    FC_PROF(5);
  }
  if (m == 1 && wave == 0 && lane == 0) idxs[0] = 0;
#ifdef PVN3D_FC_PROF
  if (lane < 8) {
    unsigned v = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) v = lane == p ? fc_acc[p] : v;
    reinterpret_cast<unsigned*>(ws)[wave * 8 + lane] = v;
  }
#endif
}

}  // namespace

// Workspace words (4 bytes each) per cloud for n points; 0 = this size is not served by the cell kernel.
int pvn3d_fps_cells_ws_words(int n) {
  if (n <= 64 || n > 12288) return 0;
  const int spc = (n + 4095) / 4096;
  return 64 * spc * 64 + n;
}

// b clouds of n points (64 < n <= 12288), ws = b * pvn3d_fps_cells_ws_words(n) words.  Returns -1 when the
// shape is not served.
// waves: 0 / 1 = the one-wave kernel (the default: faster, see the multi-wave kernel's header); >= 2 = one wave per
// 64-point slot of a cell (n > 4096: 2 or 3 waves).
int pvn3d_fps_cells_launch(int b, int n, int m, int L, int Q, const float* dataset, int* ws, int* idxs,
                           int* dmax, int waves, hipStream_t st) {
  if (n <= 64 || n > 12288 || !ws) return -1;
  const int spc = (n + 4095) / 4096;
  const size_t lds = (size_t)((spc < 3 ? 4 : 3) * 64 * spc * 64 + FC_AUX_INTS) * sizeof(float);
  static_assert(FX_WORDS <= FC_BINS1, "the exchange area lives in the histogram region");
  if (waves >= 2 && spc >= 2) {
#define FC_LAUNCH_MW(SPC)                                                                      \
  do {                                                                                         \
    auto kern = fps_cells_mw_kernel<SPC>;                                                      \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(kern));                                \
    hipLaunchKernelGGL(kern, dim3(b), dim3(256), lds, st, n, m, L, Q, dataset, ws, idxs, dmax); \
  } while (0)
    if (spc == 2) FC_LAUNCH_MW(2);
    else FC_LAUNCH_MW(3);
#undef FC_LAUNCH_MW
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
#define FC_LAUNCH(SPC)                                                                         \
  do {                                                                                         \
    auto kern = fps_cells_kernel<SPC>;                                                         \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(kern));                                \
    hipLaunchKernelGGL(kern, dim3(b), dim3(256), lds, st, n, m, L, Q, dataset, ws, idxs, dmax); \
  } while (0)
  if (spc == 1) FC_LAUNCH(1);
  else if (spc == 2) FC_LAUNCH(2);
  else FC_LAUNCH(3);
#undef FC_LAUNCH
  PVN3D_LAUNCH_CHECK();
  return 0;
}
