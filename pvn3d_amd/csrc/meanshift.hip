// meanshift.hip -- vote assembly + batched Gaussian mean-shift clustering for gfx950.
//
// Replaces, for a whole batch of fits at once, MeanShiftTorch.fit
// (pvn3d/lib/utils/meanshift_pytorch.py:24-51) and the vote assembly / mask compaction of
// cal_frame_poses(_lm) (pvn3d/lib/utils/pvn3d_eval_utils.py:41-42, 83, 160-175).
//
// The reference materialises (n,n,3) tensors (~150*n^2 bytes of HBM traffic and ~12 launches
// plus a host sync per iteration).  Here one launch per iteration serves EVERY fit of the
// batch: a workgroup owns a tile of seeds of one fit (seed state in VGPRs), streams that fit's
// points through LDS (broadcast reads), and publishes its largest seed shift with one
// atomicMax.  Compulsory HBM traffic is 32*n bytes per iteration, so the kernel is bound by
// fp32 VALU + v_exp_f32 issue, not by HBM.
// Convergence is decided on the device: iteration t of fit s runs iff t == 1 or
// (max_shift[s][t-1] >= thresh and t-1 <= max_iter) -- the reference's
// `if max(Adis) < stop_thresh or it > max_iter: break` (:42) -- so finished fits cost one
// early-exit per workgroup and no host round trip.
//
// Numerics (tolerance 1e-4 on centres): seeds and points are kept relative to
// the fit's first point and pre-scaled by kappa = sqrt(0.5*log2(e))/bw so that the Gaussian
// weight is exp2(-|c'-a'|^2) (one v_exp_f32, constant factor dropped: it cancels in
// sum(w*a)/sum(w)); the exponent is evaluated as 2c'.a' - |a'|^2 - |c'|^2 (9 instead of 11
// VALU instructions per pair; |a'|^2 rides in the unused w lane of the LDS point record).  Centring removes the ~1 m offset of camera-frame coordinates from the fp32
// accumulators.  The neighbour count / labels pass uses the ORIGINAL coordinates and the
// oracle's exact unfused fp32 distance so labels are bit-identical to it.
// This TU is compiled with -ffp-contract=off; FMAs in the hot loop are explicit fmaf().
#include "common.h"

namespace {

constexpr int MS_THREADS = 256;
// points staged in LDS per step: 8 KiB, so that a workgroup of these VALU kernels still fits beside
// a 151 KB fused-MLP workgroup (csrc/sa_mlp.hip) on the same CU when the two halves of the path
// run concurrently (measured: no slowdown vs 16 KiB chunks when running alone)
constexpr int MS_CHUNK = 512;

struct MsState {        // device-side layout inside the caller's workspace
  float4* cbuf[2];      // seed positions, scaled+centred frame, double-buffered
  unsigned* maxshift;   // [n_seg][max_iter+2]  float bits (>= 0 so uint order == float order)
  unsigned* cmmax;      // [n_seg][max_iter+2]  max |c'|^2 over the seeds after iteration t (float bits)
  int* iters;           // [n_seg]
  unsigned long long* best;  // [n_seg]  (count << 32) | ~index
  int* active;          // [2]
  int* counts;          // [total]  pruned neighbour count of every point (core rows: without their n_core core
                        //          neighbours, which ms_argmax_kernel adds)
  int* counts_t;        // [total]  non-core points: their hits among the CORE points, collected column-wise by the count
                        //          kernel from the (core row, non-core column) tests it makes anyway (round 6)
  int* core_idx;        // [total]  per segment: indices of "core" points, ascending
  int* nc_idx;          // [total]  per segment: indices of the other points, ascending
  int* n_core;          // [n_seg]
  // exact early-out of converged seeds (ms_compact_kernel)
  unsigned* frozen_cm;  // [n_seg]  max |c'|^2 over the seeds found to be bitwise fixed points (float bits)
  int* act_cnt;         // [n_seg]  0: no list (every seed is iterated); k + 1: act_idx holds k seeds
  int* act_form;        // [n_seg]  arithmetic form (1 exact, 2 fast) the list was built for
  int* act_idx;         // [total]  per segment: the seeds that are not fixed points under act_form, ascending
  float4* apts;         // [total + 32]  scaled+centred points (x', y', z', -|a'|^2) for the LDS-free iteration kernel
  // winner record (ms_iter kernels): the position of seed max_idx at the first iteration whose update returned its
  // own position bit for bit, and that iteration (0: not yet)
  float4* win_pos;      // [n_seg]
  int* win_it;          // [n_seg]
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline size_t ms_layout(int n_seg, int total, int max_iter, char* base, MsState* st) {
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  const size_t o_c0 = take(sizeof(float4) * (size_t)total);
  const size_t o_c1 = take(sizeof(float4) * (size_t)total);
  const size_t o_aidx = take(sizeof(int) * (size_t)total);
  const size_t o_apts = take(sizeof(float4) * ((size_t)total + 32));
  const size_t o_ms = take(sizeof(unsigned) * (size_t)n_seg * (max_iter + 2));
  const size_t o_cm = take(sizeof(unsigned) * (size_t)n_seg * (max_iter + 2));
  const size_t o_it = take(sizeof(int) * (size_t)n_seg);
  const size_t o_best = take(sizeof(unsigned long long) * (size_t)n_seg);
  const size_t o_act = take(sizeof(int) * 2);
  const size_t o_cnt = take(sizeof(int) * (size_t)total);
  const size_t o_cntt = take(sizeof(int) * (size_t)total);
  const size_t o_core = take(sizeof(int) * (size_t)total);
  const size_t o_nc = take(sizeof(int) * (size_t)total);
  const size_t o_ncore = take(sizeof(int) * (size_t)n_seg);
  const size_t o_fcm = take(sizeof(unsigned) * (size_t)n_seg);
  const size_t o_acnt = take(sizeof(int) * (size_t)n_seg);
  const size_t o_aform = take(sizeof(int) * (size_t)n_seg);
  const size_t o_wpos = take(sizeof(float4) * (size_t)n_seg);
  const size_t o_wit = take(sizeof(int) * (size_t)n_seg);
  if (st) {
    st->cbuf[0] = (float4*)(base + o_c0);
    st->cbuf[1] = (float4*)(base + o_c1);
    st->maxshift = (unsigned*)(base + o_ms);
    st->cmmax = (unsigned*)(base + o_cm);
    st->iters = (int*)(base + o_it);
    st->best = (unsigned long long*)(base + o_best);
    st->active = (int*)(base + o_act);
    st->counts = (int*)(base + o_cnt);
    st->counts_t = (int*)(base + o_cntt);
    st->core_idx = (int*)(base + o_core);
    st->nc_idx = (int*)(base + o_nc);
    st->n_core = (int*)(base + o_ncore);
    st->frozen_cm = (unsigned*)(base + o_fcm);
    st->act_cnt = (int*)(base + o_acnt);
    st->act_form = (int*)(base + o_aform);
    st->act_idx = (int*)(base + o_aidx);
    st->apts = (float4*)(base + o_apts);
    st->win_pos = (float4*)(base + o_wpos);
    st->win_it = (int*)(base + o_wit);
  }
  return off;
}

// ---------------------------------------------------------------------------------------
// vote assembly + order-preserving compaction.  grid: (n_inst, n_kps+1), block 1024.
// Every wave owns one contiguous range of the cloud (ceil(n_pts / 16) rounded up to whole 64-point steps): it counts
// its matches, the sixteen counts are prefixed once, and the wave walks its range again writing at its own running
// offset -- three workgroup barriers per pass instead of four per 1024 points (48 at N = 12 288: the kernel is one
// workgroup's latency, 22-29 us of a 0.7 ms single-frame call).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void vote_compact_kernel(
    int n_pts, int seg_stride, int n_kps, int v_first, const float* __restrict__ pcld,
    const int* __restrict__ mask, const float* __restrict__ ctr_of,
    const float* __restrict__ pred_kp_of, const int* __restrict__ inst_frame,
    const int* __restrict__ inst_cls, const uint8_t* __restrict__ sel, long long sel_inst_stride,
    float4* __restrict__ votes, int* __restrict__ seg_off, int* __restrict__ seg_cnt) {
  __shared__ int s_w1[16], s_w2[16];
  __shared__ int s_nsel;
  const int inst = blockIdx.x, v = v_first + blockIdx.y;
  const int f = inst_frame[inst], cls = inst_cls[inst];
  const int seg = inst * (n_kps + 1) + v;
  const float* P = pcld + (size_t)f * n_pts * 3;
  const int* M = mask + (size_t)f * n_pts;
  const float* O = (v < n_kps) ? pred_kp_of + ((size_t)f * n_kps + v) * n_pts * 3
                               : ctr_of + (size_t)f * n_pts * 3;
  const uint8_t* S = sel ? sel + (size_t)inst * sel_inst_stride : nullptr;
  float4* out = votes + (size_t)seg * seg_stride;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int span = ((n_pts + 15) / 16 + 63) & ~63;          // points per wave
  const int p_lo = w * span, p_hi = min(n_pts, p_lo + span);
  if (tid == 0) s_nsel = 0;
  // pass 1: rows (mask matches) per wave
  int c1 = 0;
  for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
    const int p = p0 + lane;
    c1 += __builtin_popcountll(__ballot(p < p_hi && M[p] == cls));
  }
  if (lane == 0) s_w1[w] = c1;
  __syncthreads();
  int row0 = 0, total = 0;
  for (int i = 0; i < 16; ++i) { const int c = s_w1[i]; row0 += i < w ? c : 0; total += c; }
  // row filter: S[r] refers to the r-th row of the mask-compacted sequence (the labels of an
  // earlier fit on the same instance).  "if ctr_labels.sum() < 1: ctr_labels[0] = 1"
  // (pvn3d_eval_utils.py:86-87,178-179): with no selected row, row 0 is kept.
  bool force_row0 = false;
  int pos0 = row0, kept = total;
  if (S) {
    int mine = 0;
    for (int r = tid; r < total; r += 1024) mine += S[r] ? 1 : 0;
    if (mine) atomicAdd(&s_nsel, mine);
    __syncthreads();
    force_row0 = (s_nsel == 0);
    // pass 2: kept rows per wave
    int c2 = 0, row = row0;
    for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
      const int p = p0 + lane;
      const bool m1 = p < p_hi && M[p] == cls;
      const unsigned long long bal1 = __ballot(m1);
      const int r = row + pvn3d_mbcnt(bal1);
      const bool keep = m1 && (S[r] != 0 || (force_row0 && r == 0));
      c2 += __builtin_popcountll(__ballot(keep));
      row += __builtin_popcountll(bal1);
    }
    if (lane == 0) s_w2[w] = c2;
    __syncthreads();
    pos0 = 0; kept = 0;
    for (int i = 0; i < 16; ++i) { const int c = s_w2[i]; pos0 += i < w ? c : 0; kept += c; }
  }
  // pass 3: write
  int row = row0, pos = pos0;
  for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
    const int p = p0 + lane;
    const bool m1 = p < p_hi && M[p] == cls;
    const unsigned long long bal1 = __ballot(m1);
    bool keep = m1;
    if (S && m1) {
      const int r = row + pvn3d_mbcnt(bal1);
      keep = S[r] != 0 || (force_row0 && r == 0);
    }
    const unsigned long long bal2 = __ballot(keep);
    if (keep)
      out[pos + pvn3d_mbcnt(bal2)] = make_float4(P[p * 3 + 0] - O[p * 3 + 0], P[p * 3 + 1] - O[p * 3 + 1],
                                                 P[p * 3 + 2] - O[p * 3 + 2], 0.f);
    row += __builtin_popcountll(bal1);
    pos += __builtin_popcountll(bal2);
  }
  if (tid == 0) {
    seg_off[seg] = seg * seg_stride;
    seg_cnt[seg] = kept;
  }
}

// ---------------------------------------------------------------------------------------
// one mean-shift iteration for every still-running fit.  grid: (tiles, n_seg), block 256.
// PK = false: one seed per lane, 8 (fast form) or 9 VALU instructions per (seed, point) pair.
// PK = true : two seeds per lane held as float2; the exponent and the four accumulations are packed
//             fp32 instructions (v_pk_fma_f32 / v_pk_add_f32 with the point operand broadcast through
//             op_sel / op_sel_hi -- operand positions chosen by hand, see ms_pair2_packed), 3.5 packed + 1
//             v_exp_f32 per pair instead of 7 + 1.
// SPLIT = false: a workgroup owns 256 * S seeds, every wave walks ALL points of the fit for its seeds.
// SPLIT = true : a workgroup owns 64 * S seeds and its four waves walk one quarter of every staged point
//             chunk each, then add their partial sums through LDS: a workgroup's latency -- the floor
//             of an iteration launch once only a few fits are still running (one wave alone needs 74 us
//             for 3072 points) -- drops four-fold, for four times as many workgroups.  The
//             default (host side, pvn3d_meanshift_fit_batch).
// Winner stop (exact): only C[max_idx] reaches the output (meanshift_pytorch.py:46-51), max_idx depends on the
// ORIGINAL points only (it is computed before the first iteration here), and a seed's trajectory depends on no
// other seed.  The first time the update of seed max_idx returns its own position bit for bit, that position is
// recorded (win_pos / win_it): it is a fixed point of the iteration function, every later iteration would return
// the same bits, so the fit's result is known and its remaining iterations -- which only wait for slower seeds
// (far outliers creeping towards the mode, tens to hundreds of iterations on heavy-tailed votes) to pass the
// reference's stop test -- are not run (stop_on_win; without it they run, for the iteration-count parity tests,
// and the output still comes from the record, so both modes return identical bits).  A fit whose winner never
// lands on a bitwise fixed point stops by the reference's rule as before.
// Canonical summation order (all four variants, so that they give identical bits): per seed four
// partial sums, one per quarter [128 w, 128 w + 128) of every 512-point chunk, each accumulated in
// point order across the chunks; total = (P0 + P1) + (P2 + P3).
// ---------------------------------------------------------------------------------------
typedef float ms_f2 __attribute__((ext_vector_type(2)));
#ifdef MS_FROZEN_PROBE
// tools/ms_frozen_stats.py: [t] = seeds whose iteration-t update is a bitwise fixed point, [512 + t] = seeds iterated
__device__ int g_ms_probe[1024];
#endif

__device__ __forceinline__ ms_f2 ms_fma2(ms_f2 a, ms_f2 b, ms_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ ms_f2 ms_splat(float x) { return ms_f2{x, x}; }

constexpr int MS_QUARTER = MS_CHUNK / 4;

// the four running sums of S seeds (S = 1: element .x only)
struct MsAcc {
  ms_f2 w, x, y, z;
};

// A staged point record is (x', y', -|a'|^2, z'): the register pairs (x', y') and (-|a'|^2, z').
//
// Operand placement of the packed path is hand-written, for a measured reason (round 5, tools/sg_fault_repro.hip
// variant 20, tools/ms_beside_mfma.py, profiles/r05_pk_opsel_fault.txt): on gfx950 a packed-fp32 instruction whose
// LOW half selects the HIGH register of a VGPR pair in its src1 or src2 position (op_sel bit 1 / bit 2) returns that
// operand as +0 in lanes 48-63 now and then WHILE the other wave of its SIMD runs an MFMA / LDS K loop -- i.e. while a
// kernel of the MLP stream shares the SIMD.  (The compiler's own code for ms_splat(a.y) was exactly that form, and a
// MeanShift batch beside the split GEMM changed the y of 2-5 % of its centres by up to 1.5e-5.)  The same selection in
// the src0 position, the opposite one (high half takes the low register, op_sel_hi = 0) in any position, and scalar
// (SGPR) sources never failed in 5e7 executions each.  So: a value that sits in the HIGH register of its pair (y', z')
// always enters as src0, values in LOW registers (x', -|a'|^2) are broadcast with op_sel_hi = 0.  fma(a, b, c) ==
// fma(b, a, c) bit for bit, so the results are those of the one-seed-per-lane path, as before.
template <bool FAST>
__device__ __forceinline__ void ms_pair2_packed(MsAcc& A, const float4 ra, const float4 rb, ms_f2 p2x, ms_f2 p2y, ms_f2 p2z,
                                                ms_f2 pcm) {
  const ms_f2 xa = {ra.x, ra.y}, ya = {ra.z, ra.w}, xb = {rb.x, rb.y}, yb = {rb.z, rb.w};
  ms_f2 ea, eb;
  // -|c'-a'|^2 = 2c'.a' - |a'|^2 - |c'|^2 : (one subtract +) three FMAs
  if (FAST)
    asm("v_pk_fma_f32 %[ea], %[p2x], %[xa], %[ya] op_sel_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %[eb], %[p2x], %[xb], %[yb] op_sel_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %[ea], %[xa], %[p2y], %[ea] op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %[eb], %[xb], %[p2y], %[eb] op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %[ea], %[ya], %[p2z], %[ea] op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %[eb], %[yb], %[p2z], %[eb] op_sel:[1,0,0]"
        : [ea] "=&v"(ea), [eb] "=&v"(eb)
        : [p2x] "v"(p2x), [p2y] "v"(p2y), [p2z] "v"(p2z), [xa] "v"(xa), [ya] "v"(ya), [xb] "v"(xb), [yb] "v"(yb));
  else
    asm("v_pk_add_f32 %[ea], %[ya], %[pcm] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[eb], %[yb], %[pcm] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[p2x], %[xa], %[ea] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[eb], %[p2x], %[xb], %[eb] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[xa], %[p2y], %[ea] op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %[eb], %[xb], %[p2y], %[eb] op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %[ea], %[ya], %[p2z], %[ea] op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %[eb], %[yb], %[p2z], %[eb] op_sel:[1,0,0]"
        : [ea] "=&v"(ea), [eb] "=&v"(eb)
        : [p2x] "v"(p2x), [p2y] "v"(p2y), [p2z] "v"(p2z), [pcm] "v"(pcm), [xa] "v"(xa), [ya] "v"(ya), [xb] "v"(xb),
          [yb] "v"(yb));
  const ms_f2 wa = ms_f2{__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const ms_f2 wb = ms_f2{__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)};
  asm("s_nop 0\n\t"                                         // v_exp_f32 (trans) -> VALU read of its result
      "v_pk_add_f32 %[Aw], %[Aw], %[wa]\n\t"
      "v_pk_fma_f32 %[Ax], %[wa], %[xa], %[Ax] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[Ay], %[xa], %[wa], %[Ay] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[Az], %[ya], %[wa], %[Az] op_sel:[1,0,0]\n\t"
      "v_pk_add_f32 %[Aw], %[Aw], %[wb]\n\t"
      "v_pk_fma_f32 %[Ax], %[wb], %[xb], %[Ax] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[Ay], %[xb], %[wb], %[Ay] op_sel:[1,0,0]\n\t"
      "v_pk_fma_f32 %[Az], %[yb], %[wb], %[Az] op_sel:[1,0,0]"
      : [Aw] "+v"(A.w), [Ax] "+v"(A.x), [Ay] "+v"(A.y), [Az] "+v"(A.z)
      : [wa] "v"(wa), [wb] "v"(wb), [xa] "v"(xa), [ya] "v"(ya), [xb] "v"(xb), [yb] "v"(yb));
}

// one seed per lane: record fields as single registers, nothing to select
template <bool FAST>
__device__ __forceinline__ void ms_pair_scalar(MsAcc& A, const float4 r, ms_f2 p2x, ms_f2 p2y, ms_f2 p2z, ms_f2 pcm) {
  const float ax = r.x, ay = r.y, aw = r.z, az = r.w;
  const float e0 = FAST ? aw : aw - pcm.x;
  const float e = fmaf(p2z.x, az, fmaf(p2y.x, ay, fmaf(p2x.x, ax, e0)));
  const float w = __builtin_amdgcn_exp2f(e);
  A.w.x += w;
  A.x.x = fmaf(w, ax, A.x.x);
  A.y.x = fmaf(w, ay, A.y.x);
  A.z.x = fmaf(w, az, A.z.x);
}

template <bool PK, bool FAST>
__device__ __forceinline__ void ms_four(MsAcc& A, const float4 r0, const float4 r1, const float4 r2, const float4 r3,
                                        ms_f2 p2x, ms_f2 p2y, ms_f2 p2z, ms_f2 pcm) {
  if (PK) {
    ms_pair2_packed<FAST>(A, r0, r1, p2x, p2y, p2z, pcm);
    ms_pair2_packed<FAST>(A, r2, r3, p2x, p2y, p2z, pcm);
  } else {
    ms_pair_scalar<FAST>(A, r0, p2x, p2y, p2z, pcm);
    ms_pair_scalar<FAST>(A, r1, p2x, p2y, p2z, pcm);
    ms_pair_scalar<FAST>(A, r2, p2x, p2y, p2z, pcm);
    ms_pair_scalar<FAST>(A, r3, p2x, p2y, p2z, pcm);
  }
}

// [q_begin, q_end) is a whole number of 4-point groups (the callers pad the chunk to a multiple of four).  The next
// group is read from LDS before the current one is evaluated: when a launch has one wave per SIMD (a single frame's
// fits, the tail of a heavy batch) nothing else hides the read's ~100 cycles, a third of a group's time.
template <bool PK, bool FAST>
__device__ __forceinline__ void ms_accumulate(MsAcc& A, const float4* __restrict__ sp, int q_begin, int q_end,
                                              ms_f2 p2x, ms_f2 p2y, ms_f2 p2z, ms_f2 pcm) {
  if (q_begin >= q_end) return;
  float4 a0 = sp[q_begin], a1 = sp[q_begin + 1], a2 = sp[q_begin + 2], a3 = sp[q_begin + 3];
  float4 b0, b1, b2, b3;
  for (int q = q_begin; q < q_end; q += 8) {      // two register sets in turn: no copies
    const int qb = min(q + 4, q_end - 4);         // the last group re-reads itself (unused)
    b0 = sp[qb]; b1 = sp[qb + 1]; b2 = sp[qb + 2]; b3 = sp[qb + 3];
    __builtin_amdgcn_sched_barrier(0);            // or the scheduler sinks the reads to the end of the block again
    ms_four<PK, FAST>(A, a0, a1, a2, a3, p2x, p2y, p2z, pcm);
    if (q + 4 >= q_end) break;
    const int qa = min(q + 8, q_end - 4);
    a0 = sp[qa]; a1 = sp[qa + 1]; a2 = sp[qa + 2]; a3 = sp[qa + 3];
    __builtin_amdgcn_sched_barrier(0);
    ms_four<PK, FAST>(A, b0, b1, b2, b3, p2x, p2y, p2z, pcm);
  }
}

// (a + b) + (c + d) of the four quarter sums.  With one seed per lane only element .x is live and the compiler's SLP
// pass packs the four independent scalar chains across the accumulators, shuffling through op_sel (the operand form
// of the note above, found by tools/pk_opsel_lint.py): there the three additions are single-lane instructions.
__device__ __forceinline__ float ms_add1(float a, float b) {
  float r;
  asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool PK>
__device__ __forceinline__ ms_f2 ms_sum4(ms_f2 a, ms_f2 b, ms_f2 c, ms_f2 d) {
  if (PK) return (a + b) + (c + d);
  return ms_f2{ms_add1(ms_add1(a.x, b.x), ms_add1(c.x, d.x)), 0.f};
}

template <bool PK, bool SPLIT>
__global__ __launch_bounds__(MS_THREADS) void ms_iter_kernel(
    const float4* __restrict__ pts, const int* __restrict__ seg_off,
    const int* __restrict__ seg_cnt, const float4* __restrict__ cin, float4* __restrict__ cout,
    unsigned* __restrict__ maxshift, unsigned* __restrict__ cmmax, int* __restrict__ iters, int t,
    int max_iter, float thresh, float kappa, float inv_kappa, unsigned* __restrict__ frozen_cm,
    const int* __restrict__ act_cnt, const int* __restrict__ act_form, const int* __restrict__ act_idx,
    const unsigned long long* __restrict__ best, float4* __restrict__ win_pos, int* __restrict__ win_it,
    int stop_on_win) {
  constexpr int S = PK ? 2 : 1;
  constexpr int LANES = SPLIT ? 64 : MS_THREADS;     // distinct seed lanes of the workgroup
  __shared__ float4 s_pts[MS_CHUNK];
  __shared__ float s_red[3][MS_THREADS / 64];
  __shared__ MsAcc s_part[SPLIT ? 3 : 1][SPLIT ? 64 : 1];
  // grid (fits, tiles): the fit index is the FAST one.  The grid is sized for the host's bound on the vote count (all N points
  // of a cloud); with (tiles, fits) every fit's real tiles were followed by its empty ones -- 24 real + 72 empty at 3072 of
  // 12288 votes -- and that periodic pattern left a quarter of the chip without real workgroups (the dispatcher deals
  // consecutive workgroups round-robin over XCDs and shader engines: 1.16 ms per iteration against 0.92 ms with 97 tiles
  // per fit, tools/ms_rate_real2.py).  Tile-major, the empty workgroups are the tail of the grid.
  const int seg = blockIdx.x;
  const int n = seg_cnt[seg];
  const int tile0 = blockIdx.y * (LANES * S);
  if (tile0 >= n) return;
  unsigned* ms = maxshift + (size_t)seg * (max_iter + 2);
  if (t > 1) {
    const float prev = __uint_as_float(ms[t - 1]);
    if (!(prev >= thresh) || (t - 1) > max_iter) return;  // converged / capped (:42)
  }
  const int won = win_it[seg];
  if (stop_on_win && won) return;                         // the result is known (winner stop)
  const int max_idx = (int)(~(unsigned)(best[seg] & 0xffffffffULL));
  unsigned* cmx = cmmax + (size_t)seg * (max_iter + 2);
  // The weight exp2(-|c'-a'|^2) = exp2(2c'.a' - |a'|^2) * exp2(-|c'|^2) and the last factor is
  // constant per seed, so it cancels in new_c = sum(w a) / sum(w): when every seed of the fit has
  // |c'|^2 <= 64 (no overflow: the largest weight is exp2(|c'|^2)) the per-pair subtraction of
  // |c'|^2 is dropped.  The bound comes from the previous iteration's output (seeds that are no longer
  // iterated -- below -- through frozen_cm); iteration 1 always takes the exact form.
  const bool fast = t > 1 && fmaxf(__uint_as_float(cmx[t - 1]), __uint_as_float(frozen_cm[seg])) <= 64.f;
  // Exact early-out: a seed whose update returned its own position bit for bit is a fixed point of this
  // iteration function -- it would return the same bits in every later iteration of the same arithmetic form,
  // shift 0 -- so it is dropped from the seed list (ms_compact_kernel) and only the others are iterated.  Only
  // C[max_idx] and the per-iteration maximum shift reach the output (meanshift_pytorch.py:42-51), and both
  // position buffers hold a fixed point's position.  A list built for the other form is ignored.
  const int form = fast ? 2 : 1;
  const int lc = act_cnt[seg];
  const bool use_list = lc > 0 && act_form[seg] == form;
  const int n_eff = use_list ? lc - 1 : n;
  if (tile0 >= n_eff) return;
#ifdef MS_FROZEN_PROBE
  if (threadIdx.x == 0 && t < 256) atomicAdd(&g_ms_probe[768 + t], 1);      // workgroups that do work
#endif
  const int base = seg_off[seg];
  const float4 org = pts[base];  // frame origin: the fit's first point
  const int tid = threadIdx.x;
  const int sl = SPLIT ? (tid & 63) : tid;            // seed lane
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float cx[S], cy[S], cz[S];
  int sid[S];             // seed index of this lane's seeds (-1: none)
  ms_f2 p2x = ms_splat(0.f), p2y = ms_splat(0.f), p2z = ms_splat(0.f), pcm = ms_splat(0.f);
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int j = tile0 + s * LANES + sl;
    const int i = j < n_eff ? (use_list ? act_idx[base + j] : j) : -1;
    sid[s] = i;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i >= 0) {
      if (t == 1) {
        const float4 a = pts[base + i];
        c = make_float4((a.x - org.x) * kappa, (a.y - org.y) * kappa, (a.z - org.z) * kappa, 0.f);
      } else {
        c = cin[base + i];
      }
    }
    cx[s] = c.x; cy[s] = c.y; cz[s] = c.z;
    p2x[s] = 2.f * c.x; p2y[s] = 2.f * c.y; p2z[s] = 2.f * c.z;
    pcm[s] = fmaf(c.z, c.z, fmaf(c.y, c.y, c.x * c.x));
  }
  constexpr int NACC = SPLIT ? 1 : 4;
  MsAcc acc[NACC];
#pragma unroll
  for (int w = 0; w < NACC; ++w) acc[w].w = acc[w].x = acc[w].y = acc[w].z = ms_splat(0.f);

  // The next chunk's points are requested (clamped address, unconditional) before the current chunk is walked: when
  // only a few workgroups still run -- the long tail of a heavy-tailed batch -- a launch lasts as long as ONE
  // workgroup, and six exposed global round trips (3072 points) were a third of that.
  constexpr int PPT = MS_CHUNK / MS_THREADS;
  float4 pre[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) pre[u] = pts[base + min(tid + u * MS_THREADS, n - 1)];
  for (int j0 = 0; j0 < n; j0 += MS_CHUNK) {
    const int cnt = min(MS_CHUNK, n - j0);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int q = tid + u * MS_THREADS;
      float4 a;
      if (q < cnt) {
        const float4 r = pre[u];
        const float ax = (r.x - org.x) * kappa, ay = (r.y - org.y) * kappa,
                    az = (r.z - org.z) * kappa;
        a = make_float4(ax, ay, -fmaf(az, az, fmaf(ay, ay, ax * ax)), az);   // (x', y', -|a'|^2, z')
      } else {
        a = make_float4(0.f, 0.f, -1e30f, 0.f);  // exp2(-huge) == 0: padded rows weigh 0
      }
      s_pts[q] = a;
    }
    if (j0 + MS_CHUNK < n) {
#pragma unroll
      for (int u = 0; u < PPT; ++u) pre[u] = pts[base + min(j0 + MS_CHUNK + tid + u * MS_THREADS, n - 1)];
    }
    __syncthreads();
    const int cnt4 = (cnt + 3) & ~3;
    if (SPLIT) {
      const int qb = wave * MS_QUARTER, qe = min(qb + MS_QUARTER, cnt4);
      if (fast) ms_accumulate<PK, true>(acc[0], s_pts, qb, qe, p2x, p2y, p2z, pcm);
      else ms_accumulate<PK, false>(acc[0], s_pts, qb, qe, p2x, p2y, p2z, pcm);
    } else {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int qb = w * MS_QUARTER, qe = min(qb + MS_QUARTER, cnt4);
        if (fast) ms_accumulate<PK, true>(acc[w], s_pts, qb, qe, p2x, p2y, p2z, pcm);
        else ms_accumulate<PK, false>(acc[w], s_pts, qb, qe, p2x, p2y, p2z, pcm);
      }
    }
  }
  // total = (P0 + P1) + (P2 + P3)
  MsAcc tot;
  if (SPLIT) {
    if (wave > 0) s_part[wave - 1][sl] = acc[0];
    __syncthreads();
    if (wave > 0) return;
    const MsAcc p1 = s_part[0][sl], p2 = s_part[1][sl], p3 = s_part[2][sl];
    tot.w = ms_sum4<PK>(acc[0].w, p1.w, p2.w, p3.w);
    tot.x = ms_sum4<PK>(acc[0].x, p1.x, p2.x, p3.x);
    tot.y = ms_sum4<PK>(acc[0].y, p1.y, p2.y, p3.y);
    tot.z = ms_sum4<PK>(acc[0].z, p1.z, p2.z, p3.z);
  } else {
    tot.w = (acc[0].w + acc[NACC > 1 ? 1 : 0].w) + (acc[NACC > 2 ? 2 : 0].w + acc[NACC > 3 ? 3 : 0].w);
    tot.x = (acc[0].x + acc[NACC > 1 ? 1 : 0].x) + (acc[NACC > 2 ? 2 : 0].x + acc[NACC > 3 ? 3 : 0].x);
    tot.y = (acc[0].y + acc[NACC > 1 ? 1 : 0].y) + (acc[NACC > 2 ? 2 : 0].y + acc[NACC > 3 ? 3 : 0].y);
    tot.z = (acc[0].z + acc[NACC > 1 ? 1 : 0].z) + (acc[NACC > 2 ? 2 : 0].z + acc[NACC > 3 ? 3 : 0].z);
  }

  float mshift = 0.f, mcm = 0.f, fcm = 0.f;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int i = sid[s];
    if (i >= 0) {
      const float inv = 1.0f / tot.w[s];
      const float nx = tot.x[s] * inv, ny = tot.y[s] * inv, nz = tot.z[s] * inv;
      const float ex = nx - cx[s], ey = ny - cy[s], ez = nz - cz[s];
      const float sh = sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex))) * inv_kappa;
      mshift = fmaxf(mshift, sh);
      const float ncm = fmaf(nz, nz, fmaf(ny, ny, nx * nx));
      mcm = (ncm <= mcm) ? mcm : ((ncm != ncm) ? __builtin_inff() : ncm);   // NaN counts as +inf
      // .w = the form under which the seed is a bitwise fixed point (0: it moved)
      const bool fixed = nx == cx[s] && ny == cy[s] && nz == cz[s];
      cout[base + i] = make_float4(nx, ny, nz, fixed ? (float)form : 0.f);
      if (fixed) fcm = fmaxf(fcm, ncm);
      if (fixed && i == max_idx && !won) {       // one thread of one workgroup owns seed max_idx
        win_pos[seg] = make_float4(nx, ny, nz, 0.f);
        win_it[seg] = t;
      }
#ifdef MS_FROZEN_PROBE
      if (t < 512) {
        atomicAdd(&g_ms_probe[512 + t], 1);
        if (nx == cx[s] && ny == cy[s] && nz == cz[s]) atomicAdd(&g_ms_probe[t], 1);
      }
#endif
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    mshift = fmaxf(mshift, __shfl_xor(mshift, o, 64));
    mcm = fmaxf(mcm, __shfl_xor(mcm, o, 64));
    fcm = fmaxf(fcm, __shfl_xor(fcm, o, 64));
  }
  if (SPLIT) {          // one wave left
    if (tid == 0) {
      atomicMax(ms + t, __float_as_uint(mshift));
      atomicMax(cmx + t, __float_as_uint(mcm));
      if (fcm > 0.f) atomicMax(frozen_cm + seg, __float_as_uint(fcm));
      atomicMax(iters + seg, t);
    }
    return;
  }
  if ((tid & 63) == 0) { s_red[0][tid >> 6] = mshift; s_red[1][tid >> 6] = mcm; s_red[2][tid >> 6] = fcm; }
  __syncthreads();
  if (tid == 0) {
    float m = s_red[0][0], c = s_red[1][0], f = s_red[2][0];
    for (int i = 1; i < MS_THREADS / 64; ++i) {
      m = fmaxf(m, s_red[0][i]); c = fmaxf(c, s_red[1][i]); f = fmaxf(f, s_red[2][i]);
    }
    atomicMax(ms + t, __float_as_uint(m));
    atomicMax(cmx + t, __float_as_uint(c));
    if (f > 0.f) atomicMax(frozen_cm + seg, __float_as_uint(f));
    atomicMax(iters + seg, t);
  }
}

// ---------------------------------------------------------------------------------------
// LDS-free form of the iteration (PVN3D_MS_SGPR_POINTS): for running BESIDE the fused-MLP kernels.  Those keep
// the LDS pipe busy with MFMA fragment reads, and ms_iter_kernel's broadcast ds_read_b128 per point queues behind
// them (tools/coexec_probe.py: a VALU kernel with one broadcast LDS read per 8 VALU instructions hides 0 % of
// its time under the MLP forward, the same kernel without the read 78 %).  A point is wave-uniform, so here it is
// an SGPR operand: ms_prep_kernel writes the scaled, centred records once per call, and a wave streams them
// with s_load_dwordx16 (4 points each), two 8-point blocks in flight.  One wave = one tile of 128 seeds, no
// barrier, no LDS; the grid is a fixed number of waves that stride over the (fit, tile) items, so the caller
// chooses how many SIMD slots the kernel takes (PVN3D_MS_WAVE_CAP).  Same canonical summation order, same bits.
// The loads are inline asm: SMEM returns out of order, so the only wait is lgkmcnt(0), and it has to stand
// BEFORE the next block's loads are issued -- the compiler's own placement (at first use) would wait for the
// prefetch it has just issued.
// ---------------------------------------------------------------------------------------
typedef float ms_f16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(MS_THREADS) void ms_prep_kernel(const float4* __restrict__ pts,
                                                             const int* __restrict__ seg_off,
                                                             const int* __restrict__ seg_cnt, float kappa,
                                                             float4* __restrict__ apts) {
  const int seg = blockIdx.y;
  const int n = seg_cnt[seg];
  const int q = blockIdx.x * MS_THREADS + threadIdx.x;
  if (q >= ((n + 3) & ~3)) return;
  const int base = seg_off[seg];
  const float4 org = pts[base];
  float4 a = make_float4(0.f, -1e30f, 0.f, 0.f);   // padded rows weigh exp2(-huge) == 0 (as in ms_iter_kernel)
  if (q < n) {
    const float4 r = pts[base + q];
    const float ax = (r.x - org.x) * kappa, ay = (r.y - org.y) * kappa, az = (r.z - org.z) * kappa;
    a = make_float4(ax, -fmaf(az, az, fmaf(ay, ay, ax * ax)), ay, az);   // record = (x', -|a'|^2, y', z')
  }
  apts[base + q] = a;
}

#define MS_SLOAD(d0, d1, ptr)                                                                           \
  asm volatile("s_load_dwordx16 %0, %2, 0x0\n\ts_load_dwordx16 %1, %2, 0x40" : "=&s"(d0), "=&s"(d1) : "s"(ptr))
#define MS_SWAIT(d0, d1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(d0), "+s"(d1))

// Two points (A then B) into the running sums of a lane's two seeds.  A record is two SGPR pairs, (x', -|a'|^2) and
// (y', z'), so that every packed instruction names ONE pair (the constant bus carries one scalar operand per VALU
// instruction) and broadcasts the half it needs through op_sel -- the fast form's first FMA takes both of its scalar
// operands from the same pair.  The statements are volatile so that loads, waits and arithmetic stay in the written
// order; the two points are interleaved for a lone wave (dependent issue costs 8 cycles, independent 5).
template <bool FAST>
__device__ __forceinline__ void ms_acc2(MsAcc& A, ms_f2 xa, ms_f2 ya, ms_f2 xb, ms_f2 yb, ms_f2 p2x, ms_f2 p2y,
                                        ms_f2 p2z, ms_f2 pcm) {
  ms_f2 ea, eb;
  if (FAST)
    asm volatile(
        "v_pk_fma_f32 %[ea], %[p2x], %[xa], %[xa] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[eb], %[p2x], %[xb], %[xb] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[p2y], %[ya], %[ea] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[eb], %[p2y], %[yb], %[eb] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[p2z], %[ya], %[ea] op_sel:[0,1,0]\n\t"
        "v_pk_fma_f32 %[eb], %[p2z], %[yb], %[eb] op_sel:[0,1,0]"
        : [ea] "=&v"(ea), [eb] "=&v"(eb)
        : [p2x] "v"(p2x), [p2y] "v"(p2y), [p2z] "v"(p2z), [xa] "s"(xa), [ya] "s"(ya), [xb] "s"(xb), [yb] "s"(yb));
  else
    asm volatile(
        "v_pk_add_f32 %[ea], %[xa], %[pcm] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_add_f32 %[eb], %[xb], %[pcm] op_sel:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[p2x], %[xa], %[ea] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[eb], %[p2x], %[xb], %[eb] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[p2y], %[ya], %[ea] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[eb], %[p2y], %[yb], %[eb] op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %[ea], %[p2z], %[ya], %[ea] op_sel:[0,1,0]\n\t"
        "v_pk_fma_f32 %[eb], %[p2z], %[yb], %[eb] op_sel:[0,1,0]"
        : [ea] "=&v"(ea), [eb] "=&v"(eb)
        : [p2x] "v"(p2x), [p2y] "v"(p2y), [p2z] "v"(p2z), [pcm] "v"(pcm), [xa] "s"(xa), [ya] "s"(ya), [xb] "s"(xb),
          [yb] "s"(yb));
  const ms_f2 wa = ms_f2{__builtin_amdgcn_exp2f(ea.x), __builtin_amdgcn_exp2f(ea.y)};
  const ms_f2 wb = ms_f2{__builtin_amdgcn_exp2f(eb.x), __builtin_amdgcn_exp2f(eb.y)};
  asm volatile(
      "s_nop 0\n\t"                                         // v_exp_f32 (trans) -> VALU read of its result
      "v_pk_add_f32 %[Aw], %[Aw], %[wa]\n\t"
      "v_pk_fma_f32 %[Ax], %[wa], %[xa], %[Ax] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[Ay], %[wa], %[ya], %[Ay] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[Az], %[wa], %[ya], %[Az] op_sel:[0,1,0]\n\t"
      "v_pk_add_f32 %[Aw], %[Aw], %[wb]\n\t"
      "v_pk_fma_f32 %[Ax], %[wb], %[xb], %[Ax] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[Ay], %[wb], %[yb], %[Ay] op_sel_hi:[1,0,1]\n\t"
      "v_pk_fma_f32 %[Az], %[wb], %[yb], %[Az] op_sel:[0,1,0]"
      : [Aw] "+v"(A.w), [Ax] "+v"(A.x), [Ay] "+v"(A.y), [Az] "+v"(A.z)
      : [wa] "v"(wa), [wb] "v"(wb), [xa] "s"(xa), [ya] "s"(ya), [xb] "s"(xb), [yb] "s"(yb));
}

#define MS_PAIR(b, k) __builtin_shufflevector(b, b, k, (k) + 1)
template <bool FAST>
__device__ __forceinline__ void ms_acc4(MsAcc& A, const ms_f16 b, ms_f2 p2x, ms_f2 p2y, ms_f2 p2z, ms_f2 pcm) {
  ms_acc2<FAST>(A, MS_PAIR(b, 0), MS_PAIR(b, 2), MS_PAIR(b, 4), MS_PAIR(b, 6), p2x, p2y, p2z, pcm);
  ms_acc2<FAST>(A, MS_PAIR(b, 8), MS_PAIR(b, 10), MS_PAIR(b, 12), MS_PAIR(b, 14), p2x, p2y, p2z, pcm);
}

// One quarter (<= 128 points, a multiple of 4) of the canonical order into A; ap points at its first record.
// No load is in flight on entry or on exit: between MS_SLOAD and MS_SWAIT the destination registers are
// undefined as far as the hardware is concerned but live as far as the compiler is, so that window must hold
// nothing but the statements below (a compiler-generated SGPR spill or copy of a0/a1 there would save stale
// data -- checked in the ISA: the loop bodies hold no v_writelane/s_mov of the buffers).  The look-ahead of the
// last full step fetches the next quarter's first block: wasted, but it makes that quarter's first load a
// scalar-cache hit.
template <bool FAST>
__device__ __forceinline__ void ms_sgpr_quarter(MsAcc& A, const float4* ap, int pe, ms_f2 p2x, ms_f2 p2y, ms_f2 p2z,
                                                ms_f2 pcm) {
  ms_f16 a0, a1, b0, b1;
  int left = pe;
  MS_SLOAD(a0, a1, ap);
  for (; left >= 16; left -= 16) {
    MS_SWAIT(a0, a1);
    MS_SLOAD(b0, b1, ap + 8);
    ms_acc4<FAST>(A, a0, p2x, p2y, p2z, pcm);
    ms_acc4<FAST>(A, a1, p2x, p2y, p2z, pcm);
    MS_SWAIT(b0, b1);
    MS_SLOAD(a0, a1, ap + 16);
    ms_acc4<FAST>(A, b0, p2x, p2y, p2z, pcm);
    ms_acc4<FAST>(A, b1, p2x, p2y, p2z, pcm);
    ap += 16;
  }
  MS_SWAIT(a0, a1);
  if (left > 0) {                   // the fit's last quarter only: 4, 8 or 12 points
    ms_acc4<FAST>(A, a0, p2x, p2y, p2z, pcm);
    if (left >= 8) ms_acc4<FAST>(A, a1, p2x, p2y, p2z, pcm);
    if (left >= 12) {
      MS_SLOAD(b0, b1, ap + 8);
      MS_SWAIT(b0, b1);
      ms_acc4<FAST>(A, b0, p2x, p2y, p2z, pcm);
    }
  }
}

__global__ __launch_bounds__(64) void ms_iter_sgpr_kernel(
    const float4* __restrict__ pts, const float4* __restrict__ apts, const int* __restrict__ seg_off,
    const int* __restrict__ seg_cnt, const float4* __restrict__ cin, float4* __restrict__ cout,
    unsigned* __restrict__ maxshift, unsigned* __restrict__ cmmax, int* __restrict__ iters, int t,
    int max_iter, float thresh, float kappa, float inv_kappa, unsigned* __restrict__ frozen_cm,
    const int* __restrict__ act_cnt, const int* __restrict__ act_form, const int* __restrict__ act_idx,
    int tiles_per_seg, int n_items, const unsigned long long* __restrict__ best, float4* __restrict__ win_pos,
    int* __restrict__ win_it, int stop_on_win) {
  constexpr int S = 2;
  const int sl = threadIdx.x;
  // item -> (tile, fit), the fit index fast: the tiles beyond a fit's vote count (the grid is sized for the host's bound,
  // all N points of a cloud) are the TAIL of the grid instead of a periodic pattern inside it (see ms_iter_kernel)
  const int n_seg = n_items / tiles_per_seg;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
    const int tile = item / n_seg;
    const int seg = item - tile * n_seg;
    const int tile0 = tile * (64 * S);
    const int n = seg_cnt[seg];
    if (tile0 >= n) continue;
    unsigned* ms = maxshift + (size_t)seg * (max_iter + 2);
    if (t > 1) {
      const float prev = __uint_as_float(ms[t - 1]);
      if (!(prev >= thresh) || (t - 1) > max_iter) continue;
    }
    const int won = win_it[seg];
    if (stop_on_win && won) continue;
    const int max_idx = (int)(~(unsigned)(best[seg] & 0xffffffffULL));
    unsigned* cmx = cmmax + (size_t)seg * (max_iter + 2);
    const bool fast = t > 1 && fmaxf(__uint_as_float(cmx[t - 1]), __uint_as_float(frozen_cm[seg])) <= 64.f;
    const int form = fast ? 2 : 1;
    const int lc = act_cnt[seg];
    const bool use_list = lc > 0 && act_form[seg] == form;
    const int n_eff = use_list ? lc - 1 : n;
    if (tile0 >= n_eff) continue;
    const int base = __builtin_amdgcn_readfirstlane(seg_off[seg]);
    float cx[S], cy[S], cz[S];
    int sid[S];
    ms_f2 p2x = ms_splat(0.f), p2y = ms_splat(0.f), p2z = ms_splat(0.f), pcm = ms_splat(0.f);
    const float4* ap = apts + base;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int j = tile0 + s * 64 + sl;
      const int i = j < n_eff ? (use_list ? act_idx[base + j] : j) : -1;
      sid[s] = i;
      float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i >= 0) {
        if (t == 1) {
          const float4 org = pts[base];
          const float4 a = pts[base + i];
          c = make_float4((a.x - org.x) * kappa, (a.y - org.y) * kappa, (a.z - org.z) * kappa, 0.f);
        } else {
          c = cin[base + i];
        }
      }
      cx[s] = c.x; cy[s] = c.y; cz[s] = c.z;
      p2x[s] = 2.f * c.x; p2y[s] = 2.f * c.y; p2z[s] = 2.f * c.z;
      pcm[s] = fmaf(c.z, c.z, fmaf(c.y, c.y, c.x * c.x));
    }
    // acc[0] always receives the current quarter; the four sums rotate after each one, and once more at the end so
    // that acc[w] is quarter w's sum again (one copy of the hot loop instead of four)
    MsAcc acc[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) acc[w].w = acc[w].x = acc[w].y = acc[w].z = ms_splat(0.f);
    const int cnt4 = (n + 3) & ~3;
    int nq = 0;
    for (int q0 = 0; q0 < cnt4; q0 += MS_QUARTER, ++nq) {
      const int pe = min(MS_QUARTER, cnt4 - q0);
      if (fast) ms_sgpr_quarter<true>(acc[0], ap + q0, pe, p2x, p2y, p2z, pcm);
      else ms_sgpr_quarter<false>(acc[0], ap + q0, pe, p2x, p2y, p2z, pcm);
      const MsAcc r = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = acc[3]; acc[3] = r;
    }
    for (int k = (4 - (nq & 3)) & 3; k > 0; --k) {
      const MsAcc r = acc[0]; acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = acc[3]; acc[3] = r;
    }
    MsAcc tot;
    tot.w = (acc[0].w + acc[1].w) + (acc[2].w + acc[3].w);
    tot.x = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x);
    tot.y = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
    tot.z = (acc[0].z + acc[1].z) + (acc[2].z + acc[3].z);
    float mshift = 0.f, mcm = 0.f, fcm = 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int i = sid[s];
      if (i >= 0) {
        const float inv = 1.0f / tot.w[s];
        const float nx = tot.x[s] * inv, ny = tot.y[s] * inv, nz = tot.z[s] * inv;
        const float ex = nx - cx[s], ey = ny - cy[s], ez = nz - cz[s];
        const float sh = sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex))) * inv_kappa;
        mshift = fmaxf(mshift, sh);
        const float ncm = fmaf(nz, nz, fmaf(ny, ny, nx * nx));
        mcm = (ncm <= mcm) ? mcm : ((ncm != ncm) ? __builtin_inff() : ncm);
        const bool fixed = nx == cx[s] && ny == cy[s] && nz == cz[s];
        cout[base + i] = make_float4(nx, ny, nz, fixed ? (float)form : 0.f);
        if (fixed) fcm = fmaxf(fcm, ncm);
        if (fixed && i == max_idx && !won) {
          win_pos[seg] = make_float4(nx, ny, nz, 0.f);
          win_it[seg] = t;
        }
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      mshift = fmaxf(mshift, __shfl_xor(mshift, o, 64));
      mcm = fmaxf(mcm, __shfl_xor(mcm, o, 64));
      fcm = fmaxf(fcm, __shfl_xor(fcm, o, 64));
    }
    if (sl == 0) {
      atomicMax(ms + t, __float_as_uint(mshift));
      atomicMax(cmx + t, __float_as_uint(mcm));
      if (fcm > 0.f) atomicMax(frozen_cm + seg, __float_as_uint(fcm));
      atomicMax(iters + seg, t);
    }
  }
}

// Seed list for the iterations after t: the seeds of every still-running fit that are not bitwise fixed
// points under the arithmetic form iteration t+1 will use, in ascending order.  grid (n_seg), block 1024.
__global__ __launch_bounds__(1024) void ms_compact_kernel(
    const int* __restrict__ seg_off, const int* __restrict__ seg_cnt, const float4* __restrict__ cur,
    const unsigned* __restrict__ maxshift, const unsigned* __restrict__ cmmax,
    const unsigned* __restrict__ frozen_cm, int t, int max_iter, float thresh, int* __restrict__ act_cnt,
    int* __restrict__ act_form, int* __restrict__ act_idx, const int* __restrict__ win_it, int stop_on_win) {
  __shared__ int s_wave[16];
  __shared__ int s_run;
  const int seg = blockIdx.x;
  const int n = seg_cnt[seg];
  if (n <= 0) return;
  const float prev = __uint_as_float(maxshift[(size_t)seg * (max_iter + 2) + t]);
  if (!(prev >= thresh) || t > max_iter) return;        // no iteration t+1 for this fit
  if (stop_on_win && win_it[seg]) return;
  const bool fast = fmaxf(__uint_as_float(cmmax[(size_t)seg * (max_iter + 2) + t]),
                          __uint_as_float(frozen_cm[seg])) <= 64.f;
  const float code = fast ? 2.f : 1.f;
  const int base = seg_off[seg];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_run = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool active = i < n && cur[base + i].w != code;
    const unsigned long long b = __ballot(active);
    if (lane == 0) s_wave[wave] = __builtin_popcountll(b);
    __syncthreads();
    int before = s_run;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (active) act_idx[base + before + pvn3d_mbcnt(b)] = i;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_run += tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
    act_cnt[seg] = s_run + 1;
    act_form[seg] = fast ? 2 : 1;
  }
}

// number of fits that would still run iteration t+1.  grid 1, block 256.
__global__ void ms_poll_kernel(const unsigned* __restrict__ maxshift,
                               const int* __restrict__ seg_cnt, int n_seg, int t, int max_iter,
                               float thresh, const int* __restrict__ win_it, int stop_on_win,
                               int* __restrict__ active_slot) {
  __shared__ int s_any;
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  int mine = 0;
  for (int s = threadIdx.x; s < n_seg; s += blockDim.x) {
    if (seg_cnt[s] <= 0) continue;
    const float prev = __uint_as_float(maxshift[(size_t)s * (max_iter + 2) + t]);
    if (prev >= thresh && t <= max_iter && !(stop_on_win && win_it[s])) mine++;
  }
  if (mine) atomicAdd(&s_any, mine);
  __syncthreads();
  if (threadIdx.x == 0) *active_slot = s_any;
}

// ---------------------------------------------------------------------------------------
// Pruned neighbour count (exact).  The reference counts, for every point i, the points within bw
// (meanshift_pytorch.py:46-48): n^2 distance tests.  Votes are tightly clustered, so most pairs
// are decided by the triangle inequality around the segment mean m: with r_i = |a_i - m|,
//   r_i <= R and r_j <= R, R = 0.499 bw   =>   |a_i - a_j| <= 0.998 bw,
// which is a hit under the reference's fp32 test as well (its rounding error is ~1e-7 relative,
// the margin 2e-3).  "Core" points (r <= R) therefore all count each other without a test;
// only pairs with a non-core partner are evaluated, with the oracle's exact arithmetic:
// `dis < bw` (meanshift_pytorch.py:48) as d2 <= d2_max, d2 = (dx*dx + dy*dy) + dz*dz unfused, where
// d2_max is the largest fp32 whose correctly rounded sqrt is < bw (host-computed).  counts[i] = (core_i ? n_core : hits among core columns)
//                              + hits among non-core columns.
// Worst case (no core points) = the full n^2 scan; typical votes: ~10 % of it.
// ---------------------------------------------------------------------------------------
// grid (n_seg), block 1024
// It is also the first kernel of a fit batch and zeroes its segment's per-call state (a zero-fill pass of its own was
// one more launch per batch -- a tenth of a single-frame call).
__global__ __launch_bounds__(1024) void ms_classify_kernel(
    const float4* __restrict__ pts, const int* __restrict__ seg_off, const int* __restrict__ seg_cnt,
    float r_core, int* __restrict__ core_idx, int* __restrict__ nc_idx, int* __restrict__ n_core, MsState S,
    int ms_stride) {
  __shared__ double s_sum[3][16];
  __shared__ float s_mean[3];
  __shared__ int s_wc[16], s_run[2];
  const int seg = blockIdx.x, tid = threadIdx.x, wid = tid >> 6, lane = tid & 63;
  for (int i = tid; i < ms_stride; i += 1024) {
    S.maxshift[(size_t)seg * ms_stride + i] = 0u;
    S.cmmax[(size_t)seg * ms_stride + i] = 0u;
  }
  if (tid == 0) {
    S.iters[seg] = 0; S.best[seg] = 0ULL; S.frozen_cm[seg] = 0u; S.act_cnt[seg] = 0; S.act_form[seg] = 0;
    S.win_pos[seg] = make_float4(0.f, 0.f, 0.f, 0.f); S.win_it[seg] = 0;
    if (seg == 0) { S.active[0] = 0; S.active[1] = 0; }
  }
  const int n = seg_cnt[seg];
  if (n <= 0) { if (tid == 0) n_core[seg] = 0; return; }
  const int base = seg_off[seg];
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int i = tid; i < n; i += 1024) { const float4 a = pts[base + i]; sx += a.x; sy += a.y; sz += a.z; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64);
  }
  if (lane == 0) { s_sum[0][wid] = sx; s_sum[1][wid] = sy; s_sum[2][wid] = sz; }
  if (tid < 2) s_run[tid] = 0;
  __syncthreads();
  if (tid < 3) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += s_sum[tid][w];
    s_mean[tid] = (float)(t / (double)n);
  }
  __syncthreads();
  const float mx = s_mean[0], my = s_mean[1], mz = s_mean[2];
  const float r2 = r_core * r_core;
  // order-preserving compaction of both classes (deterministic lists)
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    bool core = false, valid = i < n;
    if (valid) {
      const float4 a = pts[base + i];
      const float dx = a.x - mx, dy = a.y - my, dz = a.z - mz;
      core = (dx * dx + dy * dy) + dz * dz <= r2;
    }
    const unsigned long long bc = __ballot(valid && core);
    const unsigned long long bn = __ballot(valid && !core);
    if (lane == 0) s_wc[wid] = __popcll(bc) | (__popcll(bn) << 16);
    __syncthreads();
    int pre_c = 0, pre_n = 0;
    for (int w = 0; w < wid; ++w) { pre_c += s_wc[w] & 0xffff; pre_n += s_wc[w] >> 16; }
    const int run_c = s_run[0], run_n = s_run[1];
    if (valid) {
      if (core) core_idx[base + run_c + pre_c + pvn3d_mbcnt(bc)] = i;
      else { nc_idx[base + run_n + pre_n + pvn3d_mbcnt(bn)] = i; S.counts_t[base + i] = 0; }
    }
    __syncthreads();
    if (tid == 0) {
      int tc = 0, tn = 0;
      for (int w = 0; w < 16; ++w) { tc += s_wc[w] & 0xffff; tn += s_wc[w] >> 16; }
      s_run[0] = run_c + tc; s_run[1] = run_n + tn;
    }
    __syncthreads();
  }
  if (tid == 0) n_core[seg] = s_run[0];
}

// ROWS_NC = false: rows = every point, columns = the non-core list; writes counts[i].
// ROWS_NC = true : rows = the non-core list, columns = the core list; adds to counts[i].
// SPLIT = false: a workgroup owns 256 rows, every lane walks all columns.  grid (n_seg, ceil(max_cnt/256)).
// SPLIT = true : a workgroup owns 64 rows and its four waves walk one quarter of every staged column chunk each,
//                then add their partial counts through LDS (integers: any order is exact).  grid (n_seg,
//                ceil(max_cnt/64)).  Used for the ROWS_NC launch: it has few rows (the non-core points) and many
//                columns, so its duration was one lane's walk over all core points -- 217 us per 576-fit batch,
//                33 us per single-frame batch, for 1 us of work per SIMD.
template <bool ROWS_NC, bool SPLIT>
__global__ __launch_bounds__(MS_THREADS) void ms_count_pruned_kernel(
    const float4* __restrict__ pts, const int* __restrict__ seg_off, const int* __restrict__ seg_cnt,
    const int* __restrict__ core_idx, const int* __restrict__ nc_idx, const int* __restrict__ n_core,
    float d2_max, int* __restrict__ counts) {
  constexpr int LANES = SPLIT ? 64 : MS_THREADS;
  __shared__ float4 s_pts[MS_CHUNK];
  __shared__ int s_cnt[SPLIT ? 3 : 1][SPLIT ? 64 : 1];
  // grid (n_seg, tiles): the segment is the FAST grid dimension.  Most tiles of the non-core
  // launch are empty (few non-core rows); with the tile as the fast dimension the surviving
  // workgroups (tile 0/1 of every segment, linear ids 12*seg + {0,1}) all land on the same four
  // of the eight XCDs and the launch took 5x longer than its work.
  // gridDim.y is NOT the tile count of the host's bound on a segment (N = 12 288 votes: 49 / 193 tiles, of which a fit of
  // 3 072 votes with ~300 non-core points uses 12 / 5): a workgroup walks tiles blockIdx.y, + gridDim.y, ... of its
  // segment, and the launcher sizes gridDim.y for ~8 k workgroups in all.  With one workgroup per bound tile the headline
  // batch launched 111 k workgroups for 3 k tiles of work in the non-core launch -- 155 us of workgroups that read two
  // words and exit (round 6).
  const int seg = blockIdx.x;
  const int n = seg_cnt[seg];
  const int ncore = n_core[seg], nnc = n - ncore;
  const int n_rows = ROWS_NC ? nnc : n;
  const int n_cols = ROWS_NC ? ncore : nnc;
  const int base = seg_off[seg];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int* cols = (ROWS_NC ? core_idx : nc_idx) + base;
  for (int tile0 = blockIdx.y * LANES; tile0 < n_rows; tile0 += gridDim.y * LANES) {
    const int r = tile0 + (SPLIT ? (tid & 63) : tid);
    int i = -1;
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n_rows) {
      i = ROWS_NC ? nc_idx[base + r] : r;
      c = pts[base + i];
    }
    int count = 0;
    for (int j0 = 0; j0 < n_cols; j0 += MS_CHUNK) {
      const int cnt = min(MS_CHUNK, n_cols - j0);
      __syncthreads();
      for (int q = tid; q < cnt; q += MS_THREADS) s_pts[q] = pts[base + cols[j0 + q]];
      __syncthreads();
      const int qb = SPLIT ? wave * MS_QUARTER : 0, qe = SPLIT ? min(qb + MS_QUARTER, cnt) : cnt;
      for (int q = qb; q < qe; ++q) {
        const float4 a = s_pts[q];
        const float dx = a.x - c.x, dy = a.y - c.y, dz = a.z - c.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        count += (d2 <= d2_max) ? 1 : 0;
      }
    }
    if (SPLIT) {
      __syncthreads();                                 // (the previous tile's partial counts have been read)
      if (wave > 0) s_cnt[wave - 1][tid & 63] = count;
      __syncthreads();
      if (wave == 0) count += s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid];
    }
    if (i >= 0 && (!SPLIT || wave == 0)) {
      if (ROWS_NC) counts[base + i] += count;   // one thread per row, after the first launch: no race
      else counts[base + i] = count;
    }
  }
}

// Round 6: one launch instead of the two above.  Rows = every point in LIST order (the core list, then the others),
// columns = the non-core list.  A (core row i, non-core column j) test is the same test as (row j, column i) of the
// second launch above -- d2 is symmetric bit for bit ((-x)^2 == x^2) -- so the hits of the core rows are also summed per
// COLUMN: the wave's ballot of a column's hits, masked to its core rows, popcounted, added to an LDS counter of the
// column (64 columns per LDS atomic), and flushed to counts_t[column's point] with one global atomic per column, chunk
// and workgroup.
// counts[i] = hits of row i among the non-core columns (as before); a non-core point's hits among the core points are
// counts_t[i] (integers: any order is exact).  The second launch walked n_nc x n_core pairs again: 150 of the 280 us of
// the two launches on the headline batch.  A workgroup owns TPW consecutive 256-row tiles of its segment (grid (n_seg,
// ceil(bound tiles / TPW))) so that a column's LDS counter collects TPW tiles before it costs an atomic.
template <int TPW>
__global__ __launch_bounds__(MS_THREADS) void ms_count_sym_kernel(
    const float4* __restrict__ pts, const int* __restrict__ seg_off, const int* __restrict__ seg_cnt,
    const int* __restrict__ core_idx, const int* __restrict__ nc_idx, const int* __restrict__ n_core,
    float d2_max, int* __restrict__ counts, int* __restrict__ counts_t) {
  typedef __attribute__((ext_vector_type(2))) float f2s;
  __shared__ float4 s_pts[TPW >= 2 ? 1 : MS_CHUNK];
  __shared__ f2s s_xx[TPW >= 2 ? MS_CHUNK : 1], s_yy[TPW >= 2 ? MS_CHUNK : 1], s_zz[TPW >= 2 ? MS_CHUNK : 1];
  __shared__ int s_col[MS_CHUNK];
  const int seg = blockIdx.x;
  const int n = seg_cnt[seg];
  const int row0 = blockIdx.y * (TPW * MS_THREADS);
  if (row0 >= n) return;
  const int ncore = n_core[seg], nnc = n - ncore;
  const int base = seg_off[seg];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_row = __builtin_amdgcn_readfirstlane(tid & ~63);
  const int* cols = nc_idx + base;
  int idx[TPW], count[TPW];
  float4 c[TPW];
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int r = row0 + t * MS_THREADS + tid;
    idx[t] = -1;
    count[t] = 0;
    c[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < n) {
      idx[t] = r < ncore ? core_idx[base + r] : nc_idx[base + r - ncore];
      c[t] = pts[base + idx[t]];
    }
  }
  const bool any_core = row0 < ncore;                  // (workgroup-uniform) some row of this workgroup is a core row
  for (int j0 = 0; j0 < nnc; j0 += MS_CHUNK) {
    const int cnt = min(MS_CHUNK, nnc - j0);
    __syncthreads();
    const int cnt64 = (cnt + 63) & ~63;
    for (int q = tid; q < cnt64; q += MS_THREADS) {
      const float inf = __builtin_inff();
      const float4 a = q < cnt ? pts[base + cols[j0 + q]] : make_float4(inf, inf, inf, 0.f);
      if constexpr (TPW >= 2) {
        s_xx[q] = f2s{a.x, a.x}; s_yy[q] = f2s{a.y, a.y}; s_zz[q] = f2s{a.z, a.z};
      } else {
        s_pts[q] = a;
      }
      s_col[q] = 0;
    }
    __syncthreads();
    if constexpr (TPW >= 2) {
      // two row tiles per pass, packed: lane holds rows r and r + 256 as (x, x'), (y, y'), (z, z') pairs, a column is read
      // once as (x, x), (y, y), (z, z) -- v_pk_add / v_pk_mul are IEEE per component, the sum order is the scalar one
      typedef __attribute__((ext_vector_type(2))) float f2;
#pragma unroll
      for (int t = 0; t < TPW; t += 2) {
        const int wr0 = row0 + t * MS_THREADS + wave_row, wr1 = wr0 + MS_THREADS;     // wave-uniform first rows
        if (wr0 >= n) continue;
        const unsigned long long cm0 = wr0 >= ncore ? 0ULL : wr0 + 64 <= ncore ? ~0ULL : ((1ULL << (ncore - wr0)) - 1ULL);
        const unsigned long long cm1 = wr1 >= ncore ? 0ULL : wr1 + 64 <= ncore ? ~0ULL : ((1ULL << (ncore - wr1)) - 1ULL);
        const f2 cx = {c[t].x, c[t + 1].x}, cy = {c[t].y, c[t + 1].y}, cz = {c[t].z, c[t + 1].z};
        float d2m = d2_max;
        asm("" : "+v"(d2m));                            // (one register for the loop, not a move per compare)
        // the walk over the columns, with and without the column sums (cm1 != 0 implies cm0 != 0); the LDS index lives in
        // a VGPR so that the 64 unrolled columns are immediate offsets from one address
#define MS_SYM_WALK(COLS_TOO)                                                                          \
  for (int qb = 0; qb < cnt; qb += 64) {          /* (columns cnt .. roundup64(cnt) hold +inf: never a hit) */ \
    int colv = 0, qv = qb;                                                                             \
    asm("" : "+v"(qv));                                                                                \
    _Pragma("unroll") for (int k = 0; k < 64; ++k) {                                                   \
      const f2 ax = s_xx[qv + k], ay = s_yy[qv + k], az = s_zz[qv + k];                                \
      const f2 dx = ax - cx, dy = ay - cy, dz = az - cz;                                               \
      const f2 d2 = (dx * dx + dy * dy) + dz * dz;                                                     \
      unsigned long long m0, m1;                                                                       \
      asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m0) : "v"(d2.x), "v"(d2m));                             \
      asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m1) : "v"(d2.y), "v"(d2m));                             \
      asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(count[t]) : "s"(m0) : "vcc");                  \
      asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(count[t + 1]) : "s"(m1) : "vcc");              \
      if (COLS_TOO) {                                                                                  \
        const int pc = __popcll(m0 & cm0) + __popcll(m1 & cm1);                                        \
        asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(pc), "i"(k));                              \
      }                                                                                                \
    }                                                                                                  \
    if (COLS_TOO && qb + lane < cnt) atomicAdd(&s_col[qb + lane], colv);                               \
  }
        if (cm0 != 0ULL) { MS_SYM_WALK(true) } else { MS_SYM_WALK(false) }
#undef MS_SYM_WALK
      }
    } else {
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
      const int wr0 = row0 + t * MS_THREADS + wave_row;            // first row of this wave in tile t (wave-uniform)
      if (wr0 >= n) continue;
      const bool live = idx[t] >= 0;
      if (wr0 < ncore) {
        // (rows of the wave below n_core are core rows: a mask of the lanes)
        // (rows past n hold the origin and are never stored; rows past n_core are masked out of the column sums -- n_core
        // <= n, so the mask covers both)
        const unsigned long long core_mask = wr0 + 64 <= ncore ? ~0ULL : ((1ULL << (ncore - wr0)) - 1ULL);
        const float d2m = d2_max;
        // a column's count among this wave's core rows = popcount of the compare's lane mask (scalar unit); it is parked
        // in lane (q & 63) of a register and added to the LDS counters once per 64 columns (a branch and an LDS atomic
        // per column and wave cost as much as the pair tests they saved)
        for (int qb = 0; qb < cnt; qb += 64) {          // (columns cnt .. roundup64(cnt) hold +inf: never a hit)
          int colv = 0;
#pragma unroll
          for (int k = 0; k < 64; ++k) {
            const float4 a = s_pts[qb + k];
            const float dx = a.x - c[t].x, dy = a.y - c[t].y, dz = a.z - c[t].z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            // the compare's lane mask lands in an SGPR pair (a ballot of a bool costs a v_cndmask and a second compare):
            // + 1 per lane through the carry-in of an add, popcount of the core rows' bits on the scalar unit
            unsigned long long m;
            asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(m) : "v"(d2), "v"(d2m));
            asm("v_addc_co_u32_e64 %0, vcc, 0, %0, %1" : "+v"(count[t]) : "s"(m) : "vcc");
            const int pc = __popcll(m & core_mask);
            asm("v_writelane_b32 %0, %1, %2" : "+v"(colv) : "s"(pc), "i"(k));
          }
          if (qb + lane < cnt) atomicAdd(&s_col[qb + lane], colv);
        }
      } else {
        for (int q = 0; q < cnt; ++q) {
          const float4 a = s_pts[q];
          const float dx = a.x - c[t].x, dy = a.y - c[t].y, dz = a.z - c[t].z;
          const float d2 = dx * dx + dy * dy + dz * dz;
          count[t] += (live && d2 <= d2_max) ? 1 : 0;
        }
      }
    }
    }
    if (any_core) {
      __syncthreads();
      for (int q = tid; q < cnt; q += MS_THREADS) {
        const int v = s_col[q];
        if (v) atomicAdd(&counts_t[base + cols[j0 + q]], v);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; ++t)
    if (idx[t] >= 0) counts[base + idx[t]] = count[t];
}

// arg-max of the neighbour counts with the reference's first-maximum rule.  grid (ceil(max_cnt/256), n_seg), block 256.
// The rows are walked list by list (core list, then the others): a core row's count is its stored count (hits among
// the non-core columns) + n_core -- every core point is within bw of every core point -- added here instead of in a
// pass of its own; the key carries ~index, so the walk order does not matter.
__global__ __launch_bounds__(MS_THREADS) void ms_argmax_kernel(
    const int* __restrict__ seg_off, const int* __restrict__ seg_cnt, const int* __restrict__ counts,
    const int* __restrict__ counts_t, const int* __restrict__ core_idx, const int* __restrict__ nc_idx,
    const int* __restrict__ n_core, unsigned long long* __restrict__ best) {
  __shared__ unsigned long long s_red[MS_THREADS / 64];
  const int seg = blockIdx.y;
  const int n = seg_cnt[seg];
  const int tile0 = blockIdx.x * MS_THREADS;
  if (tile0 >= n) return;
  const int base = seg_off[seg];
  const int ncore = n_core[seg];
  const int tid = threadIdx.x;
  const int r = tile0 + tid;
  unsigned long long key = 0ULL;
  if (r < n) {
    const int i = r < ncore ? core_idx[base + r] : nc_idx[base + r - ncore];
    const int c = counts[base + i] + (r < ncore ? ncore : counts_t[base + i]);
    key = ((unsigned long long)(unsigned)c << 32) | (unsigned long long)(~(unsigned)i);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)key, o, 64);
    const unsigned hi = __shfl_xor((unsigned)(key >> 32), o, 64);
    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
    key = other > key ? other : key;
  }
  if ((tid & 63) == 0) s_red[tid >> 6] = key;
  __syncthreads();
  if (tid == 0) {
    unsigned long long m = s_red[0];
    for (int w = 1; w < MS_THREADS / 64; ++w) m = s_red[w] > m ? s_red[w] : m;
    atomicMax(best + seg, m);
  }
}

// labels = |A_j - A_maxidx| < bw ; ctr = C[maxidx] back in the camera frame.
// grid: (tiles, n_seg), block 256.  Tile 0 also writes ctr / iters.
__global__ __launch_bounds__(MS_THREADS) void ms_final_kernel(
    const float4* __restrict__ pts, const int* __restrict__ seg_off,
    const int* __restrict__ seg_cnt, const float4* __restrict__ c0,
    const float4* __restrict__ c1, const unsigned long long* __restrict__ best,
    const int* __restrict__ iters_ws, const float4* __restrict__ win_pos, const int* __restrict__ win_it,
    const unsigned* __restrict__ maxshift, int max_iter, float thresh, int limit,
    float d2_max, float inv_kappa, float* __restrict__ ctr, uint8_t* __restrict__ labels,
    int* __restrict__ iters) {
  const int seg = blockIdx.y;
  const int n = seg_cnt[seg];
  const int tid = threadIdx.x;
  if (n <= 0) {
    if (blockIdx.x == 0 && tid == 0) {
      ctr[seg * 3 + 0] = ctr[seg * 3 + 1] = ctr[seg * 3 + 2] = 0.f;
      if (iters) iters[seg] = 0;
    }
    return;
  }
  const int j = blockIdx.x * MS_THREADS + tid;
  if (blockIdx.x * MS_THREADS >= n) return;
  const int base = seg_off[seg];
  const int max_idx = (int)(~(unsigned)(best[seg] & 0xffffffffULL));
  const float4 c = pts[base + max_idx];
  if (j < n && labels) {
    const float4 a = pts[base + j];
    const float dx = a.x - c.x, dy = a.y - c.y, dz = a.z - c.z;
    const float d2 = dx * dx + dy * dy + dz * dz;
    labels[base + j] = (d2 <= d2_max) ? 1 : 0;
  }
  if (blockIdx.x == 0 && tid == 0) {
    const int it = iters_ws[seg];
    const float4 org = pts[base];   // frame origin used by the iterations
    // the winner's recorded fixed point if there is one, else its position after the last iteration run
    const float4 m = win_it[seg] ? win_pos[seg] : ((it & 1) ? c1[base + max_idx] : c0[base + max_idx]);
    ctr[seg * 3 + 0] = m.x * inv_kappa + org.x;
    ctr[seg * 3 + 1] = m.y * inv_kappa + org.y;
    ctr[seg * 3 + 2] = m.z * inv_kappa + org.z;
    // enqueue limit (no host poll): a fit that would still run reports -(iterations run); its centre is not final
    const bool unfinished = it >= limit && it <= max_iter && !win_it[seg] &&
                            __uint_as_float(maxshift[(size_t)seg * (max_iter + 2) + it]) >= thresh;
    if (iters) iters[seg] = unfinished ? -it : it;
  }
}

// largest fp32 t with sqrtf(t) < bw (sqrtf correctly rounded on the host)
float d2_threshold(float bw) {
  float t = bw * bw;
  while (sqrtf(t) < bw) t = nextafterf(t, INFINITY);
  while (!(sqrtf(t) < bw)) t = nextafterf(t, -INFINITY);
  return t;
}

}  // namespace

extern "C" size_t pvn3d_meanshift_workspace_bytes(int n_seg, int total, int max_iter) {
  if (n_seg < 0 || total < 0 || max_iter < 0) return 0;
  return ms_layout(n_seg, total, max_iter, nullptr, nullptr);
}

static int vote_compact_any(int n_pts, int seg_stride, int n_kps, int n_inst, int v_first, int v_count, const float* pcld,
                            const int* mask, const float* ctr_of, const float* pred_kp_of, const int* inst_frame,
                            const int* inst_cls, const uint8_t* sel, long long sel_inst_stride, float* votes,
                            int* seg_off, int* seg_cnt, void* stream) {
  if (n_inst <= 0 || v_count <= 0) return 0;
  if (n_pts <= 0 || seg_stride < n_pts || n_kps < 0 || v_first < 0 || v_first + v_count > n_kps + 1 || !pcld ||
      !mask || !ctr_of || !votes || !seg_off || !seg_cnt || (v_first < n_kps && !pred_kp_of) ||
      (long long)seg_stride * n_inst * (n_kps + 1) > 0x7fffffffLL)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(vote_compact_kernel, dim3(n_inst, v_count), dim3(1024), 0,
                     (hipStream_t)stream, n_pts, seg_stride, n_kps, v_first, pcld, mask, ctr_of, pred_kp_of,
                     inst_frame, inst_cls, sel, sel_inst_stride, (float4*)votes, seg_off,
                     seg_cnt);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
extern "C" int pvn3d_vote_compact(int n_frames, int n_pts, int n_kps, int n_inst, int v_first,
                                  int v_count, const float* pcld, const int* mask,
                                  const float* ctr_of, const float* pred_kp_of,
                                  const int* inst_frame, const int* inst_cls,
                                  const uint8_t* sel, long long sel_inst_stride, float* votes,
                                  int* seg_off, int* seg_cnt, void* stream) {
  (void)n_frames;
  return vote_compact_any(n_pts, n_pts, n_kps, n_inst, v_first, v_count, pcld, mask, ctr_of, pred_kp_of, inst_frame, inst_cls,
                          sel, sel_inst_stride, votes, seg_off, seg_cnt, stream);
}
// Rows per segment as an argument.  With one segment of n_pts = 12288 rows per (instance, keypoint) every segment starts
// 3 * 2^16 bytes after the previous one: the iteration kernels' waves -- one fit each, all walking their fit's points at
// the same pace -- then ask the SAME memory channels for their next block at the same time (measured, tools/ms_rate.py:
// 576 fits of 3072 votes, 1.26 ms per iteration at 12288 rows per segment, 1.00 ms at 12288 + 32, 0.95 ms packed tight).
extern "C" int pvn3d_vote_compact_strided(int n_frames, int n_pts, int seg_stride_rows, int n_kps, int n_inst, int v_first,
                                          int v_count, const float* pcld, const int* mask, const float* ctr_of,
                                          const float* pred_kp_of, const int* inst_frame, const int* inst_cls,
                                          const uint8_t* sel, long long sel_inst_stride, float* votes, int* seg_off,
                                          int* seg_cnt, void* stream) {
  (void)n_frames;
  return vote_compact_any(n_pts, seg_stride_rows, n_kps, n_inst, v_first, v_count, pcld, mask, ctr_of, pred_kp_of,
                          inst_frame, inst_cls, sel, sel_inst_stride, votes, seg_off, seg_cnt, stream);
}

extern "C" int pvn3d_meanshift_fit_batch(const float* pts, const int* seg_off,
                                         const int* seg_cnt, int n_seg, int total,
                                         int max_cnt_host, float bandwidth, int max_iter,
                                         float* ctr, uint8_t* labels, int* iters,
                                         void* workspace, size_t workspace_bytes,
                                         int* poll_host, int poll_every, int flags,
                                         void* stream) {
  if (n_seg <= 0) return 0;
  if (!pts || !seg_off || !seg_cnt || !ctr || !workspace || total <= 0 || max_cnt_host <= 0 ||
      max_iter < 0 || !(bandwidth > 0.f))
    return (int)hipErrorInvalidValue;
  if (workspace_bytes < pvn3d_meanshift_workspace_bytes(n_seg, total, max_iter))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  MsState S;
  ms_layout(n_seg, total, max_iter, (char*)workspace, &S);
  // The per-call state (stop-rule tables, winner records, seed lists' headers) is zeroed by the batch's first kernel,
  // ms_classify_kernel -- not by hipMemsetAsync: inside a captured HIP graph (GraphedFramePoses) the runtime's memset
  // node was observed to land out of order with the kernels around it (random memory faults from a zeroed-too-late
  // n_core / counts); a kernel is an ordinary link of the chain.  counts / core_idx / nc_idx / act_idx need no zeroing
  // (every entry that is read has been written by this call).

  const float thresh = (float)((double)bandwidth * 1e-3);  // meanshift_pytorch.py:21
  const float kappa = sqrtf(0.5f * 1.44269504088896341f) / bandwidth;
  const float inv_kappa = 1.0f / kappa;
  const float d2_max = d2_threshold(bandwidth);
  const float4* P = (const float4*)pts;

  // Two seeds per lane (packed fp32 math) once the largest fit fills at least two 128-seed tiles.  The
  // split-point workgroup shape (64 seed lanes, each wave a quarter of the points) is the default: measured
  // on MI355X it is 6 % faster than whole-fit waves on the full 576-fit batch (5.65 vs 6.01 ms) and 1.6x
  // faster when stragglers of heavy-tailed vote sets run alone (3.15 vs 5.19 ms per frame), because an
  // iteration launch can never finish before its slowest workgroup.  All four kernels give identical bits
  // (tests/test_gpu_postproc.py); the FORCE flags exist for that test and for A/B timing.
  bool packed = max_cnt_host >= 256;
  if (flags & PVN3D_MS_FORCE_SCALAR) packed = false;
  if (flags & PVN3D_MS_FORCE_PACKED) packed = true;
  bool split = true;
  if (flags & PVN3D_MS_FORCE_WHOLE) split = false;
  if (flags & PVN3D_MS_FORCE_SPLIT) split = true;
  const int tile = (split ? 64 : MS_THREADS) * (packed ? 2 : 1);
  const dim3 grid_it(n_seg, pvn3d_ceil_div(max_cnt_host, tile));
  // LDS-free iteration kernel (one wave per 64 S seeds, points as SGPR operands)
  const bool sgpr = (flags & PVN3D_MS_SGPR_POINTS) != 0;
  if (sgpr && !(flags & PVN3D_MS_ALIGNED32)) return (int)hipErrorInvalidValue;   // it reads rows up to roundup32(cnt)
  const int sg_tiles = pvn3d_ceil_div(max_cnt_host, 128);   // always two seeds per lane
  const long long sg_items_ll = (long long)sg_tiles * n_seg;
  if (sgpr && sg_items_ll > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  const int sg_items = (int)sg_items_ll;
  const int sg_cap = (flags >> 8) & 0xfffff;                                      // PVN3D_MS_WAVE_CAP(n); 0: no cap
  const int sg_grid = sg_cap > 0 && sg_cap < sg_items ? sg_cap : sg_items;
  if (sgpr)
    hipLaunchKernelGGL(ms_prep_kernel, dim3(pvn3d_ceil_div(max_cnt_host + 3, MS_THREADS), n_seg), dim3(MS_THREADS), 0, st,
                       P, seg_off, seg_cnt, kappa, S.apts);
  const dim3 grid_1(pvn3d_ceil_div(max_cnt_host, MS_THREADS), n_seg);

  // max_idx = arg-max of the neighbour counts of the ORIGINAL points (meanshift_pytorch.py:46-49): independent of the
  // iterations, so it is computed first -- the iteration kernels watch that seed (winner stop, above)
  {
    const float r_core = 0.499f * bandwidth;
    hipLaunchKernelGGL(ms_classify_kernel, dim3(n_seg), dim3(1024), 0, st, P, seg_off, seg_cnt, r_core,
                       S.core_idx, S.nc_idx, S.n_core, S, max_iter + 2);
    // one symmetric count launch (ms_count_sym_kernel): four row tiles per workgroup for batches with fits enough to fill
    // the chip that way, one otherwise (a single frame's nine fits want every tile on a CU of its own)
    const int tiles_b = pvn3d_ceil_div(max_cnt_host, MS_THREADS);
    if (flags & PVN3D_MS_COUNT_TWO_PASS) {
      // rounds 2-5: the same (core, non-core) tests twice, once per side (cross-check of the symmetric launch); the
      // argmax kernel reads counts_t for the non-core rows: this path leaves it zero (ms_classify_kernel)
      const int wg_y = max(1, pvn3d_ceil_div(8192, n_seg));
      const dim3 grid_t(n_seg, min(tiles_b, wg_y));
      const dim3 grid_ts(n_seg, min(pvn3d_ceil_div(max_cnt_host, 64), wg_y));
      hipLaunchKernelGGL((ms_count_pruned_kernel<false, false>), grid_t, dim3(MS_THREADS), 0, st, P, seg_off, seg_cnt,
                         S.core_idx, S.nc_idx, S.n_core, d2_max, S.counts);
      hipLaunchKernelGGL((ms_count_pruned_kernel<true, true>), grid_ts, dim3(MS_THREADS), 0, st, P, seg_off, seg_cnt,
                         S.core_idx, S.nc_idx, S.n_core, d2_max, S.counts);
    } else if (n_seg >= 128)
      hipLaunchKernelGGL((ms_count_sym_kernel<4>), dim3(n_seg, pvn3d_ceil_div(tiles_b, 4)), dim3(MS_THREADS), 0, st, P, seg_off,
                         seg_cnt, S.core_idx, S.nc_idx, S.n_core, d2_max, S.counts, S.counts_t);
    else
      hipLaunchKernelGGL((ms_count_sym_kernel<1>), dim3(n_seg, tiles_b), dim3(MS_THREADS), 0, st, P, seg_off, seg_cnt,
                         S.core_idx, S.nc_idx, S.n_core, d2_max, S.counts, S.counts_t);
    hipLaunchKernelGGL(ms_argmax_kernel, grid_1, dim3(MS_THREADS), 0, st, seg_off, seg_cnt, S.counts, S.counts_t,
                       S.core_idx, S.nc_idx, S.n_core, S.best);
  }
  PVN3D_LAUNCH_CHECK();
  const int stop_on_win = (flags & PVN3D_MS_NO_WINNER_STOP) ? 0 : 1;

  hipEvent_t ev[2] = {nullptr, nullptr};
  int pending[2] = {0, 0};
  int slot = 0;
  const bool poll = poll_host != nullptr && poll_every > 0;
  if (poll) {
    PVN3D_RETURN_IF_ERR(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    PVN3D_RETURN_IF_ERR(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
  }
  int rc = 0;
  int next_poll = poll_every;
  // no poll buffer + poll_every = E > 0: enqueue at most E iterations and report unfinished fits through `iters`
  // (negative) -- a launch sequence of fixed length without a host round trip, i.e. capturable in a HIP graph
  const int iter_limit = (!poll && poll_every > 0 && poll_every < max_iter + 1) ? poll_every : max_iter + 1;
  for (int t = 1; t <= iter_limit; ++t) {
    const float4* cin = S.cbuf[(t - 1) & 1];
    float4* cout = S.cbuf[t & 1];
#define MS_ITER(PK_, SP_)                                                                                    \
  hipLaunchKernelGGL((ms_iter_kernel<PK_, SP_>), grid_it, dim3(MS_THREADS), 0, st, P, seg_off, seg_cnt, cin, cout, \
                     S.maxshift, S.cmmax, S.iters, t, max_iter, thresh, kappa, inv_kappa, S.frozen_cm, S.act_cnt,   \
                     S.act_form, S.act_idx, S.best, S.win_pos, S.win_it, stop_on_win)
#define MS_ITER_SGPR()                                                                                         \
  hipLaunchKernelGGL(ms_iter_sgpr_kernel, dim3(sg_grid), dim3(64), 0, st, P, S.apts, seg_off, seg_cnt, cin, cout, \
                     S.maxshift, S.cmmax, S.iters, t, max_iter, thresh, kappa, inv_kappa, S.frozen_cm, S.act_cnt,   \
                     S.act_form, S.act_idx, sg_tiles, sg_items, S.best, S.win_pos, S.win_it, stop_on_win)
    if (sgpr) MS_ITER_SGPR();
    else if (packed) { if (split) MS_ITER(true, true); else MS_ITER(true, false); }
    else { if (split) MS_ITER(false, true); else MS_ITER(false, false); }
#undef MS_ITER
#undef MS_ITER_SGPR
    if ((rc = (int)hipGetLastError()) != 0) break;
    // from iteration 5 on (the easy fits are done after ~4) the seed lists are rebuilt every fourth iteration:
    // on heavy-tailed votes 85-89 % of the seeds are bitwise fixed points after five iterations
    // (tools/ms_frozen_stats.py).  What it buys (144 fits of 3072 votes, 10 % outliers of sigma 30 cm, 267
    // iterations): 43.4 -> 26.8 ms.  The iteration launch does not get 8x shorter with 8x fewer seeds: its
    // duration is ~1.2 us per still-running fit + 7 us whatever the number of busy workgroups (measured; a
    // 16-wave workgroup shape and batched prologue loads changed nothing), so the long tail of a heavy-tailed
    // batch stays (iterations) x (that floor).
    // (a fixed sequence of at most 8 iterations -- the single-frame graph -- is over before a list pays for itself)
    if (!(flags & PVN3D_MS_NO_EARLY_OUT) && t >= 5 && (t & 3) == 1 && t <= max_iter && iter_limit > 8) {
      hipLaunchKernelGGL(ms_compact_kernel, dim3(n_seg), dim3(1024), 0, st, seg_off, seg_cnt, cout, S.maxshift,
                         S.cmmax, S.frozen_cm, t, max_iter, thresh, S.act_cnt, S.act_form, S.act_idx, S.win_it,
                         stop_on_win);
      if ((rc = (int)hipGetLastError()) != 0) break;
    }
    // Poll schedule: every poll_every iterations at first, then geometrically sparser (x1.5).  Late iterations
    // are short (few fits, few seeds left) -- four of them take less than one host round trip, so a fixed
    // cadence leaves the GPU waiting for the host -- and an iteration enqueued for nothing costs one launch.
    if (poll && t == next_poll && t <= max_iter) {
      next_poll = t + (t / 2 > poll_every ? t / 2 : poll_every);
      hipLaunchKernelGGL(ms_poll_kernel, dim3(1), dim3(256), 0, st, S.maxshift, seg_cnt, n_seg,
                         t, max_iter, thresh, S.win_it, stop_on_win, S.active + slot);
      if ((rc = (int)hipMemcpyAsync(poll_host + slot, S.active + slot, sizeof(int),
                                    hipMemcpyDeviceToHost, st)) != 0) break;
      if ((rc = (int)hipEventRecord(ev[slot], st)) != 0) break;
      pending[slot] = 1;
      const int other = slot ^ 1;  // the GPU stays one chunk ahead of the host test
      if (pending[other]) {
        if ((rc = (int)hipEventSynchronize(ev[other])) != 0) break;
        pending[other] = 0;
        if (poll_host[other] == 0) break;
      }
      slot = other;
    }
  }
  if (poll) {
    // a device-to-host copy into poll_host may still be in flight (early break): let it land before
    // the caller's buffer can be reused by the next call
    for (int k = 0; k < 2; ++k)
      if (pending[k]) (void)hipEventSynchronize(ev[k]);
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
  }
  if (rc) return rc;
  PVN3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(ms_final_kernel, grid_1, dim3(MS_THREADS), 0, st, P, seg_off, seg_cnt,
                     S.cbuf[0], S.cbuf[1], S.best, S.iters, S.win_pos, S.win_it, S.maxshift, max_iter, thresh,
                     iter_limit, d2_max, inv_kappa, ctr, labels, iters);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

#ifdef MS_FROZEN_PROBE
extern "C" int pvn3d_ms_probe_read(int* host_out, int reset) {
  hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ms_probe), sizeof(int) * 1024);
  if (e != hipSuccess) return (int)e;
  if (reset) {
    static int zeros[1024];
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_ms_probe), zeros, sizeof(int) * 1024);
  }
  return (int)e;
}
#endif
