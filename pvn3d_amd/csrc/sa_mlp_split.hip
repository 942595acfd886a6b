// sa_mlp_split.hip -- the fused "group -> SharedMLP -> max-pool" (set abstraction) and "three_interpolate -> concat ->
// SharedMLP" (feature propagation) chains of csrc/sa_mlp.hip with the fp32 contraction carried by the bf16 matrix pipe.
//
// Same data flow and the same reference semantics as sa_mlp.hip (pvn3d/lib/pointnet2_utils/pointnet2_modules.py:57-69,
// 183-206; pytorch_utils.py:25-50; eval BatchNorm folded on the host).  What changes is the arithmetic of the 1x1
// convolutions.  v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 number is the exact sum of three
// bf16 numbers x = x1 + x2 + x3 (8 + 8 + 8 significant bits), a bf16 x bf16 product is exact in fp32, so
//     w.x = sum_{i,j} w_i.x_j   accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// The six terms with i + j <= 4 are kept; the three dropped ones are below 2^-24 of the product -- fp32's own rounding
// step -- so the result carries fp32 accuracy (tools/mfma_split_bench.hip: max error 4.3e-7 of the output scale at
// K = 512 against 2.7e-7 for the fp32 FMA chain, identical rms) at 6/16 of the fp32-MFMA cost.  It is NOT a reduced
// precision path: no operand is rounded to bf16, every bit of both fp32 operands enters the product.
//   * weights: split on the host (round to nearest per piece), packed per layer as
//     [K/16 slabs][M/32 row tiles][3 pieces][64 lanes] x 16 bytes = the A fragment of one MFMA per load
//     (lane l: row mt*32 + (l & 31), k = 16*slab + 8*(l >> 5) + 0..7);
//   * activations: split by truncation where they are produced (x1 = x & 0xffff0000, x2 = (x - x1) & 0xffff0000,
//     x3 = x - x1 - x2: exact, all three pieces share the sign) and kept in LDS as three bf16 planes P[piece][column][k]
//     (row stride 2*Kcap + 16 bytes = 16 x odd: conflict-free 16-byte B-fragment reads).
//
// Workgroup = 4 MFMA waves + 4 loader waves (one of each per SIMD, 256 registers per wave), 64 columns ((centre,
// sample) pairs or unknown points), persistent over column blocks.  A kernel is instantiated per chain signature
// (row tiles per MFMA wave in layer 0 / 1 / 2): the layer sequence is straight-line code, every layer has exactly one
// loop body, and nothing of one layer's address arithmetic or fragments stays alive across another's.  The layer-0 input is gathered, split and staged by the LOADER waves into a ring of four 32-channel
// chunk buffers (loader wave j owns slot j and every fourth chunk); the MFMA waves only read LDS and stream weight
// fragments, so their vmcnt queue never holds a gather (a weight load queued behind a gather completes after it, and
// with bf16 MFMAs a chunk is multiplied in a third of the time a gather takes).  Loaders and MFMA waves meet through
// LDS sequence words (ready / consumed per slot) and the MFMA waves synchronise among themselves through an LDS
// counter -- s_barrier would stop the loaders, which run up to four chunks (into the next column block) ahead.
#include <cstdlib>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) __fp16 fp16x2_t;

// Arithmetic of a chain kernel (template parameter AR):
//   0  bf16 x 3: three bf16 pieces per fp32 operand, six partial products per 16-k slab (header comment above);
//   1  fp16 x 2 (round 5): TWO fp16 pieces per operand, x s = h + l with h = fp16(x s) and l = fp16(x s - h) (both
//      rounded to nearest; the weights' pieces come from the host), THREE partial products wh.xh + wh.xl + wl.xh on
//      v_mfma_f32_32x32x16_f16 -- half the matrix-pipe time and two thirds of the LDS bytes of (0).  An fp16 x fp16
//      product is exact in fp32 (11 + 11 significant bits), the two pieces carry >= 20 bits of the operand and the
//      dropped wl.xl is below 2^-20 of the product; measured (tools/mfma_fp16x2_bench.hip, 256 tiles, K = 512 / 1536,
//      against fp64): max error 5.5e-7 / 9.2e-7 of the output scale vs 6.2e-7 / 1.0e-6 for the fp32 FMA chain and
//      8.6e-7 / 1.7e-6 for (0) -- the error of all three is the fp32 ACCUMULATION's, not the operands'.
//      fp16 has 5 exponent bits, so every operand is scaled by a power of two (exact) into its range: weights by a
//      per-layer sw (host), the layer-0 input by s_0 = 2^14 / 2^ceil(log2 B_0) with B_0 a bound on |input| read from
//      device memory (abs-max of the feature tables, written by pvn3d_absmax), hidden layers by the rigorous bound
//      B_{l+1} = ||W_l||_inf B_l + max|b_l| (host constants).  A loose bound costs nothing that matters: values far
//      below the bound lose low-piece bits against an ABSOLUTE floor of 2^-38 of the bound.  Scales are undone exactly
//      (powers of two) where a layer's accumulators leave: results differ from (0) only in rounding.

constexpr int S3_COLS = 64;
constexpr int S3_KC = 32;                       // input channels per layer-0 chunk (two 16-k slabs)
constexpr int S3_CS = 80;                       // bytes per column of a chunk buffer (64 + 16: 16 x odd)
constexpr int S3_CPS = S3_COLS * S3_CS;         // bytes per piece plane of a chunk buffer
constexpr int s3_np(int ar) { return ar == 1 ? 2 : 3; }                 // pieces per operand
constexpr int s3_chunk(int ar) { return s3_np(ar) * S3_CPS; }           // bytes of one chunk buffer (15360 / 10240)
constexpr int S3_RING = 4;                      // chunk buffers = loader waves
constexpr int S3_NWC = 4;                       // MFMA (consumer) waves
constexpr int S3_NWL = 4;                       // loader waves
constexpr int S3_THREADS = 64 * (S3_NWC + S3_NWL);
constexpr int S3_MAX_LAYERS = 4;
constexpr int S3_EPAD = 68;                     // row stride (floats) of a wave's max-pool patch

struct S3Args {
  int n_layers;
  int K[S3_MAX_LAYERS], M[S3_MAX_LAYERS];
  const uint4* W[S3_MAX_LAYERS];                // packed split weights
  const float* bias[S3_MAX_LAYERS];             // [ceil(M/32)*32] zero padded
  int is_sa;
  // set abstraction
  const float* xyz;                             // (B, n, 3)
  const float* new_xyz;                         // (B, m, 3)
  int n, m, ns, tail_xyz;
  // feature propagation
  const float* weight;                          // (B, n_cols, 3)
  const int* idx;                               // SA (B, m, ns); FP (B, n_cols, 3)
  // layer-0 row sources (point-major tables)
  const float* tabA; int rowsA, ldA, nA;        // SA: features, FP: known points; nA = 32-channel chunks
  const float* tabB; int rowsB, ldB, nB;        // FP: the unknown points' own features (full chunks)
  int tail_w;                                   // channels of the <= 8 wide tail chunk (SA: 3 xyz; FP: widthB - 32 nB)
  int cols_total, bpf, n_blocks;                // columns per frame, column blocks per frame, blocks in all
  int n_frames;
  int rs, ps;                                   // P: bytes per column, bytes per piece plane
  int ring_off;                                 // byte offset of the chunk ring in dynamic LDS (0 = overlaid on P)
  int ident_a;                                  // PVN3D_MLP_IDENTITY_A: layer 0's table-A slabs are an identity block (pre-contracted chains): their low weight pieces are zero, the wl.xh product is not issued
  int bias_off, bias_all;
  float* out; int point_major, ld_out, coff;
  // fp16 x 2 only: per-layer weight scale (power of two), ||W_l||_inf and max|b_l| of the true (folded) weights, and the
  // device-side bounds of the layer-0 input: B_0 = max(*bound_a, mul_b * *bound_b, 1e-30)
  float sw[S3_MAX_LAYERS], wnorm[S3_MAX_LAYERS], bmax[S3_MAX_LAYERS];
  const float* bound_a; const float* bound_b; float mul_b;
  unsigned* out_absmax;                         // device or nullptr: atomic max of |output| (bit pattern) for the consumer
  const float* out_row_mul;                     // device [ceil(M_last / 32) * 32] or nullptr: per-output-channel multiplier of the results
  int no_narrow;                                // per-call: keep the chain off the narrow-chain kernels (A/B measurements)
  int dbg;                                      // tuning builds only (-DPVN3D_S3_TUNING, env PVN3D_S3_DBG): 1 no gathers, 2 no index loads, 64 cycle stamps of workgroup 0
};

// The tuning probes change results (synthetic gather values / indices) and write a global buffer: they exist only in a
// build with -DPVN3D_S3_TUNING (tools/build_probe_lib.sh); the shipped library reads no environment variable and
// S3_DBG() is a compile-time false.
#ifdef PVN3D_S3_TUNING
#define S3_DBG(a, bit) (((a).dbg & (bit)) != 0)
#else
#define S3_DBG(a, bit) false
#endif

// tuning probe (PVN3D_S3_DBG & 64): cycle stamps of workgroup 0 -- [0..63] MFMA wave 0 (8 stamps per column block),
// [64..127] loader wave 0 (one stamp per chunk it staged, before / after); [128..255] the same stamps on the 100 MHz
// real-time counter
#ifdef PVN3D_S3_TUNING
__device__ unsigned long long g_s3_prof[256];
#define S3_STAMP(IDX)                                                                                   \
  do {                                                                                                  \
    if ((a.dbg & 64) && blockIdx.x == 0 && (IDX) < 128 && (threadIdx.x & 63) == 0) {                     \
      g_s3_prof[(IDX)] = __builtin_readcyclecounter();                                                  \
      g_s3_prof[128 + (IDX)] = __builtin_amdgcn_s_memrealtime();   /* 100 MHz: the pair gives the shader clock */ \
    }                                                                                                   \
  } while (0)
#else
#define S3_STAMP(IDX) do { } while (0)
#endif

// LDS control words (static): sequence numbers, all monotone
struct S3Ctl {
  unsigned rdy[S3_RING];      // chunk number + 1 that the slot holds (written by its loader wave)
  unsigned fin[S3_RING];      // MFMA waves that have finished reading the slot, summed over its uses
  unsigned bar;               // arrivals at the MFMA waves' barrier
  unsigned blk;               // column blocks whose LDS the MFMA waves have released (ring overlaid on P only)
};

__device__ __forceinline__ unsigned lds_peek(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_wait_ge(const unsigned* p, unsigned v) {
  // watchdog: a protocol error must end as a kernel fault, never as a hung GPU (2^24 polls of >= 64 cycles ~ 1 s; the
  // longest legitimate wait is one layer of one column block, tens of microseconds)
  unsigned spins = 0;
  while ((int)(lds_peek(p) - v) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1u << 24)) __builtin_trap();
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void lds_signal_add(unsigned* p, int lane, unsigned n = 1u) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // this wave's LDS traffic has completed
  if (lane == 0) __hip_atomic_fetch_add(p, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// barrier of the MFMA waves only
__device__ __forceinline__ void cbar(S3Ctl* ctl, unsigned& phase, int lane) {
  lds_signal_add(&ctl->bar, lane);
  phase += S3_NWC;
  lds_wait_ge(&ctl->bar, phase);
}

// block q of the launch -> (frame, column block): frame f on XCD f mod 8 when the frame count allows (persistent
// workgroup w visits q = w, w + gridDim, ...; gridDim is a multiple of 8, so a workgroup stays on "its" frames' XCD)
__device__ __forceinline__ void s3_block_map(const S3Args& a, int q, int& bi, int& bx) {
  if ((a.n_frames & 7) == 0) {
    const int r = q >> 3;
    bi = (q & 7) + 8 * (r / a.bpf);
    bx = r % a.bpf;
  } else {
    bi = q / a.bpf;
    bx = q % a.bpf;
  }
}

__device__ __forceinline__ void split4(const float (&x)[4], uint2& h, uint2& m, uint2& l);

// ---- fp16 x 2: the power-of-two scales of a chain, computed by every thread from the device-side input bound ----------
struct S3Scales {
  float s_in[S3_MAX_LAYERS];      // scale of layer l's input (activations)
  float bias_mul[S3_MAX_LAYERS];  // sw_l * s_in[l]: what the layer's accumulators carry
  float next_mul[S3_MAX_LAYERS];  // s_in[l + 1] / bias_mul[l]: accumulator -> next layer's input (last layer: 1 / bias_mul)
};
__device__ __forceinline__ float s3_pow2_scale(float bound) {      // largest power of two s with bound * s <= 2^14
  int e;
  (void)frexpf(fmaxf(bound, 1e-30f), &e);                          // bound = m 2^e, m in [0.5, 1)
  return ldexpf(1.f, 14 - e);
}
__device__ __forceinline__ S3Scales s3_scales(const S3Args& a) {
  S3Scales sc;
  float B = a.bound_a ? *a.bound_a : 1.f;
  if (a.bound_b) B = fmaxf(B, a.mul_b * *a.bound_b);
  // (an all-zero input table has bound 0: with the 1e-30 floor of s3_pow2_scale the scale products sw * s_in reached inf and
  // bias * inf = NaN for the zero pad entries; 2^-60 keeps every product finite -- s_in <= 2^74, sw <= 2^40 -- and changes
  // nothing for a table whose bound is a real value)
  B = fmaxf(B, 8.67e-19f);
#pragma unroll
  for (int l = 0; l < S3_MAX_LAYERS; ++l) {
    if (l < a.n_layers) {
      sc.s_in[l] = s3_pow2_scale(B);
      sc.bias_mul[l] = a.sw[l] * sc.s_in[l];
      B = (a.wnorm[l] * B + a.bmax[l]) * 1.01f;                    // |y| <= ||W||_inf max|x| + max|b| (+ rounding slack)
    } else {
      sc.s_in[l] = 1.f; sc.bias_mul[l] = 1.f;
    }
  }
#pragma unroll
  for (int l = 0; l < S3_MAX_LAYERS; ++l)
    sc.next_mul[l] = l + 1 < a.n_layers ? sc.s_in[l + 1] / sc.bias_mul[l] : 1.f / sc.bias_mul[l];
  return sc;
}
// relu on the bit pattern: one v_max_i32 (a float is negative iff its pattern is a negative integer; fmaxf(x, 0) costs a
// second, canonicalising v_max per value because the compiler cannot know that an MFMA result is no signalling NaN)
__device__ __forceinline__ float s3_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// two fp16 pieces of four (already scaled) fp32 values, both rounded to nearest (v_cvt_pk_f16_f32): h = fp16(x),
// l = fp16(x - h) -- the residual x - h is exact in fp32, |x - h - l| <= 2^-22 |x|, and the rounding is symmetric (a
// truncating split, one instruction cheaper per pair, biases every post-ReLU activation downwards: the sums over
// thousands of points that the module-level tests check drifted by 1e-6 of their magnitude)
typedef __attribute__((ext_vector_type(2))) _Float16 s3_h2;
typedef __attribute__((ext_vector_type(2))) float s3_f2;
__device__ __forceinline__ void split4h(const float (&x)[4], uint2& h, uint2& l) {
  pvn3d_split2_f16(x[0], x[1], h.x, l.x);            // (common.h: three instructions per pair)
  pvn3d_split2_f16(x[2], x[3], h.y, l.y);
}
// store four consecutive-k values as pieces: planes `plane` bytes apart
template <int AR>
__device__ __forceinline__ void s3_put4(char* d, size_t plane, const float (&x)[4], float mul) {
  if (AR == 1) {
    const float y[4] = {x[0] * mul, x[1] * mul, x[2] * mul, x[3] * mul};
    uint2 h, l;
    split4h(y, h, l);
    *reinterpret_cast<uint2*>(d) = h;
    *reinterpret_cast<uint2*>(d + plane) = l;
  } else {
    uint2 h, m, l;
    split4(x, h, m, l);
    *reinterpret_cast<uint2*>(d) = h;
    *reinterpret_cast<uint2*>(d + plane) = m;
    *reinterpret_cast<uint2*>(d + 2 * plane) = l;
  }
}

// ---- exact 3-way split of four fp32 values (consecutive k) into three packed bf16x4 ---------------------------------
__device__ __forceinline__ void split4(const float (&x)[4], uint2& h, uint2& m, uint2& l) {
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    hb[i] = __float_as_uint(x[i]) & 0xffff0000u;
    const float r1 = x[i] - __uint_as_float(hb[i]);
    mb[i] = __float_as_uint(r1) & 0xffff0000u;
    lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));      // <= 8 significant bits: its top half-word is exact
  }
  h.x = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u); h.y = __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u);
  m.x = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u); m.y = __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u);
  l.x = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u); l.y = __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u);
}

// ---- loader waves -----------------------------------------------------------------------------------------------------
// One loader wave stages one whole chunk (32 channels x 64 columns, or the <= 8-channel tail): lane -> row group
// g = lane & 7 (channels 4g..4g+3), columns (lane >> 3) + 8 i, i < 8.  The per-column gather information (neighbour
// indices, interpolation weights) is loaded ONCE per column block into registers (LoaderCols), so that a chunk costs one
// global round trip, not two.
template <bool IS_SA>
struct LoaderCols {
  int id[8][IS_SA ? 1 : 3];
  float w[8][IS_SA ? 1 : 3];
};

template <bool IS_SA>
__device__ __forceinline__ void loader_cols(const S3Args& a, LoaderCols<IS_SA>& lcx, int bi, int col0, int lane) {
  const int cq = lane >> 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int gc = min(col0 + cq + 8 * i, a.cols_total - 1);
    if (IS_SA) {
      lcx.id[i][0] = S3_DBG(a, 2) ? gc % a.rowsA : a.idx[(size_t)bi * a.cols_total + gc];
      lcx.w[i][0] = 1.f;
    } else {
      const size_t o = ((size_t)bi * a.cols_total + gc) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) { lcx.id[i][k] = a.idx[o + k]; lcx.w[i][k] = a.weight[o + k]; }
    }
  }
}

template <bool IS_SA, int AR>
__device__ __forceinline__ void loader_chunk(const S3Args& a, const LoaderCols<IS_SA>& lcx, char* slot, int bi, int col0,
                                             int lc /*local chunk*/, int lane, float s0) {
  const int g = lane & 7, cq = lane >> 3;
  const int n_full = a.nA + a.nB;
  if (lc < n_full) {
    const bool fromA = lc < a.nA;
    const float* tab = fromA ? a.tabA : a.tabB;
    const int rows = fromA ? a.rowsA : a.rowsB, ld = fromA ? a.ldA : a.ldB;
    const int cbase = (fromA ? lc : lc - a.nA) * S3_KC + 4 * g;
    const float* t = tab + (size_t)bi * rows * ld + cbase;
    float4 v[8];
    if (IS_SA || !fromA) {
      // one source row per column: SA neighbour index / FP the unknown point itself
      if (S3_DBG(a, 1)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = make_float4(0.25f * lcx.id[i][0], 1.f, 2.f, 3.f);
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int id = IS_SA ? lcx.id[i][0] : min(col0 + cq + 8 * i, a.cols_total - 1);
          v[i] = *reinterpret_cast<const float4*>(t + (size_t)id * ld);
        }
      }
    } else {
      // three_interpolate (pointnet2_utils.py:136-170): p0*w0 + p1*w1 + p2*w2, unfused, in this order.  All 24 row
      // segments of the chunk are requested before the first is used: one round trip per chunk
      constexpr int K1 = IS_SA ? 0 : 1, K2 = IS_SA ? 0 : 2;
      float4 p[8][IS_SA ? 1 : 3];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int k = 0; k < (IS_SA ? 1 : 3); ++k) p[i][k] = *reinterpret_cast<const float4*>(t + (size_t)lcx.id[i][k] * ld);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float w0 = lcx.w[i][0], w1 = lcx.w[i][K1], w2 = lcx.w[i][K2];
        v[i].x = p[i][0].x * w0 + p[i][K1].x * w1 + p[i][K2].x * w2;
        v[i].y = p[i][0].y * w0 + p[i][K1].y * w1 + p[i][K2].y * w2;
        v[i].z = p[i][0].z * w0 + p[i][K1].z * w1 + p[i][K2].z * w2;
        v[i].w = p[i][0].w * w0 + p[i][K1].w * w1 + p[i][K2].w * w2;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cq + 8 * i;
      const bool ok = col0 + c < a.cols_total;
      const float x[4] = {ok ? v[i].x : 0.f, ok ? v[i].y : 0.f, ok ? v[i].z : 0.f, ok ? v[i].w : 0.f};
      s3_put4<AR>(slot + c * S3_CS + 8 * g, S3_CPS, x, s0);
    }
  } else {
    // tail chunk: one 16-k slab, channels [0, tail_w) real, the rest zero.  lane -> column lane, rows 0..15
    float x[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) x[k] = 0.f;
    const int gc = col0 + lane;
    if (gc < a.cols_total) {
      if (IS_SA) {
        const int id = a.idx[(size_t)bi * a.cols_total + gc];
        const float* p = a.xyz + ((size_t)bi * a.n + id) * 3;
        const float* c = a.new_xyz + ((size_t)bi * a.m + gc / a.ns) * 3;
        x[0] = p[0] - c[0]; x[1] = p[1] - c[1]; x[2] = p[2] - c[2];      // grouped_xyz -= new_xyz
      } else {
        const float* p = a.tabB + ((size_t)bi * a.rowsB + gc) * a.ldB + a.nB * S3_KC;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k < a.tail_w) x[k] = p[k];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float y[4] = {x[4 * q], x[4 * q + 1], x[4 * q + 2], x[4 * q + 3]};
      s3_put4<AR>(slot + lane * S3_CS + 8 * q, S3_CPS, y, s0);
    }
  }
}

// Set abstraction: a chunk in two steps, so that a loader wave can keep TWO chunks' gathers in flight (round 6).  Layer 0
// of a column block is bound by the loaders' concurrency, not by bandwidth: a wave with one chunk -- eight 16-byte loads per
// lane, 8 KB -- in flight per ~4.8 k-cycle L2 round trip moves 6.7 B per cycle and CU for all four loaders, which is
// exactly the 5.4 k cycles the MFMA waves spend waiting in layer 0 of a SA level 2 block (r05_s3_prof.txt: 1.7 k of MFMAs).
// sa_fetch issues the loads of one chunk into a register set, sa_store scales / splits / writes it; the loop alternates
// two sets, so the second chunk's round trip runs under the first one's.
struct SaChunk {
  float4 v[8];       // full chunk: this lane's 16-byte row segment of eight columns
  float x[3];        // tail chunk: this lane's column, relative coordinates
};
__device__ __forceinline__ void sa_fetch(const S3Args& a, const int (&id)[8], SaChunk& d, int bi, int col0, int lc, int lane) {
  const int g = lane & 7;
  if (lc < a.nA) {
    const float* t = a.tabA + (size_t)bi * a.rowsA * a.ldA + lc * S3_KC + 4 * g;
    if (S3_DBG(a, 1)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) d.v[i] = make_float4(0.25f * id[i], 1.f, 2.f, 3.f);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) d.v[i] = *reinterpret_cast<const float4*>(t + (size_t)id[i] * a.ldA);
    }
  } else {
    d.x[0] = d.x[1] = d.x[2] = 0.f;
    const int gc = col0 + lane;
    if (gc < a.cols_total) {
      const int pid = a.idx[(size_t)bi * a.cols_total + gc];
      const float* p = a.xyz + ((size_t)bi * a.n + pid) * 3;
      const float* c = a.new_xyz + ((size_t)bi * a.m + gc / a.ns) * 3;
      d.x[0] = p[0] - c[0]; d.x[1] = p[1] - c[1]; d.x[2] = p[2] - c[2];      // grouped_xyz -= new_xyz
    }
  }
}
template <int AR>
__device__ __forceinline__ void sa_store(const S3Args& a, const SaChunk& d, char* slot, int col0, int lc, int lane, float s0) {
  const int g = lane & 7, cq = lane >> 3;
  if (lc < a.nA) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = cq + 8 * i;
      const bool ok = col0 + c < a.cols_total;
      const float x[4] = {ok ? d.v[i].x : 0.f, ok ? d.v[i].y : 0.f, ok ? d.v[i].z : 0.f, ok ? d.v[i].w : 0.f};
      s3_put4<AR>(slot + c * S3_CS + 8 * g, S3_CPS, x, s0);
    }
  } else {
    // tail chunk: one 16-k slab, rows 0..2 the relative coordinates, the rest zero.  lane -> column lane, rows 0..15
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float y[4] = {q == 0 ? d.x[0] : 0.f, q == 0 ? d.x[1] : 0.f, q == 0 ? d.x[2] : 0.f, 0.f};
      s3_put4<AR>(slot + lane * S3_CS + 8 * q, S3_CPS, y, s0);
    }
  }
}

// ---- MFMA waves -------------------------------------------------------------------------------------------------------
// A layer's packed weights as the MFMA loops address them: a wave-uniform (SGPR) base that advances by one slab, plus
// one 32-bit per-lane byte offset per row tile; the three pieces of a tile sit 1 KiB apart (instruction offsets).
// Nothing of an A-fragment address is computed on the VALU inside the loops -- on this chip a wave's VALU work does
// not overlap its own MFMAs, so every vector instruction in a slab is time taken from the matrix pipe.
template <int NTC>
struct WSrc {
  const char* sbase;        // layer's weights (uniform)
  unsigned voff[NTC];       // (tile * 3 * 64 + lane) * 16
  unsigned sstride;         // bytes per slab = mt_total * 3 * 64 * 16
  int last;                 // last slab
};

// the partial products of one slab (six bf16 / three fp16), smallest terms first; consecutive MFMAs hit different
// accumulators.  Fragments travel as uint4 (eight 16-bit pieces of consecutive k).
// TR: the transposed product (activation fragment as the A operand, weight fragment as B -- the two fragment layouts of
// v_mfma_f32_32x32x16 are the same, so the registers are simply swapped in the instruction): the accumulator then holds
// lane = output channel, registers = the tile's 32 columns.
template <int AR, int NTC, int NT, int NR, bool TR = false>
__device__ __forceinline__ void mm_slab(f32x16 (&acc)[NT][2], const uint4 (&a)[NR][s3_np(AR)], const uint4 (&b)[2][s3_np(AR)],
                                        bool a_lo_zero = false) {
  if (AR == 1 && TR) {
#define S3_MM(PA, PB)                                                                                                  \
  _Pragma("unroll") for (int t = 0; t < NTC; ++t) _Pragma("unroll") for (int c = 0; c < 2; ++c)                        \
      acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, b[c][PB]),                          \
                                                         __builtin_bit_cast(f16x8, a[t][PA]), acc[t][c], 0, 0, 0)
    S3_MM(0, 1); S3_MM(1, 0); S3_MM(0, 0);
#undef S3_MM
  } else if (AR == 1) {
#define S3_MM(PA, PB)                                                                                                  \
  _Pragma("unroll") for (int t = 0; t < NTC; ++t) _Pragma("unroll") for (int c = 0; c < 2; ++c)                        \
      acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[t][PA]),                          \
                                                         __builtin_bit_cast(f16x8, b[c][PB]), acc[t][c], 0, 0, 0)
    // (a_lo_zero, wave-uniform: the weight slab's low piece is all zeros -- the identity block of a pre-contracted chain --
    // so its product with the activations' high piece adds exact zeros and is not issued: a third of the block's MFMAs)
    S3_MM(0, 1);
    if (!a_lo_zero) S3_MM(1, 0);
    S3_MM(0, 0);
#undef S3_MM
  } else {
#define S3_MM(PA, PB)                                                                                                  \
  _Pragma("unroll") for (int t = 0; t < NTC; ++t) _Pragma("unroll") for (int c = 0; c < 2; ++c)                        \
      acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[t][PA]),                        \
                                                          __builtin_bit_cast(bf16x8, b[c][PB]), acc[t][c], 0, 0, 0)
    S3_MM(0, 2); S3_MM(2, 0); S3_MM(1, 1);
    S3_MM(0, 1); S3_MM(1, 0); S3_MM(0, 0);
#undef S3_MM
  }
}

__device__ __forceinline__ void acc_bias(f32x16& acc, const float* __restrict__ sb, int half) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(sb + 8 * g + 4 * half);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
}

// One layer's MFMA loop shapes: NTC row tiles (tile0 + 4 t, clamped to the layer's last tile: a wave without a t-th
// tile recomputes the last one and drops the result) x both column tiles.
// PASSES: the last layer's row tiles are worked off in PASSES rounds of 4 * NLAST tiles (512-wide last layers: the
// accumulators of all 16 tiles would not fit); its input P is only read, so the rounds need no barrier between them,
// and their max-pool runs on DPP lane shifts instead of LDS patches (P is still being read by the other waves).
// TRL: the last layer transposed with the max-pool in registers (pool_tr; fp16 x 2 set abstraction, nsample 16 / 32 / 64) --
// an instantiation of its own, so that neither form carries the other's epilogue in its register budget
template <bool IS_SA, int N0, int N1, int N2, int PASSES, int AR, bool TRL = false>
struct S3Consumer {
  static constexpr int NP = s3_np(AR);
  static constexpr int CHUNK = s3_chunk(AR);
  static constexpr int NMAX = N0 > N1 ? (N0 > N2 ? N0 : N2) : (N1 > N2 ? N1 : N2);
  static constexpr int NL = N2 > 0 ? 3 : 2;
  const S3Args& a;
  char* P;
  char* ring;
  const float* s_bias;
  const float* s_om;       // per row of the last layer: accumulator -> output
  S3Ctl* ctl;
  S3Scales sc;             // fp16 x 2: power-of-two scales (all 1 for bf16 x 3)
  float amax;              // running max |output| of this lane (out_absmax)
  int lane_, wave;
  unsigned phase;          // barrier arrivals expected so far
  unsigned chunk_no;       // chunks of this workgroup consumed so far (all blocks)
  int blk_no;              // column blocks done (probe only)
  // layer 0's xyz slab of a pre-contracted set-abstraction chain: the only weights of that layer that are not the unit
  // matrix, the same for every column block -- loaded once per workgroup (with the identity chunks worked off in a few
  // hundred cycles, their L2 round trip at the head of every block was 2.4 k of SA level 2's 23.8 k cycles per block)
  uint4 w0t[2][s3_np(AR)];
  int w0t_ok;

  template <int NTC>
  __device__ __forceinline__ WSrc<NTC> wsrc(int l, int slabs, int lane, int tile_base = 0) const {
    const int mt_total = (a.M[l] + 31) >> 5;
    WSrc<NTC> w;
    w.sbase = reinterpret_cast<const char*>(a.W[l]);
    w.sstride = (unsigned)mt_total * (unsigned)NP * 64u * 16u;
    w.last = slabs - 1;
#pragma unroll
    for (int t = 0; t < NTC; ++t)
      w.voff[t] = ((unsigned)min(tile_base + wave + S3_NWC * t, mt_total - 1) * (unsigned)(NP * 64) + (unsigned)lane) * 16u;
    return w;
  }
  template <int NTC, int NR>
  __device__ __forceinline__ void a_load(uint4 (&r)[NR][NP], const WSrc<NTC>& w, int slab) const {
    const char* sb = w.sbase + (size_t)((unsigned)min(slab, w.last) * w.sstride);      // scalar
#pragma unroll
    for (int t = 0; t < NTC; ++t)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) r[t][pc] = *reinterpret_cast<const uint4*>(sb + w.voff[t] + pc * 1024);
  }
  // LDS byte addresses of this lane's B fragments: [column tile][piece]; a slab is + 32 bytes (an instruction offset)
  struct BSrc {
    const char* p[2][NP];
  };
  __device__ __forceinline__ BSrc bsrc(const char* base, int ps, int ct_bytes) const {
    BSrc b;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) b.p[c][pc] = base + (size_t)c * ct_bytes + (size_t)pc * ps;
    return b;
  }
  template <int OFF>
  __device__ __forceinline__ void b_ld(uint4 (&b)[2][NP], const BSrc& src) const {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int pc = 0; pc < NP; ++pc) b[c][pc] = *reinterpret_cast<const uint4*>(src.p[c][pc] + OFF);
  }
  __device__ __forceinline__ void tiles_bias(int l, int (&tile)[NMAX], int n) const {
    const int mt_total = (a.M[l] + 31) >> 5;
#pragma unroll
    for (int t = 0; t < NMAX; ++t) tile[t] = t < n ? min(wave + S3_NWC * t, mt_total - 1) : 0;
  }
  template <int NTC>
  __device__ __forceinline__ void init_acc(f32x16 (&acc)[NMAX][2], int l, int boff, int half, int tile_base = 0) const {
    const int mt_total = (a.M[l] + 31) >> 5;
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      acc_bias(acc[t][0], s_bias + boff + min(tile_base + wave + S3_NWC * t, mt_total - 1) * 32, half);
      acc[t][1] = acc[t][0];
    }
  }

  // Weight fragments: a ring of four slab slots per wave, slab s in slot s & 3; at the start of slab s the slot slab
  // s - 1 has just left is refilled with slab s + 3: three slabs of MFMAs cover the L2 round trip (~2k cycles with
  // every CU streaming weights).
  // layer 0: the input arrives chunk by chunk (two slabs) from the loader waves.
  template <int NTC>
  __device__ __forceinline__ void layer0(f32x16 (&acc)[NMAX][2], int lane) {
    static_assert(NTC <= 2, "layer 0 runs the four-slot weight ring");
    const int half = lane >> 5, col = lane & 31;
    const int n_full = a.nA + a.nB;
    const bool tail = a.tail_w > 0;
    const int slabs = 2 * n_full + (tail ? 1 : 0);
    if (wave >= ((a.M[0] + 31) >> 5)) {
      // no row tile of layer 0: keep the ring's hand-over protocol only -- a slot's use counts as finished by this wave
      // once the chunk is there (waiting for it keeps the wave from running ahead of the waves that do read, whose
      // counts the loaders rely on before they refill a slot)
      const int n_ch = n_full + (tail ? 1 : 0);
      // (an identity chunk is read, and counted as finished for all four MFMA waves, by the wave whose row tile it concerns)
      for (int c = (AR == 1 && a.ident_a != 0) ? a.nA : 0; c < n_ch; ++c) {
        const unsigned cn = chunk_no + c;
        const int slot = cn & (S3_RING - 1);
        lds_wait_ge(&ctl->rdy[slot], cn + 1);
        lds_signal_add(&ctl->fin[slot], lane);
      }
      chunk_no += n_ch;
      return;
    }
    const WSrc<NTC> w = wsrc<NTC>(0, slabs, lane);
    init_acc<NTC>(acc, 0, 0, half);
    // PVN3D_MLP_IDENTITY_A (pre-contracted chains): the a.nA chunks of the gathered table meet an identity block.  Row tile
    // mt has non-zero weights in chunk mt only, and there they are the (scaled) unit matrix: those chunks are worked off
    // first, by the one wave whose tile they concern, with constant weight fragments and without the zero low piece -- eight
    // MFMAs per chunk and workgroup instead of 12 per slab and row tile (FP level 1: 64 instead of 768 per column block).
    // The chunks that follow (skip features / xyz) run the pipelined loop below from chunk c0 on.
    const bool idz_ = AR == 1 && a.ident_a != 0;
    const int c0 = idz_ ? a.nA : 0;
    const bool tail_only = idz_ && c0 == n_full;       // nothing but the xyz slab behind the identity chunks
    uint4 ringA[4][NTC][NP];
    if (tail_only) {
      if (tail && !w0t_ok) { a_load<NTC, 2>(w0t, w, 2 * c0); w0t_ok = 1; }
    } else {
      a_load<NTC, NTC>(ringA[0], w, 2 * c0);
      a_load<NTC, NTC>(ringA[1], w, 2 * c0 + 1);
      a_load<NTC, NTC>(ringA[2], w, 2 * c0 + 2);
    }
    // this lane's fragment addresses in ring slot 0; slot k is + k * CHUNK
    const BSrc bs0 = bsrc(ring + col * S3_CS + half * 16, S3_CPS, 32 * S3_CS);
    const int n_chunks = n_full + (tail ? 1 : 0);
    if constexpr (AR == 1) {
      if (idz_) {
        // unit-matrix fragments of a 32-row tile's two slabs: lane (row m, k half h) holds k = 16 j + 8 h + i, i < 8, of
        // slab j: the weight scale (a power of two, exact in fp16) where k == m
        const int m = lane & 31;
        const unsigned one = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)a.sw[0]);
        uint4 idA[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int i = m - 16 * j - 8 * half;
          const unsigned v = (i >= 0 && i < 8) ? one << (16 * (i & 1)) : 0u;
          idA[j] = make_uint4((i >> 1) == 0 ? v : 0u, (i >> 1) == 1 ? v : 0u, (i >> 1) == 2 ? v : 0u, (i >> 1) == 3 ? v : 0u);
        }
        S3_STAMP((wave == 0 && blk_no == 40 ? 32 : 1 << 20));                             // (probe builds: phase I of block 40 starts)
        // chunk C is read by ONE wave -- the one whose row tile wave + 4 t equals C -- which also reports it finished for all
        // four MFMA waves (the loaders wait for four reports per use of a slot): the other waves neither wait for it nor
        // touch its control words (a poll and a fenced LDS atomic per chunk and wave were 550 cycles each: 1.6 k of SA
        // level 2's 4.0 k cycles of layer 0, twice that at SA level 3)
#pragma unroll
        for (int t = 0; t < NTC; ++t) {
          const int C = wave + S3_NWC * t;
          if (C < c0) {
            const unsigned cn = chunk_no + C;
            const int slot = cn & (S3_RING - 1);
            lds_wait_ge(&ctl->rdy[slot], cn + 1);
            BSrc bs;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
              for (int p2 = 0; p2 < NP; ++p2) bs.p[c2][p2] = bs0.p[c2][p2] + slot * CHUNK;
            uint4 b0[2][NP], b1[2][NP];
            b_ld<0>(b0, bs);
            b_ld<32>(b1, bs);
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
              // (the dense loop's order per accumulator: slab 2 C -- hi x lo, hi x hi --, then slab 2 C + 1)
              acc[t][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, idA[0]),
                                                                  __builtin_bit_cast(f16x8, b0[c2][1]), acc[t][c2], 0, 0, 0);
              acc[t][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, idA[0]),
                                                                  __builtin_bit_cast(f16x8, b0[c2][0]), acc[t][c2], 0, 0, 0);
              acc[t][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, idA[1]),
                                                                  __builtin_bit_cast(f16x8, b1[c2][1]), acc[t][c2], 0, 0, 0);
              acc[t][c2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, idA[1]),
                                                                  __builtin_bit_cast(f16x8, b1[c2][0]), acc[t][c2], 0, 0, 0);
            }
            lds_signal_add(&ctl->fin[slot], lane, (unsigned)S3_NWC);
            S3_STAMP((wave == 0 && blk_no == 40 && t < 6 ? 33 + t : 1 << 20));             // (probe builds: own chunk t done)
          }
        }
        S3_STAMP((wave == 0 ? blk_no * 8 : 1 << 20) + 6);          // (probe builds: identity chunks done)
      }
    }
    // Software pipeline over the chunks: the fragments of a chunk's first slab are requested during the previous
    // chunk (after its first slab's MFMAs have been issued, so the ready poll and the LDS round trip run under MFMAs
    // that are already in the pipe); a chunk step therefore starts with b0 in registers.
    uint4 bA[2][NP], bB[2][NP];           // first-slab fragments of the current / next chunk, second-slab fragments
    {
      const int slot = (chunk_no + c0) & (S3_RING - 1);
      lds_wait_ge(&ctl->rdy[slot], chunk_no + c0 + 1);
      BSrc bs;
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int p2 = 0; p2 < NP; ++p2) bs.p[c2][p2] = bs0.p[c2][p2] + slot * CHUNK;
      b_ld<0>(bA, bs);
    }
    // one full chunk C = slabs 2C (weight slot U0) and 2C + 1 (slot U0 + 1); NEXT: there is a chunk C + 1 in this block
#define S3_CHUNK_STEP(C, U0)                                                                           \
  do {                                                                                                 \
    const unsigned cn_ = chunk_no + (C);                                                               \
    const int slot_ = cn_ & (S3_RING - 1);                                                             \
    BSrc bs_;                                                                                          \
    _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) \
        bs_.p[c_][p_] = bs0.p[c_][p_] + slot_ * CHUNK;                                                 \
    a_load<NTC, NTC>(ringA[((U0) + 3) & 3], w, 2 * (C) + 3);                                                \
    b_ld<32>(bB, bs_);                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    mm_slab<AR, NTC, NMAX>(acc, ringA[(U0)], bA);                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    a_load<NTC, NTC>(ringA[((U0) + 4) & 3], w, 2 * (C) + 4);                                                \
    if ((C) + 1 < n_chunks) {                                                                          \
      const int slotn_ = (cn_ + 1) & (S3_RING - 1);                                                    \
      lds_wait_ge(&ctl->rdy[slotn_], cn_ + 2);                                                         \
      BSrc bn_;                                                                                        \
      _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) _Pragma("unroll") for (int p_ = 0; p_ < NP; ++p_) \
          bn_.p[c_][p_] = bs0.p[c_][p_] + slotn_ * CHUNK;                                              \
      b_ld<0>(bA, bn_);                                                                                \
    }                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    mm_slab<AR, NTC, NMAX>(acc, ringA[(U0) + 1], bB);                                               \
    lds_signal_add(&ctl->fin[slot_], lane);                                                            \
  } while (0)
    int c = c0;
    for (; c + 2 <= n_full; c += 2) {
      S3_CHUNK_STEP(c, 0);
      S3_CHUNK_STEP(c + 1, 2);
    }
    const bool odd = c < n_full;
    if (odd) S3_CHUNK_STEP(c, 0);
#undef S3_CHUNK_STEP
    if (tail) {
      // the tail chunk's single slab: its fragments are already in bA
      const unsigned cn = chunk_no + n_full;
      const int slot = cn & (S3_RING - 1);
      if (tail_only) mm_slab<AR, NTC, NMAX, 2>(acc, w0t, bA);
      else if (odd) mm_slab<AR, NTC, NMAX>(acc, ringA[2], bA);
      else mm_slab<AR, NTC, NMAX>(acc, ringA[0], bA);
      lds_signal_add(&ctl->fin[slot], lane);
    }
    chunk_no += n_full + (tail ? 1 : 0);
    S3_STAMP((wave == 0 && blk_no == 40 ? 39 : 1 << 20));                                 // (probe builds: layer 0 of block 40 done)
  }

  // layers >= 1: the input is P.  Three row tiles per wave (NTC = 3: 96 accumulator registers) run a two-slot weight
  // ring -- one slab of 36 MFMAs (1.1k cycles) of cover -- the others the four-slot ring.  The first RD - 1 slabs of a
  // layer's weights are requested by preloadA BEFORE the barrier + activation store + barrier that precede the layer
  // (the ring registers are free then): the ~2k-cycle L2 round trip runs under the store phase instead of heading the
  // layer (short layers -- 8 slabs -- spent a quarter of their time there).
  template <int NTC>
  __device__ __forceinline__ void preloadA(uint4 (&R)[4][NMAX][NP], int l, int lane, int tile_base = 0) const {
    constexpr int RD = NTC >= 3 ? 2 : 4;
    if (tile_base + wave >= ((a.M[l] + 31) >> 5)) return;        // this wave owns no row tile of the layer
    const int slabs = (a.K[l] + 15) >> 4;
    const WSrc<NTC> w = wsrc<NTC>(l, slabs, lane, tile_base);
#pragma unroll
    for (int u = 0; u < RD - 1; ++u) a_load<NTC, NMAX>(R[u], w, u);
  }
  // TR (fp16 x 2, the last layer of a set-abstraction chain): the transposed product -- accumulators start at zero (an
  // inline constant of the first MFMA) and the bias, one value per lane there, is added by pool_tr
  template <int NTC, bool TR = false>
  __device__ __forceinline__ void layerN(f32x16 (&acc)[NMAX][2], uint4 (&R)[4][NMAX][NP], int l, int boff, int lane,
                                         int tile_base = 0) {
    const int half = lane >> 5, col = lane & 31;
    const int slabs = (a.K[l] + 15) >> 4;
    // a layer narrower than 32 x S3_NWC channels leaves waves without a row tile (SA level 1: 64 channels = 2 tiles): they
    // used to recompute the last tile and drop the result -- matrix-pipe, LDS and L2 traffic for nothing
    if (tile_base + wave >= ((a.M[l] + 31) >> 5)) return;
    const WSrc<NTC> w = wsrc<NTC>(l, slabs, lane, tile_base);
    if (TR) {
#pragma unroll
      for (int t = 0; t < NTC; ++t)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][c2][r] = 0.f;
    } else {
      init_acc<NTC>(acc, l, boff, half, tile_base);
    }
    BSrc bs = bsrc(P + (size_t)col * a.rs + half * 16, a.ps, 32 * a.rs);
    constexpr int RD = NTC >= 3 ? 2 : 4;
    uint4 b[2][2][NP];
    b_ld<0>(b[0], bs);
    // slab S (weight slot U): fragments of slab S + 1 are requested at the offset OFFN from the current bases
#define S3_SLAB_STEP(S, U, OFFN)                                                                       \
  do {                                                                                                 \
    a_load<NTC, NMAX>(R[((U) + RD - 1) % RD], w, (S) + RD - 1);                                        \
    b_ld<(OFFN)>(b[((U) + 1) & 1], bs);                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
    mm_slab<AR, NTC, NMAX, NMAX, TR>(acc, R[(U) % RD], b[(U) & 1]);                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                 \
  } while (0)
    // (the layer's slab count is padded: P holds zero rows up to a multiple of 32 k, and the fragment reads of the
    // slab after the last stay inside the column's row -- rs = 2 * kcap + 16 bytes)
    int s = 0;
    for (; s + 4 <= slabs; s += 4) {
      S3_SLAB_STEP(s, 0, 32);
      S3_SLAB_STEP(s + 1, 1, 64);
      S3_SLAB_STEP(s + 2, 2, 96);
      S3_SLAB_STEP(s + 3, 3, 128);
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int p2 = 0; p2 < NP; ++p2) bs.p[c2][p2] += 128;
    }
    if (s < slabs) S3_SLAB_STEP(s, 0, 32);
    if (s + 1 < slabs) S3_SLAB_STEP(s + 1, 1, 64);
    if (s + 2 < slabs) S3_SLAB_STEP(s + 2, 2, 96);
#undef S3_SLAB_STEP
  }

  // relu(tile) -> three bf16 planes of P: register group g of a tile holds rows mt*32 + 8g + 4*half + 0..3 of the lane's
  // column = four consecutive k of the next layer = one 8-byte store per plane
  __device__ __forceinline__ void store_tile(const f32x16& acc, int mt, int colx, int lane, float mul) {
    const int half = lane >> 5;
    char* d0 = P + (size_t)colx * a.rs + 64 * mt + 8 * half;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float x[4] = {fmaxf(acc[4 * g], 0.f), fmaxf(acc[4 * g + 1], 0.f), fmaxf(acc[4 * g + 2], 0.f),
                          fmaxf(acc[4 * g + 3], 0.f)};
      s3_put4<AR>(d0 + 16 * g, (size_t)a.ps, x, mul);
    }
  }
  template <int NTC>
  __device__ __forceinline__ void store_layer(const f32x16 (&acc)[NMAX][2], int l, int lane) {
    const int mt_total = (a.M[l] + 31) >> 5;
    const int col = lane & 31;
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      const int mt = wave + S3_NWC * t;
      if (mt < mt_total) {
        store_tile(acc[t][0], mt, col, lane, sc.next_mul[l]);
        store_tile(acc[t][1], mt, 32 + col, lane, sc.next_mul[l]);
      }
    }
  }

  // max over the nsample (16, 32 or 64) columns of each centre for one row tile, on DPP lane shifts (no LDS), and the
  // store of the pooled rows: lane 15 (nsample 16) / 31 of each 32-lane half ends up with the result of its centre for
  // the rows 8g + 4*half + 0..3 of the tile -- four 16-byte stores.  The reduction is written as v_max_f32_dpp, one
  // instruction per (register, step), step-major over the 16 registers: consecutive instructions are independent, and a
  // register written in one step is read 15 instructions later (a DPP read needs 2 wait states after a VALU write; the
  // builtin form cost a v_mov + canonicalising v_max + s_nop per step and 15k cycles per two tiles).
  template <int NS>
  __device__ __forceinline__ void pool_dpp_t(const f32x16 (&acc)[2], int mt, int bi, int col0, int lane) {
    const int half = lane >> 5, col = lane & 31;
    const int M = a.M[NL - 1];
    const bool writer = NS == 16 ? (lane & 15) == 15 : (lane & 31) == 31;
#pragma unroll
    for (int ct = 0; ct < (NS == 64 ? 1 : 2); ++ct) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        v[r] = fmaxf(acc[ct][r], 0.f);
        if (NS == 64) v[r] = fmaxf(v[r], fmaxf(acc[1][r], 0.f));
      }
      asm volatile("s_nop 1" ::: "memory");
#define S3_DPP_STEP(CTRL)                                                                                    \
  _Pragma("unroll") for (int r = 0; r < 16; ++r)                                                             \
      asm volatile("v_max_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf" : "+v"(v[r]))
      S3_DPP_STEP("row_ror:1 row_mask:0xf");
      S3_DPP_STEP("row_ror:2 row_mask:0xf");
      S3_DPP_STEP("row_ror:4 row_mask:0xf");
      S3_DPP_STEP("row_ror:8 row_mask:0xf");
      if (NS >= 32) S3_DPP_STEP("row_bcast:15 row_mask:0xa");      // rows 1, 3 take lane 15 of rows 0, 2
#undef S3_DPP_STEP
      asm volatile("s_nop 1" ::: "memory");
      const int centre = NS == 64 ? col0 / 64 : (col0 + 32 * ct + (NS == 16 ? (col & 16) : 0)) / NS;
      if (writer && centre < a.m) {
        float* o = a.out + ((size_t)bi * a.m + centre) * a.ld_out + a.coff + mt * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = mt * 32 + 8 * g + 4 * half;
          const float4 om = *reinterpret_cast<const float4*>(s_om + row);      // accumulator -> output, per row
          const float w[4] = {v[4 * g] * om.x, v[4 * g + 1] * om.y, v[4 * g + 2] * om.z, v[4 * g + 3] * om.w};
          if (row + 3 < M) {
            *reinterpret_cast<float4*>(o + 8 * g) = make_float4(w[0], w[1], w[2], w[3]);
            amax = fmaxf(amax, fmaxf(fmaxf(w[0], w[1]), fmaxf(w[2], w[3])));   // (post-ReLU: >= 0)
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (row + k < M) { o[8 * g + k] = w[k]; amax = fmaxf(amax, w[k]); }
          }
        }
      }
    }
  }
  // Max over the nsample (16 / 32 / 64) columns of each centre for one row tile of a TRANSPOSED last layer (lane = output
  // channel 32 mt + (lane & 31), registers = columns 8 j + 4 half + i of the tile, both column tiles): fifteen integer
  // max per tile in registers -- signed-integer max orders the non-negative floats correctly and ranks every negative one
  // below them, the 0 in the chain is the ReLU -- one exchange between the two halves of the wave, no LDS, no barrier
  // (the [32][S3_EPAD] patch round trip it replaces: 4.1 of the 26.6 k cycles of a SA level 2 column block); a centre's
  // 32 channels of the tile leave as one 128-byte row segment.
  __device__ __forceinline__ void pool_tr(const f32x16 (&acc)[2], int mt, int boff, int bi, int col0, int lane) {
    const int half = lane >> 5, col = lane & 31;
    const int M = a.M[NL - 1], row = mt * 32 + col;
    const float b = s_bias[boff + row];
    const float om = s_om[row];
    const int ns = a.ns;
    int lo[2], hi[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      int k[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) k[r] = __float_as_int(acc[c][r] + b);
      lo[c] = max(max(max(max(k[0], k[1]), max(k[2], k[3])), max(max(k[4], k[5]), max(k[6], k[7]))), 0);        // columns 0..15
      hi[c] = max(max(max(max(k[8], k[9]), max(k[10], k[11])), max(max(k[12], k[13]), max(k[14], k[15]))), 0);  // columns 16..31
    }
    float* const o = a.out + (size_t)bi * a.m * a.ld_out + a.coff + row;
    if (ns == 16) {
      // four centres per 64 columns: (tile c, columns 0..15 | 16..31); each half of the wave stores two of them
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int l2 = max(lo[c], __shfl_xor(lo[c], 32, 64)), h2 = max(hi[c], __shfl_xor(hi[c], 32, 64));
        const int centre = (col0 >> 4) + 2 * c + half;
        const float v = __int_as_float(half ? h2 : l2) * om;
        if (row < M && centre < a.m) { o[(size_t)centre * a.ld_out] = v; amax = fmaxf(amax, v); }
      }
    } else if (ns == 32) {
      const int v0 = max(lo[0], hi[0]), v1 = max(lo[1], hi[1]);
      const int w0 = max(v0, __shfl_xor(v0, 32, 64)), w1 = max(v1, __shfl_xor(v1, 32, 64));
      const int centre = (col0 >> 5) + half;
      const float v = __int_as_float(half ? w1 : w0) * om;
      if (row < M && centre < a.m) { o[(size_t)centre * a.ld_out] = v; amax = fmaxf(amax, v); }
    } else {
      const int v0 = max(max(lo[0], hi[0]), max(lo[1], hi[1]));
      const float v = __int_as_float(max(v0, __shfl_xor(v0, 32, 64))) * om;
      const int centre = col0 >> 6;
      if (half == 0 && row < M && centre < a.m) { o[(size_t)centre * a.ld_out] = v; amax = fmaxf(amax, v); }
    }
  }
  __device__ __forceinline__ void pool_dpp(const f32x16 (&acc)[2], int mt, int bi, int col0, int lane) {
    if (a.ns == 16) pool_dpp_t<16>(acc, mt, bi, col0, lane);
    else if (a.ns == 32) pool_dpp_t<32>(acc, mt, bi, col0, lane);
    else pool_dpp_t<64>(acc, mt, bi, col0, lane);
  }

  // a fresh opaque copy of the lane id per layer: every address of the layer derives from it, nothing can be hoisted
  __device__ __forceinline__ int fresh_lane() const {
    int l = lane_;
    asm volatile("" : "+v"(l));
    return l;
  }

  __device__ __forceinline__ void run_block(int bi, int col0) {
    f32x16 acc[NMAX][2];
    uint4 R[4][NMAX][NP];          // weight-fragment ring of the layers >= 1 (layer 0 has its own)
    int boff = 0;
    [[maybe_unused]] const int pb = wave == 0 ? blk_no * 8 : 1 << 20;
    S3_STAMP(pb + 0);
    {
      const int lane = fresh_lane();
      layer0<N0>(acc, lane);
      preloadA<N1>(R, 1, lane);
      S3_STAMP(pb + 1);
      // every MFMA wave has finished reading its input (and the previous block's epilogue patches in P): P is free
      cbar(ctl, phase, lane);
      S3_STAMP(pb + 2);
      store_layer<N0>(acc, 0, lane);
      cbar(ctl, phase, lane);
      S3_STAMP(pb + 3);
      boff += ((a.M[0] + 31) >> 5) * 32;
    }
    if (NL == 3) {
      const int lane = fresh_lane();
      layerN<N1>(acc, R, 1, boff, lane);
      constexpr int NP2 = (N2 > 0) ? N2 : 1;
      preloadA<NP2>(R, 2, lane);
      S3_STAMP(pb + 4);
      cbar(ctl, phase, lane);
      store_layer<N1>(acc, 1, lane);
      cbar(ctl, phase, lane);
      S3_STAMP(pb + 5);
      boff += ((a.M[1] + 31) >> 5) * 32;
    }
    constexpr int NLAST = NL == 3 ? N2 : N1;
    const int M = a.M[NL - 1];
    const int mt_total = (M + 31) >> 5;
    float* const out = a.out;
    if constexpr (TRL) {
      // ---- fp16 x 2 set abstraction with whole 16 / 32 / 64-column centres: the last layer TRANSPOSED, in PASSES rounds of
      // 4 * NLAST row tiles, max-pool in registers (pool_tr).  P is only read: no barrier between the rounds or before
      // the pool, and nothing is parked in P for the next block to wait for.
#pragma unroll 1
      for (int pass = 0; pass < PASSES; ++pass) {
        const int lane = fresh_lane();
        const int tile_base = pass * S3_NWC * NLAST;
        layerN<NLAST, true>(acc, R, NL - 1, boff, lane, tile_base);
        if (pass + 1 < PASSES) preloadA<NLAST>(R, NL - 1, lane, tile_base + S3_NWC * NLAST);   // under the pool below
#pragma unroll
        for (int t = 0; t < NLAST; ++t) {
          const int mt = tile_base + wave + S3_NWC * t;
          if (mt < mt_total) pool_tr(acc[t], mt, boff, bi, col0, lane);
        }
      }
      ++blk_no;
      S3_STAMP(pb + 7);
      return;
    } else if (PASSES > 1) {
      // ---- last layer in rounds of 4 * NLAST row tiles; max-pool on DPP, no LDS (set abstraction only)
#pragma unroll 1
      for (int pass = 0; pass < PASSES; ++pass) {
        const int lane = fresh_lane();
        const int tile_base = pass * S3_NWC * NLAST;
        layerN<NLAST>(acc, R, NL - 1, boff, lane, tile_base);
        if (pass + 1 < PASSES) preloadA<NLAST>(R, NL - 1, lane, tile_base + S3_NWC * NLAST);   // under the pool below
#pragma unroll
        for (int t = 0; t < NLAST; ++t) {
          const int mt = tile_base + wave + S3_NWC * t;
          if (mt < mt_total) pool_dpp(acc[t], mt, bi, col0, lane);
        }
      }
      ++blk_no;
      S3_STAMP(pb + 7);
      return;
    }
    {
      const int lane = fresh_lane();
      layerN<NLAST>(acc, R, NL - 1, boff, lane);
      S3_STAMP(pb + 6);
      cbar(ctl, phase, lane);              // P is dead: the SA epilogue parks its patches there
    }
    ++blk_no;

    // ---- epilogue on the last layer's accumulators (rows wave + 4 t, both column tiles)
    const int lane = fresh_lane();
    const int half = lane >> 5, col = lane & 31;
    if (IS_SA) {
      // max over the nsample columns of each centre through a wave-private [32][S3_EPAD] patch in P (see sa_mlp.hip;
      // measured against the DPP form of the multi-round kernels: 4.0k vs 7.6k cycles for two tiles)
      float* sc = reinterpret_cast<float*>(P) + (size_t)wave * (32 * S3_EPAD);
      const int ns = a.ns;
      const int nout = ns >= 32 ? 1 : 32 / ns;
      const int jbase = ns >= 64 ? col0 / ns : (col0 + 32 * half) / ns;
#pragma unroll
      for (int t = 0; t < NLAST; ++t) {
        const int mt = wave + S3_NWC * t;
        if (mt < mt_total) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
            sc[rr * S3_EPAD + col] = fmaxf(acc[t][0][r], 0.f);
            sc[rr * S3_EPAD + 32 + col] = fmaxf(acc[t][1][r], 0.f);
          }
          __builtin_amdgcn_wave_barrier();
          int v[32];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int4 q = *reinterpret_cast<const int4*>(sc + col * S3_EPAD + 32 * half + 4 * i);
            v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
          }
          __builtin_amdgcn_wave_barrier();
          if (ns >= 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = max(v[2 * i], v[2 * i + 1]);
          }
          if (ns >= 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = max(v[2 * i], v[2 * i + 1]);
          }
          if (ns >= 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = max(v[2 * i], v[2 * i + 1]);
          }
          if (ns >= 16) { v[0] = max(v[0], v[1]); v[1] = max(v[2], v[3]); }
          if (ns >= 32) v[0] = max(v[0], v[1]);
          if (ns >= 64) v[0] = max(v[0], __shfl_xor(v[0], 32, 64));
          const int row = mt * 32 + col;
          const float om = s_om[row];                    // accumulator -> output of this row
          if (row < M && (ns < 64 || half == 0)) {
            float* o = out + ((size_t)bi * a.m + jbase) * a.ld_out + a.coff + row;
            if (nout <= 2) {
              if (jbase < a.m) { o[0] = __int_as_float(v[0]) * om; amax = fmaxf(amax, __int_as_float(v[0]) * om); }
              if (nout == 2 && jbase + 1 < a.m) { o[a.ld_out] = __int_as_float(v[1]) * om; amax = fmaxf(amax, __int_as_float(v[1]) * om); }
            } else {
#pragma unroll
              for (int q = 0; q < 32; ++q)
                if (q < nout && jbase + q < a.m) { o[(size_t)q * a.ld_out] = __int_as_float(v[q]) * om; amax = fmaxf(amax, __int_as_float(v[q]) * om); }
            }
          }
        }
      }
    } else {
      // (16-byte stores need the row's channel offset and stride to be multiples of four floats and an aligned base)
      const bool pm_vec = a.point_major && ((a.ld_out | a.coff) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
#pragma unroll
      for (int t = 0; t < NLAST; ++t) {
        const int mt = wave + S3_NWC * t;
        if (mt < mt_total) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int row = mt * 32 + 8 * g + 4 * half;
            const int g0 = col0 + col, g1 = col0 + 32 + col;
            const float4 om4 = *reinterpret_cast<const float4*>(s_om + row);     // accumulator -> output, per row
            const float omk[4] = {om4.x, om4.y, om4.z, om4.w};
            if (pm_vec && row + 3 < M) {
              // point-major rows, four consecutive channels of a point: one 16-byte store per column tile (written value
              // by value the compiler cannot prove the alignment and emits four 4-byte stores per lane and row group --
              // 64 scattered dword stores per lane and block: 19 k of FP level 1's 73 k cycles per block)
              float v0[4], v1[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                v0[k] = fmaxf(acc[t][0][4 * g + k], 0.f) * omk[k];
                v1[k] = fmaxf(acc[t][1][4 * g + k], 0.f) * omk[k];
              }
              if (g0 < a.cols_total) {
                amax = fmaxf(amax, fmaxf(fmaxf(v0[0], v0[1]), fmaxf(v0[2], v0[3])));
                *reinterpret_cast<float4*>(out + ((size_t)bi * a.cols_total + g0) * a.ld_out + a.coff + row) =
                    make_float4(v0[0], v0[1], v0[2], v0[3]);
              }
              if (g1 < a.cols_total) {
                amax = fmaxf(amax, fmaxf(fmaxf(v1[0], v1[1]), fmaxf(v1[2], v1[3])));
                *reinterpret_cast<float4*>(out + ((size_t)bi * a.cols_total + g1) * a.ld_out + a.coff + row) =
                    make_float4(v1[0], v1[1], v1[2], v1[3]);
              }
              continue;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (row + k < M) {
                const float om = omk[k];
                const float v0 = fmaxf(acc[t][0][4 * g + k], 0.f) * om, v1 = fmaxf(acc[t][1][4 * g + k], 0.f) * om;
                if (g0 < a.cols_total) amax = fmaxf(amax, v0);
                if (g1 < a.cols_total) amax = fmaxf(amax, v1);
                if (a.point_major) {
                  if (g0 < a.cols_total) out[((size_t)bi * a.cols_total + g0) * a.ld_out + a.coff + row + k] = v0;
                  if (g1 < a.cols_total) out[((size_t)bi * a.cols_total + g1) * a.ld_out + a.coff + row + k] = v1;
                } else {
                  if (g0 < a.cols_total) out[((size_t)bi * M + row + k) * a.cols_total + g0] = v0;
                  if (g1 < a.cols_total) out[((size_t)bi * M + row + k) * a.cols_total + g1] = v1;
                }
              }
            }
          }
        }
      }
    }
    S3_STAMP(pb + 7);
  }
};

template <bool IS_SA, int N0, int N1, int N2, int PASSES, int AR, bool TRL = false>
__global__ __launch_bounds__(S3_THREADS, 1) void mlp_chain_s3_kernel(S3Args a) {
  // 16-byte aligned dynamic LDS (every fragment read is a ds_read_b128: a base that is only 8-byte aligned -- what a
  // static __shared__ object in front of it produces -- turns each of them into a slow misaligned access); the control
  // words live in a 64-byte static block so that the dynamic segment starts on a multiple of 16
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  __shared__ __attribute__((aligned(64))) S3Ctl ctl_storage[2];      // 2 x 40 bytes -> padded to 128
  S3Ctl& ctl = ctl_storage[0];
  char* P = s_mem;
  char* ring = s_mem + a.ring_off;
  float* s_bias = reinterpret_cast<float*>(s_mem + a.bias_off);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < (int)(sizeof(S3Ctl) / 4)) reinterpret_cast<unsigned*>(&ctl)[tid] = 0u;
  S3Scales sc;
  if (AR == 1) {
    sc = s3_scales(a);
  } else {
#pragma unroll
    for (int l = 0; l < S3_MAX_LAYERS; ++l) sc.s_in[l] = sc.bias_mul[l] = sc.next_mul[l] = 1.f;
  }
  {   // all biases of the chain -> LDS, once per (persistent) workgroup (fp16 x 2: in the accumulators' scale, exact)
    int off = 0;
#pragma unroll
    for (int l = 0; l < S3_MAX_LAYERS; ++l) {          // (constant trip count: the scales stay in registers)
      if (l < a.n_layers) {
        const int mp = ((a.M[l] + 31) >> 5) << 5;
        const float bm = sc.bias_mul[l];
        for (int i = tid; i < mp; i += S3_THREADS) s_bias[off + i] = a.bias[l][i] * bm;
        off += mp;
      }
    }
    // accumulator -> output of the last layer, per row: the chain's own (exact power-of-two) scale times the caller's
    // per-channel multiplier (the row scales it folded into the last layer's weights and bias)
    const int lastl = a.n_layers - 1;
    const int mpl = ((a.M[lastl] + 31) >> 5) << 5;
    float om = 1.f;
    if (AR == 1) {
#pragma unroll
      for (int l = 0; l < S3_MAX_LAYERS; ++l) om = l == lastl ? sc.next_mul[l] : om;      // (constant indices: registers)
    }
    for (int i = tid; i < mpl; i += S3_THREADS) s_bias[a.bias_all + i] = a.out_row_mul ? a.out_row_mul[i] * om : om;
  }
  __syncthreads();          // the only s_barrier: from here on the two roles only meet through LDS words

  const int n_chunks = a.nA + a.nB + (a.tail_w > 0 ? 1 : 0);
  if (wave >= S3_NWC) {
    // ------------------------------------------------ loader wave j: slot j, chunks j, j + 4, ...
    const int j = wave - S3_NWC;
    char* slot = ring + (size_t)j * s3_chunk(AR);
    unsigned uses = 0;                   // how often the slot has been filled
    // (not for the one-tile-per-wave signature 111, whose 124 registers let two workgroups share a CU)
    if constexpr (IS_SA && !(N0 == 1 && N1 == 1 && N2 == 1)) {
      // set abstraction: two chunks of this wave's sequence in flight at any time (see SaChunk)
      struct Item { int q; unsigned nblk; unsigned base; int lc; int bi; int bx; int ok; };   // (int ok: no padding bytes for struct copies to drag through scratch)
      auto settle = [&](Item& it) {        // first chunk of this wave at or after block it.q
        for (; it.q < a.n_blocks; it.q += gridDim.x, it.base += n_chunks, ++it.nblk) {
          const int lc = (int)((j - it.base) & (S3_RING - 1));     // first local chunk with (base + lc) % 4 == j
          if (lc < n_chunks) { it.lc = lc; s3_block_map(a, it.q, it.bi, it.bx); it.ok = 1; return; }
        }
        it.ok = 0;
      };
      auto advance = [&](Item& it) {
        it.lc += S3_RING;
        if (it.lc < n_chunks) return;
        it.q += gridDim.x; it.base += n_chunks; ++it.nblk;
        settle(it);
      };
      auto ids = [&](const Item& it, int (&id)[8]) {           // the gather indices of this lane's eight columns
        const int cq = lane >> 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int gc = min(it.bx * S3_COLS + cq + 8 * i, a.cols_total - 1);
          id[i] = S3_DBG(a, 2) ? gc % a.rowsA : a.idx[(size_t)it.bi * a.cols_total + gc];
        }
      };
      Item it0 = {(int)blockIdx.x, 0u, 0u, 0, 0, 0, 0}, it1;
      int id0[8], id1[8];
      int idq0 = -1, idq1 = -1;            // the block whose indices id0 / id1 hold
      SaChunk dat0, dat1;
      settle(it0);
      it1 = it0;
      if (it1.ok) advance(it1);
      if (it0.ok) { ids(it0, id0); idq0 = it0.q; sa_fetch(a, id0, dat0, it0.bi, it0.bx * S3_COLS, it0.lc, lane); }
      if (it1.ok) { ids(it1, id1); idq1 = it1.q; sa_fetch(a, id1, dat1, it1.bi, it1.bx * S3_COLS, it1.lc, lane); }
      // store `me`'s chunk, then give its register set the chunk after `other`'s (which is still in flight)
      auto step = [&](Item& me, const Item& other, int (&idm)[8], int& idqm, SaChunk& dm) {
        if (!me.ok) return;
        // ring overlaid on P: a block's chunks may only be written once the MFMA waves have left the previous block
        if (a.ring_off == 0 && me.nblk > 0) lds_wait_ge(&ctl.blk, me.nblk);
        lds_wait_ge(&ctl.fin[j], S3_NWC * uses);          // every MFMA wave is done with the slot's previous chunk
        sa_store<AR>(a, dm, slot, me.bx * S3_COLS, me.lc, lane, sc.s_in[0]);
        ++uses;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&ctl.rdy[j], me.base + me.lc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        Item nx = other;
        if (nx.ok) advance(nx);
        me = nx;
        if (nx.ok) {
          if (idqm != nx.q) { ids(nx, idm); idqm = nx.q; }
          sa_fetch(a, idm, dm, nx.bi, nx.bx * S3_COLS, nx.lc, lane);
        }
      };
      while (it0.ok || it1.ok) {           // (the two sets alternate: it0 is the earlier chunk at the top of a pass)
        step(it0, it1, id0, idq0, dat0);
        step(it1, it0, id1, idq1, dat1);
      }
      return;
    }
    unsigned base = 0;                   // chunk number of the current block's first chunk
    unsigned blocks_done = 0;
    for (int q = blockIdx.x; q < a.n_blocks; q += gridDim.x, base += n_chunks, ++blocks_done) {
      int bi, bx;
      s3_block_map(a, q, bi, bx);
      LoaderCols<IS_SA> lcx;
      loader_cols<IS_SA>(a, lcx, bi, bx * S3_COLS, lane);      // (in flight while the wave waits for its slot)
      // ring overlaid on P: this block's chunks may only be written once the MFMA waves have left the previous block
      if (a.ring_off == 0 && blocks_done > 0) lds_wait_ge(&ctl.blk, blocks_done);
      // first local chunk with (base + lc) % 4 == j
      for (int lc = (int)((j - base) & (S3_RING - 1)); lc < n_chunks; lc += S3_RING) {
        [[maybe_unused]] const int pl = j == 0 ? 64 + 3 * (int)uses : 1 << 20;
        S3_STAMP(pl);
        lds_wait_ge(&ctl.fin[j], S3_NWC * uses);          // every MFMA wave is done with the slot's previous chunk
        S3_STAMP(pl + 1);
        loader_chunk<IS_SA, AR>(a, lcx, slot, bi, bx * S3_COLS, lc, lane, sc.s_in[0]);
        S3_STAMP(pl + 2);
        ++uses;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&ctl.rdy[j], base + lc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
    return;
  }
  // ---------------------------------------------------- MFMA waves
  S3Consumer<IS_SA, N0, N1, N2, PASSES, AR, TRL> c{a, P, ring, s_bias, s_bias + a.bias_all, &ctl, sc, 0.f, lane, wave, 0u, 0u, 0};
  for (int q = blockIdx.x; q < a.n_blocks; q += gridDim.x) {
    int bi, bx;
    s3_block_map(a, q, bi, bx);
    c.run_block(bi, bx * S3_COLS);
    if (a.ring_off == 0) {
      // the epilogue patches (P) are read by their own wave only; when every wave is through, the loaders may overwrite
      cbar(&ctl, c.phase, lane);
      if (wave == 0 && lane == 0) __hip_atomic_fetch_add(&ctl.blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  if (a.out_absmax) {      // the consumer's bound on this table: one (conditional) atomic per MFMA wave of the launch
    float m = c.amax;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if (lane == 0 && m > 0.f && __float_as_uint(m) > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(a.out_absmax, __float_as_uint(m));
  }
}


// ======================================================================================================================
// Narrow set-abstraction chains (every layer <= 128 channels: SA levels 0-1 of the backbone), fp16 x 2 arithmetic.
//
// The 4 + 4 wave kernel above spends a narrow chain's column block on everything but matrix work (profiles/NOTES.md,
// round 5: SA level 1, 16.5 k cycles per 64 columns for 3.3 k of MFMAs -- weight fragments arrive from L2 with 0.6 k
// cycles of cover against a 1 k round trip, five MFMA-wave barriers, two activation round trips through LDS).  Here
//   * the WEIGHTS of all three layers sit in LDS for the lifetime of the (persistent) workgroup -- <= 100 KB for the
//     shapes taken -- so an A fragment is an LDS read away;
//   * a WAVE owns 32 columns (one centre of 32 samples, or two of 16) through the whole chain and never meets another
//     wave: no barrier, no control words;
//   * activations never leave registers: the D layout of v_mfma_f32_32x32x16_f16 (lane = column, registers = rows
//     8 j + 4 half + i) IS a B fragment of the next layer once the next layer's K index is permuted inside every 16-k
//     slab (k' = 8 (p >> 2) + 4 half + (p & 3) for fragment position p) -- a permutation applied to the weights while
//     they are copied into LDS;
//   * the last layer is computed TRANSPOSED (activations as the A operand, weights as B: the same register contents,
//     swapped in the instruction): lane = output channel, registers = the tile's 32 columns, so the max over nsample
//     is 15 v_max in registers and one cross-half exchange instead of 80 DPP steps per row tile;
//   * the layer-0 B fragment of lane (column c, half h) is 8 consecutive channels of the gathered row, loaded straight
//     from the point-major feature table (two 16-byte loads per 16-k slab), scaled and split in registers; the next
//     tile's rows are requested as soon as layer 0 has consumed the current ones.
// Arithmetic, scales, rounding and the reference semantics are those of the AR = 1 kernels above (same S3Args, same
// host entry): results differ from them only in the summation order inside a slab.
constexpr int NW_WAVES = 8;
constexpr int NW_THREADS = 64 * NW_WAVES;

// 4-byte aligned groups of floats read with ONE load instruction (global memory takes unaligned wide accesses; a
// divergent gather costs the texture path one pass per instruction and lane, whatever the width)
struct __attribute__((packed, aligned(4))) NwF3 { float v[3]; };
struct __attribute__((packed, aligned(4))) NwF6 { float v[6]; };

template <bool VEC, int SF>
struct NwRaw {
  float4 f[VEC ? SF : 1][2];     // VEC: feature slabs (8 channels per lane and slab); else: the lane's 8 values of the only slab
  float p[3], c[3];              // neighbour / centre coordinates (grouped_xyz -= new_xyz happens on arrival)
};

__device__ __forceinline__ void nw_split8(const float (&x)[8], float mul, uint4& bh, uint4& bl) {
  const float y0[4] = {x[0] * mul, x[1] * mul, x[2] * mul, x[3] * mul};
  const float y1[4] = {x[4] * mul, x[5] * mul, x[6] * mul, x[7] * mul};
  uint2 h0, l0, h1, l1;
  split4h(y0, h0, l0);
  split4h(y1, h1, l1);
  bh = make_uint4(h0.x, h0.y, h1.x, h1.y);
  bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
}
// w.x in three partial products, smallest first; TR: the transposed product (activations as the A operand)
template <bool TR>
__device__ __forceinline__ void nw_mm(f32x16& acc, const uint4& wh, const uint4& wl, const uint4& xh, const uint4& xl) {
  const f16x8 WH = __builtin_bit_cast(f16x8, wh), WL = __builtin_bit_cast(f16x8, wl);
  const f16x8 XH = __builtin_bit_cast(f16x8, xh), XL = __builtin_bit_cast(f16x8, xl);
  if (TR) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(XH, WL, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(XL, WH, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(XH, WH, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(WL, XH, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH, XL, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(WH, XH, acc, 0, 0, 0);
  }
}
// relu(acc) * mul of one row tile -> the B fragments (two pieces) of the next layer's slabs 2 t and 2 t + 1
__device__ __forceinline__ void nw_next_frags(const f32x16& acc, float mul, uint4 (&b0)[2], uint4 (&b1)[2]) {
  float x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = s3_relu(acc[k]);
  nw_split8(x, mul, b0[0], b0[1]);
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = s3_relu(acc[8 + k]);
  nw_split8(x, mul, b1[0], b1[1]);
}

// A layer's weight fragments sit in LDS in the order they are used (slab-major, row tile inside: fragment i = s T + t
// at i * 2048 bytes), so a layer is one stream of NF fragments; NW_PF of them are requested ahead of the MFMAs that
// use them (the compiler's own schedule was load -> wait -> three MFMAs: ~100 exposed LDS round trips per tile)
constexpr int NW_PF = 2;
struct NwFrag { uint4 h, l; };
__device__ __forceinline__ NwFrag nw_ld(const char* w, int i) {
  NwFrag f;
  f.h = *reinterpret_cast<const uint4*>(w + i * 2048);
  f.l = *reinterpret_cast<const uint4*>(w + i * 2048 + 1024);
  return f;
}
template <int NF>
__device__ __forceinline__ void nw_prime(NwFrag (&q)[NW_PF + 1], const char* w) {
#pragma unroll
  for (int i = 0; i < NW_PF; ++i)
    if (i < NF) q[i] = nw_ld(w, i);
}

// XCD-aware tile walk of the narrow-chain kernels.  Workgroups are dealt round-robin to the 8 XCDs in linear-id order and
// every XCD has its own L2: with the plain walk (tile = global wave id, += waves of the launch) consecutive tiles of one
// frame go to all eight XCDs, and each of them fetches its own copy of that frame's tables from HBM (r05 counters: 7.8 x
// the compulsory reads in the FP kernel, 6.3 x in SA level 0).  Here XCD x = blockIdx.x & 7 works through the frames
// x, x + 8, ... with its own workgroups only (the launch rounds the grid to a multiple of 8); the tile numbering below
// is LOCAL to that frame subset.  Any other frame count / grid keeps the plain walk (f0 = 0, fs = 1).
struct NwWalk { int lw, nlw, f0, fs, total; };
__device__ __forceinline__ NwWalk nw_walk(int wave, int tiles_per_frame, int tiles_total, int n_frames) {
  NwWalk w;
  if ((gridDim.x & 7) == 0 && (n_frames & 7) == 0) {
    w.f0 = blockIdx.x & 7;
    w.fs = 8;
    w.lw = (int)(blockIdx.x >> 3) * NW_WAVES + wave;
    w.nlw = (int)(gridDim.x >> 3) * NW_WAVES;
    w.total = tiles_per_frame * (n_frames >> 3);
  } else {
    w.f0 = 0;
    w.fs = 1;
    w.lw = blockIdx.x * NW_WAVES + wave;
    w.nlw = gridDim.x * NW_WAVES;
    w.total = tiles_total;
  }
  return w;
}

// S0, S1, S2: 16-k slabs of the three layers (S1 = ceil(M0 / 16), S2 = ceil(M1 / 16)); T2: 32-row tiles of the last
// layer; VEC: the feature table is read in 16-byte pieces (C a multiple of 16, aligned rows; S0 = C / 16 + 1, the last
// slab holds the relative coordinates) -- otherwise C = 6: the six-float rows of SA level 0 read in place (6 features +
// 3 coordinates = one slab); MINW: waves per SIMD the register budget is set for (2 = one workgroup per CU, 4 = two)
template <int S0, int S1, int S2, int T2, bool VEC, int MINW>
__global__ __launch_bounds__(NW_THREADS, MINW) void sa_chain_narrow_kernel(S3Args a, int tiles_per_frame, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  __shared__ float s_amax[NW_WAVES];
  constexpr int T0 = (S1 + 1) / 2, T1 = (S2 + 1) / 2;
  constexpr int SF = VEC ? S0 - 1 : 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  constexpr int s1 = S1, s2 = S2;
  constexpr int w1_off = S0 * T0 * 2048, w2_off = w1_off + S1 * T1 * 2048;
  constexpr int bias_off = w2_off + S2 * T2 * 2048;
  constexpr int om_off = bias_off + (T0 + T1 + T2) * 32 * 4;         // the last layer's per-row output multipliers
  const S3Scales sc = s3_scales(a);
  // ---- weights and biases -> LDS, once
  {
    const uint4* src = a.W[0];
    uint4* dst = reinterpret_cast<uint4*>(s_mem);
    for (int i = tid; i < S0 * T0 * 128; i += NW_THREADS) dst[i] = src[i];
  }
#pragma unroll
  for (int l = 1; l < 3; ++l) {
    // K permuted inside every slab: fragment position p of half h holds k = 8 (p >> 2) + 4 h + (p & 3), i.e. the low
    // 8 bytes of lane (m, h) come from bytes [8 h, 8 h + 8) of the packed lane (m, 0), the high 8 bytes from lane (m, 1)
    const char* src = reinterpret_cast<const char*>(a.W[l]);
    char* dst = s_mem + (l == 1 ? w1_off : w2_off);
    const int units = (l == 1 ? s1 * T1 : s2 * T2) * 256;
    for (int u = tid; u < units; u += NW_THREADS) {
      const int f = u >> 8, pc = (u >> 7) & 1, ln = (u >> 1) & 63, q = u & 1;
      const int m = ln & 31, h = ln >> 5;
      *reinterpret_cast<uint2*>(dst + f * 2048 + pc * 1024 + ln * 16 + 8 * q) =
          *reinterpret_cast<const uint2*>(src + f * 2048 + pc * 1024 + (m + 32 * q) * 16 + 8 * h);
    }
  }
  float* s_bias = reinterpret_cast<float*>(s_mem + bias_off);
  {
    int off = 0;
#pragma unroll
    for (int l = 0; l < 3; ++l) {
      const int mp = (l == 0 ? T0 : l == 1 ? T1 : T2) * 32;
      const float bm = sc.bias_mul[l];
      for (int i = tid; i < mp; i += NW_THREADS) s_bias[off + i] = a.bias[l][i] * bm;
      off += mp;
    }
    // accumulator -> output, per row: the chain's (power-of-two) scale times the caller's per-channel multiplier
    float* s_omw = reinterpret_cast<float*>(s_mem + om_off);
    for (int i = tid; i < T2 * 32; i += NW_THREADS) s_omw[i] = a.out_row_mul ? a.out_row_mul[i] * sc.next_mul[2] : sc.next_mul[2];
  }
  __syncthreads();
  const char* w0 = s_mem + lane * 16;
  const char* w1 = s_mem + w1_off + lane * 16;
  const char* w2 = s_mem + w2_off + lane * 16;
  const float* sb0 = s_bias;
  const float* sb1 = s_bias + T0 * 32;
  const float* sb2 = s_bias + (T0 + T1) * 32;
  const int ns = a.ns;
  const float s0 = sc.s_in[0];
  const float* s_om = reinterpret_cast<const float*>(s_mem + om_off);
  float amax = 0.f;

  // this lane's column of tile t: neighbour index, then the row segments of the gather
  // (frame bi, tile tin inside the frame)
  auto tile_id = [&](int bi, int tin, int& gc) -> int {
    gc = min(tin * 32 + col, a.cols_total - 1);
    return a.idx[(size_t)bi * a.cols_total + gc];
  };
  auto gather = [&](NwRaw<VEC, SF>& r, int bi, int gc, int id) {
    const float* row = a.tabA + ((size_t)bi * a.rowsA + id) * a.ldA;
    if (VEC) {
#pragma unroll
      for (int s = 0; s < SF; ++s) {
        r.f[s][0] = *reinterpret_cast<const float4*>(row + 16 * s + 8 * half);
        r.f[s][1] = *reinterpret_cast<const float4*>(row + 16 * s + 8 * half + 4);
      }
    } else {
      // the backbone's first level: six features per point (C = 6, the only element-wise shape instantiated); both
      // halves of the wave request the row, the upper half only uses fragment positions >= 8
      const NwF6 f = *reinterpret_cast<const NwF6*>(row);
      r.f[0][0] = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
      r.f[0][1] = make_float4(f.v[4], f.v[5], 0.f, 0.f);
    }
    const NwF3 p = *reinterpret_cast<const NwF3*>(a.xyz + ((size_t)bi * a.n + id) * 3);
    const NwF3 c = *reinterpret_cast<const NwF3*>(a.new_xyz + ((size_t)bi * a.m + gc / ns) * 3);
    r.p[0] = p.v[0]; r.p[1] = p.v[1]; r.p[2] = p.v[2];
    r.c[0] = c.v[0]; r.c[1] = c.v[1]; r.c[2] = c.v[2];
  };

  const NwWalk wk = nw_walk(wave, tiles_per_frame, tiles_total, a.n_frames);
  const int nw = wk.nlw;
  tiles_total = wk.total;                                                    // (local to this XCD's frames from here on)
  NwRaw<VEC, SF> raw;
  int tile = wk.lw;
  int bi, tin;
  {
    const int lf = tile / tiles_per_frame;                                   // the only division: the walk below steps
    tin = tile - lf * tiles_per_frame;
    bi = wk.f0 + wk.fs * lf;
  }
  if (tile < tiles_total) {
    int gc;
    const int id = tile_id(bi, tin, gc);
    gather(raw, bi, gc, id);
  }
  for (; tile < tiles_total; tile += nw) {
    const int tn = tile + nw;
    int bi_n = bi, tin_n = tin + nw;
    while (tin_n >= tiles_per_frame) { tin_n -= tiles_per_frame; bi_n += wk.fs; }
    int gc_n = 0, id_n = 0;
    if (tn < tiles_total) id_n = tile_id(bi_n, tin_n, gc_n);           // (arrives under layer 0)
    const int tcol = tin * 32;

    // ---- layer 0: B fragments from the gathered rows
    f32x16 acc0[T0];
    NwFrag q[NW_PF + 1];
    nw_prime<S0 * T0>(q, w0);
#pragma unroll
    for (int t = 0; t < T0; ++t) acc_bias(acc0[t], sb0 + 32 * t, half);
    const float rel[3] = {raw.p[0] - raw.c[0], raw.p[1] - raw.c[1], raw.p[2] - raw.c[2]};      // grouped_xyz -= new_xyz
#pragma unroll
    for (int s = 0; s < S0; ++s) {
      float x[8];
      if (VEC && s < SF) {
        x[0] = raw.f[s][0].x; x[1] = raw.f[s][0].y; x[2] = raw.f[s][0].z; x[3] = raw.f[s][0].w;
        x[4] = raw.f[s][1].x; x[5] = raw.f[s][1].y; x[6] = raw.f[s][1].z; x[7] = raw.f[s][1].w;
      } else if (VEC) {
        // the coordinate slab: k = C + 0..2 in the low half's first positions
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = 0.f;
        if (half == 0) { x[0] = rel[0]; x[1] = rel[1]; x[2] = rel[2]; }
      } else {
        // k = 0..5 features, 6..8 relative coordinates: positions 0..7 in the low half, position 8 opens the high half
        x[0] = half ? rel[2] : raw.f[0][0].x;
        x[1] = half ? 0.f : raw.f[0][0].y; x[2] = half ? 0.f : raw.f[0][0].z; x[3] = half ? 0.f : raw.f[0][0].w;
        x[4] = half ? 0.f : raw.f[0][1].x; x[5] = half ? 0.f : raw.f[0][1].y;
        x[6] = half ? 0.f : rel[0]; x[7] = half ? 0.f : rel[1];
      }
      uint4 bh, bl;
      nw_split8(x, s0, bh, bl);
#pragma unroll
      for (int t = 0; t < T0; ++t) {
        constexpr int NF = S0 * T0;
        const int i = s * T0 + t;
        if (i + NW_PF < NF) q[(i + NW_PF) % (NW_PF + 1)] = nw_ld(w0, i + NW_PF);
        nw_mm<false>(acc0[t], q[i % (NW_PF + 1)].h, q[i % (NW_PF + 1)].l, bh, bl);
        __builtin_amdgcn_sched_barrier(0);      // keeps the request NW_PF fragments ahead (the scheduler sinks it otherwise)
      }
    }
    // the next tile's rows: the registers are free now, the loads land under layers 1-2
    __builtin_amdgcn_sched_barrier(0);
    if (tn < tiles_total) gather(raw, bi_n, gc_n, id_n);
    __builtin_amdgcn_sched_barrier(0);

    // ---- layer 1 (its first fragments travel under the activation split)
    nw_prime<S1 * T1>(q, w1);
    uint4 b1[2 * T0][2];
#pragma unroll
    for (int t = 0; t < T0; ++t) nw_next_frags(acc0[t], sc.next_mul[0], b1[2 * t], b1[2 * t + 1]);
    f32x16 acc1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t) acc_bias(acc1[t], sb1 + 32 * t, half);
#pragma unroll
    for (int s = 0; s < S1; ++s) {
#pragma unroll
      for (int t = 0; t < T1; ++t) {
        constexpr int NF = S1 * T1;
        const int i = s * T1 + t;
        if (i + NW_PF < NF) q[(i + NW_PF) % (NW_PF + 1)] = nw_ld(w1, i + NW_PF);
        nw_mm<false>(acc1[t], q[i % (NW_PF + 1)].h, q[i % (NW_PF + 1)].l, b1[s][0], b1[s][1]);
        __builtin_amdgcn_sched_barrier(0);      // keeps the request NW_PF fragments ahead (the scheduler sinks it otherwise)
      }
    }
    // ---- layer 2, transposed: lane = output channel 32 t + col, registers = columns 8 j + 4 half + i of the tile
    nw_prime<S2 * T2>(q, w2);
    uint4 b2[2 * T1][2];
#pragma unroll
    for (int t = 0; t < T1; ++t) nw_next_frags(acc1[t], sc.next_mul[1], b2[2 * t], b2[2 * t + 1]);
    // (accumulators start at zero -- free, an inline constant of the first MFMA -- and the bias, one value per lane
    // here, is added to the results: broadcasting it into 16 registers per tile cost 64 moves)
    f32x16 acc2[T2];
#pragma unroll
    for (int t = 0; t < T2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < S2; ++s) {
#pragma unroll
      for (int t = 0; t < T2; ++t) {
        constexpr int NF = S2 * T2;
        const int i = s * T2 + t;
        if (i + NW_PF < NF) q[(i + NW_PF) % (NW_PF + 1)] = nw_ld(w2, i + NW_PF);
        nw_mm<true>(acc2[t], q[i % (NW_PF + 1)].h, q[i % (NW_PF + 1)].l, b2[s][0], b2[s][1]);
        __builtin_amdgcn_sched_barrier(0);      // keeps the request NW_PF fragments ahead (the scheduler sinks it otherwise)
      }
    }
    // ---- relu, max over the centre's columns, store
    const int M = a.M[2];
#pragma unroll
    for (int t = 0; t < T2; ++t) {
      // max_c relu(y_c + b) on the bit patterns: signed-integer max orders the non-negative floats correctly and ranks
      // every negative one below them, and the 0 in the chain is the relu (v_max3_i32: 8 per half tile pair)
      const float b = sb2[32 * t + col];
      const float om = s_om[32 * t + col];                         // this lane's output channel
      int k[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) k[r] = __float_as_int(acc2[t][r] + b);
      int ilo = max(max(max(k[0], k[1]), max(k[2], k[3])), max(max(k[4], k[5]), max(k[6], k[7])));       // columns 0..15 (j = 0, 1)
      int ihi = max(max(max(k[8], k[9]), max(k[10], k[11])), max(max(k[12], k[13]), max(k[14], k[15])));  // columns 16..31
      float lo = __int_as_float(max(ilo, 0));
      float hi = __int_as_float(max(ihi, 0));
      const int row = 32 * t + col;
      if (ns == 32) {
        float v = __int_as_float(max(__float_as_int(lo), __float_as_int(hi)));
        v = __int_as_float(max(__float_as_int(v), __float_as_int(__shfl_xor(v, 32, 64)))) * om;
        const int centre = tcol >> 5;
        if (half == 0 && row < M && centre < a.m) {
          a.out[((size_t)bi * a.m + centre) * a.ld_out + a.coff + row] = v;
          amax = fmaxf(amax, v);
        }
      } else {
        lo = __int_as_float(max(__float_as_int(lo), __float_as_int(__shfl_xor(lo, 32, 64))));
        hi = __int_as_float(max(__float_as_int(hi), __float_as_int(__shfl_xor(hi, 32, 64))));
        const int centre = (tcol >> 4) + half;
        const float v = (half ? hi : lo) * om;
        if (row < M && centre < a.m) {
          a.out[((size_t)bi * a.m + centre) * a.ld_out + a.coff + row] = v;
          amax = fmaxf(amax, v);
        }
      }
    }
    bi = bi_n;
    tin = tin_n;
  }
  if (a.out_absmax) {            // one conditional atomic per workgroup
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < NW_WAVES; ++w) m = fmaxf(m, s_amax[w]);
      if (m > 0.f && __float_as_uint(m) > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.out_absmax, __float_as_uint(m));
    }
  }
}

// ---- feature propagation, pre-contracted two-layer chain (FP level 0 of the backbone) on the same plan ---------------------
// y0 = relu(interp(z) + Wb.skip + b0), y1 = relu(W1.y0 + b1): z (B, m, M0) is the first conv's interpolated half, already
// applied per KNOWN point (_ext.fp_interp_mlp's pre-contraction).  The 4 + 4 wave kernel receives this chain with an
// identity block in front of Wb and multiplies the interpolated rows by it -- M0 / 16 slabs of MFMAs that copy.  Here a
// wave owns 32 unknown points: the three neighbours' rows of z are read in the ACCUMULATOR layout (lane = column,
// registers = rows 8 j + 4 half + i: 16-byte pieces of a row), weighted, scaled and added to the bias -- that is layer
// 0's accumulator before the skip slab's MFMAs; layer 1 and the weights in LDS as in the set-abstraction kernel; the
// result leaves in the channel-major (B, M1, n) layout of the reference, 128 contiguous bytes per register and half-wave.
struct __attribute__((packed, aligned(4))) NwI3 { int v[3]; };

template <int T, int MINW>
__global__ __launch_bounds__(NW_THREADS, MINW) void fp_chain_narrow_kernel(S3Args a, int tiles_per_frame, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char s_mem[];
  __shared__ float s_amax[NW_WAVES];
  constexpr int S1 = 2 * T;                          // layer 1 contracts over M0 = 32 T channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;
  constexpr int w1_off = T * 2048, bias_off = w1_off + S1 * T * 2048;
  constexpr int om_off = bias_off + 2 * 32 * T * 4;                  // the last layer's per-row output multipliers
  const S3Scales sc = s3_scales(a);
  {   // the skip slab of layer 0 (the slab behind the identity block), layer 1 with its K permuted, biases in the accumulators' scale
    const uint4* src = a.W[0] + (size_t)(S1 * T) * 128;
    uint4* dst = reinterpret_cast<uint4*>(s_mem);
    for (int i = tid; i < T * 128; i += NW_THREADS) dst[i] = src[i];
    const char* src1 = reinterpret_cast<const char*>(a.W[1]);
    char* dst1 = s_mem + w1_off;
    for (int u = tid; u < S1 * T * 256; u += NW_THREADS) {
      const int f = u >> 8, pc = (u >> 7) & 1, ln = (u >> 1) & 63, q = u & 1;
      const int m = ln & 31, h = ln >> 5;
      *reinterpret_cast<uint2*>(dst1 + f * 2048 + pc * 1024 + ln * 16 + 8 * q) =
          *reinterpret_cast<const uint2*>(src1 + f * 2048 + pc * 1024 + (m + 32 * q) * 16 + 8 * h);
    }
    float* sb = reinterpret_cast<float*>(s_mem + bias_off);
    for (int i = tid; i < 32 * T; i += NW_THREADS) {
      sb[i] = a.bias[0][i] * sc.bias_mul[0];
      sb[32 * T + i] = a.bias[1][i] * sc.bias_mul[1];
      sb[64 * T + i] = a.out_row_mul ? a.out_row_mul[i] * sc.next_mul[1] : sc.next_mul[1];
    }
  }
  __syncthreads();
  const char* w0 = s_mem + lane * 16;
  const char* w1 = s_mem + w1_off + lane * 16;
  const float* sb0 = reinterpret_cast<const float*>(s_mem + bias_off);
  const float* sb1 = sb0 + 32 * T;
  const float bm0 = sc.bias_mul[0], s0 = sc.s_in[0];
  const float* s_om = reinterpret_cast<const float*>(s_mem + om_off);
  const int n = a.cols_total, M1 = a.M[1];
  float amax = 0.f;

  const NwWalk wk = nw_walk(wave, tiles_per_frame, tiles_total, a.n_frames);
  const int nw = wk.nlw;
  tiles_total = wk.total;
  int tile = wk.lw;
  int bi, tin;
  {
    const int lf = tile / tiles_per_frame;
    tin = tile - lf * tiles_per_frame;
    bi = wk.f0 + wk.fs * lf;
  }
  for (; tile < tiles_total; tile += nw) {
    const int g = tin * 32 + col, gc = min(g, n - 1);
    const size_t o3 = ((size_t)bi * n + gc) * 3;
    const NwI3 id = *reinterpret_cast<const NwI3*>(a.idx + o3);
    const NwF3 wt = *reinterpret_cast<const NwF3*>(a.weight + o3);
    const NwF6 sk = *reinterpret_cast<const NwF6*>(a.tabB + ((size_t)bi * a.rowsB + gc) * a.ldB);
    const float* z0 = a.tabA + ((size_t)bi * a.rowsA + id.v[0]) * a.ldA + 4 * half;
    const float* z1 = a.tabA + ((size_t)bi * a.rowsA + id.v[1]) * a.ldA + 4 * half;
    const float* z2 = a.tabA + ((size_t)bi * a.rowsA + id.v[2]) * a.ldA + 4 * half;
    // ---- layer 0's accumulators: (three_interpolate(z) (pointnet2_utils.py:136-170: p0 w0 + p1 w1 + p2 w2, unfused, in
    // this order) in the accumulators' scale) + bias
    f32x16 acc0[T];
    NwFrag q[NW_PF + 1];
    nw_prime<T>(q, w0);
    // (rows of tile t + 1 are requested before tile t's are used; without the fences the compiler requests all four
    // tiles at once -- 192 registers -- and spills)
    float4 P[2][3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      P[0][0][j] = *reinterpret_cast<const float4*>(z0 + 8 * j);
      P[0][1][j] = *reinterpret_cast<const float4*>(z1 + 8 * j);
      P[0][2][j] = *reinterpret_cast<const float4*>(z2 + 8 * j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + 1 < T) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          P[(t + 1) & 1][0][j] = *reinterpret_cast<const float4*>(z0 + 32 * (t + 1) + 8 * j);
          P[(t + 1) & 1][1][j] = *reinterpret_cast<const float4*>(z1 + 32 * (t + 1) + 8 * j);
          P[(t + 1) & 1][2][j] = *reinterpret_cast<const float4*>(z2 + 32 * (t + 1) + 8 * j);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 bb = *reinterpret_cast<const float4*>(sb0 + 32 * t + 8 * j + 4 * half);
        const float4 a0 = P[t & 1][0][j], a1 = P[t & 1][1][j], a2 = P[t & 1][2][j];
        acc0[t][4 * j + 0] = (a0.x * wt.v[0] + a1.x * wt.v[1] + a2.x * wt.v[2]) * bm0 + bb.x;
        acc0[t][4 * j + 1] = (a0.y * wt.v[0] + a1.y * wt.v[1] + a2.y * wt.v[2]) * bm0 + bb.y;
        acc0[t][4 * j + 2] = (a0.z * wt.v[0] + a1.z * wt.v[1] + a2.z * wt.v[2]) * bm0 + bb.z;
        acc0[t][4 * j + 3] = (a0.w * wt.v[0] + a1.w * wt.v[1] + a2.w * wt.v[2]) * bm0 + bb.w;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the skip slab: k = M0 + 0..5 in the low half's positions 0..5
    {
      float x[8];
#pragma unroll
      for (int k = 0; k < 6; ++k) x[k] = half ? 0.f : sk.v[k];
      x[6] = 0.f; x[7] = 0.f;
      uint4 bh, bl;
      nw_split8(x, s0, bh, bl);
#pragma unroll
      for (int t = 0; t < T; ++t) {
        if (t + NW_PF < T) q[(t + NW_PF) % (NW_PF + 1)] = nw_ld(w0, t + NW_PF);
        nw_mm<false>(acc0[t], q[t % (NW_PF + 1)].h, q[t % (NW_PF + 1)].l, bh, bl);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- layer 1
    nw_prime<S1 * T>(q, w1);
    uint4 b1[S1][2];
#pragma unroll
    for (int t = 0; t < T; ++t) nw_next_frags(acc0[t], sc.next_mul[0], b1[2 * t], b1[2 * t + 1]);
    __builtin_amdgcn_sched_barrier(0);               // (layer 0's accumulators are dead before layer 1's are born)
    f32x16 acc1[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc_bias(acc1[t], sb1 + 32 * t, half);
#pragma unroll
    for (int s = 0; s < S1; ++s) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        constexpr int NF = S1 * T;
        const int i = s * T + t;
        if (i + NW_PF < NF) q[(i + NW_PF) % (NW_PF + 1)] = nw_ld(w1, i + NW_PF);
        nw_mm<false>(acc1[t], q[i % (NW_PF + 1)].h, q[i % (NW_PF + 1)].l, b1[s][0], b1[s][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- relu, store: register 4 j + i of tile t = channel 32 t + 8 j + 4 half + i of this lane's point (M1 = 32 T).
    // One wave-uniform row pointer walks the channels (a running scalar: 64 precomputed row offsets per lane were
    // hoisted out of the tile loop and spilled), the lane adds its point and its half's four rows once
    if (g < n) {
      const unsigned voff = ((unsigned)g + 4u * (unsigned)half * (unsigned)n) * 4u;
      const char* rowp = reinterpret_cast<const char*>(a.out + (size_t)bi * M1 * n);
      const size_t step = (size_t)n * 4;
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 om4 = *reinterpret_cast<const float4*>(s_om + 32 * t + 8 * j + 4 * half);    // per output channel
          const float omk[4] = {om4.x, om4.y, om4.z, om4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float v = s3_relu(acc1[t][4 * j + i]) * omk[i];
            *reinterpret_cast<float*>(const_cast<char*>(rowp) + voff) = v;
            amax = fmaxf(amax, v);
            rowp += step;
          }
          rowp += 4 * step;
          asm volatile("" : "+s"(rowp));      // keep it a running pointer
        }
    }
    tin += nw;
    while (tin >= tiles_per_frame) { tin -= tiles_per_frame; bi += wk.fs; }
  }
  if (a.out_absmax) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0) s_amax[wave] = amax;
    __syncthreads();
    if (tid == 0) {
      float m = 0.f;
#pragma unroll
      for (int w = 0; w < NW_WAVES; ++w) m = fmaxf(m, s_amax[w]);
      if (m > 0.f && __float_as_uint(m) > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(a.out_absmax, __float_as_uint(m));
    }
  }
}

// 1..4: instance that takes the chain; -1: none.  c = feature channels (dims[0] = c + 3); no_narrow: the caller's per-call
// PVN3D_MLP_NO_NARROW flag (A/B measurements -- there is no process-wide switch: the library keeps no mutable state)
int nw_signature(int c, int nsample, int n_layers, const int* dims, int no_narrow) {
  if (no_narrow || n_layers != 3 || (nsample != 16 && nsample != 32) || c < 0 || dims[0] != c + 3) return -1;
  for (int l = 1; l <= 3; ++l)
    if (dims[l] <= 0 || dims[l] > 128) return -1;
  const int s1 = (dims[1] + 15) / 16, s2 = (dims[2] + 15) / 16, t2 = (dims[3] + 31) / 32;
  const bool small_k = c == 6;                      // one slab, the 6-float row read in place (any row stride)
  const bool vec_k = c == 96;                       // S0 = 7 (the instantiated slab count)
  if (small_k && s1 == 1 && s2 == 1 && t2 == 1) return 1;       // 9 -> 16 -> 16 -> 32
  if (small_k && s1 == 2 && s2 == 2 && t2 == 2) return 2;       // 9 -> 32 -> 32 -> 64
  if (vec_k && s1 == 4 && s2 == 4 && t2 == 4) return 3;         // 99 -> 64 -> 64 -> 128
  if (vec_k && s1 == 4 && s2 == 6 && t2 == 4) return 4;         // 99 -> 64 -> 96 -> 128
  return -1;
}

int nw_launch(S3Args& a, int sig, hipStream_t st) {
  const int t[3] = {(a.M[0] + 31) / 32, (a.M[1] + 31) / 32, (a.M[2] + 31) / 32};
  const int s0 = (a.K[0] + 15) / 16, s1 = (a.M[0] + 15) / 16, s2 = (a.M[1] + 15) / 16;
  const size_t lds = (size_t)(s0 * t[0] + s1 * t[1] + s2 * t[2]) * 2048 + (size_t)(t[0] + t[1] + 2 * t[2]) * 32 * 4;
  if (lds > 150 * 1024) return -1;
  const int tpf = pvn3d_ceil_div(a.cols_total, 32);
  const long long total = (long long)tpf * a.n_frames;
  if (total > 0x7fffffff) return -1;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
#define NW_GO(S0, S1, S2, T2, VEC, MINW)                                                                               \
  do {                                                                                                                 \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(sa_chain_narrow_kernel<S0, S1, S2, T2, VEC, MINW>));           \
    int grid = (int)min((long long)cus * (MINW / 2), (total + NW_WAVES - 1) / NW_WAVES);                               \
    if (grid >= 8) grid &= ~7;            /* a multiple of 8: the kernel's frame -> XCD walk (nw_walk) */              \
    hipLaunchKernelGGL((sa_chain_narrow_kernel<S0, S1, S2, T2, VEC, MINW>), dim3(grid), dim3(NW_THREADS), lds, st, a,  \
                       tpf, (int)total);                                                                               \
  } while (0)
  if (sig == 1) NW_GO(1, 1, 1, 1, false, 4);
  else if (sig == 2) NW_GO(1, 2, 2, 2, false, 4);
  else if (sig == 3) NW_GO(7, 4, 4, 4, true, 2);
  else if (sig == 4) NW_GO(7, 4, 6, 4, true, 2);
  else return -1;
#undef NW_GO
  PVN3D_LAUNCH_CHECK();
  return 0;
}

// the pre-contracted FP chain the narrow kernel takes: M0 = M1 = 128, six skip channels, channel-major output
bool nwfp_ok(int c2, int c1, int n_layers, const int* dims, int out_point_major, int no_narrow) {
  return !no_narrow && n_layers == 2 && c2 == 128 && c1 == 6 && dims[0] == c2 + c1 && dims[1] == 128 && dims[2] == 128 &&
         !out_point_major;
}
int nwfp_launch(S3Args& a, hipStream_t st) {
  constexpr int T = 4;
  const size_t lds = (size_t)(T + 2 * T * T) * 2048 + (size_t)3 * 32 * T * 4;
  const int tpf = pvn3d_ceil_div(a.cols_total, 32);
  const long long total = (long long)tpf * a.n_frames;
  if (total > 0x7fffffff) return -1;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(fp_chain_narrow_kernel<T, 2>));
  int grid = (int)min((long long)cus, (total + NW_WAVES - 1) / NW_WAVES);
  if (grid >= 8) grid &= ~7;              // a multiple of 8: the kernel's frame -> XCD walk (nw_walk)
  hipLaunchKernelGGL((fp_chain_narrow_kernel<T, 2>), dim3(grid), dim3(NW_THREADS), lds, st, a, tpf, (int)total);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

bool vec_ok(const float* p, int ld) { return p != nullptr && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Chain signature = row tiles per MFMA wave in each layer (x rounds of the last layer); the instantiated ones are the
// shapes of PVN3D's backbone that gain: SA level 2 (259 -> 128 -> 196 -> 256), SA level 3 (515 -> 256 -> 256 | 384 ->
// 512: the 16 tiles of the last layer in two rounds), FP level 0 (262 -> 128 -> 128), FP level 1 (608 -> 256 -> 256).
// -1: no kernel for this chain (the caller keeps the fp32-MFMA kernels of sa_mlp.hip).
int s3_signature(int is_sa, int n_layers, const int* dims, int nsample, int arith = 0) {
  int n[3] = {0, 0, 0};
  if (n_layers < 2 || n_layers > 3) return -1;
  for (int l = 0; l < n_layers; ++l) n[l] = pvn3d_ceil_div(pvn3d_ceil_div(dims[l + 1], 32), S3_NWC);
  const int sig = n[0] * 100 + n[1] * 10 + n[2];
  if (is_sa) {
    if (sig == 111 && arith == 1) return sig;       // SA level 1 (99 -> 64 -> 64 | 96 -> 128): pays with fp16 x 2 only
    if (sig == 122) return sig;
    if ((sig == 224 || sig == 234) && nsample >= 16) return sig;      // DPP pooling: whole 16-lane rows per centre
    return -1;
  }
  return (sig == 110 || sig == 220) ? sig : -1;
}

// LDS plan; false when the chain does not fit
bool s3_plan(S3Args& a, int arith = 0) {
  int kcap = 32, bias_all = 0;
  for (int l = 0; l < a.n_layers; ++l) {
    const int mp = ((a.M[l] + 31) / 32) * 32;
    if (l + 1 < a.n_layers) kcap = max(kcap, mp);
    bias_all += mp;
  }
  a.rs = 2 * kcap + 16;                                    // 16 x odd (kcap is a multiple of 32)
  a.ps = S3_COLS * a.rs;
  size_t p_bytes = (size_t)s3_np(arith) * a.ps;
  if (a.is_sa) p_bytes = max(p_bytes, (size_t)S3_NWC * 32 * S3_EPAD * 4);
  const size_t ring_bytes = (size_t)S3_RING * s3_chunk(arith);
  // biases of every layer + the last layer's per-row output multipliers
  const size_t bias_bytes = (size_t)(bias_all + ((a.M[a.n_layers - 1] + 31) / 32) * 32) * 4;
  const size_t budget = 160 * 1024 - 256;
  if (p_bytes + ring_bytes + bias_bytes <= budget) {
    a.ring_off = (int)p_bytes;
    a.bias_off = (int)(p_bytes + ring_bytes);
  } else if (max(p_bytes, ring_bytes) + bias_bytes <= budget) {
    a.ring_off = 0;                                        // chunk ring overlaid on P
    a.bias_off = (int)max(p_bytes, ring_bytes);
  } else {
    return false;
  }
  a.bias_all = bias_all;
  return true;
}

int s3_launch(S3Args& a, int sig, hipStream_t st, int arith = 0) {
  if (!s3_plan(a, arith)) return -1;
  a.dbg = 0;
#ifdef PVN3D_S3_TUNING
  {
    const char* e = getenv("PVN3D_S3_DBG");
    a.dbg = e ? atoi(e) : 0;
  }
#endif
  const size_t lds = (size_t)a.bias_off + (size_t)(a.bias_all + ((a.M[a.n_layers - 1] + 31) / 32) * 32) * 4;
  a.bpf = pvn3d_ceil_div(a.cols_total, S3_COLS);
  a.n_blocks = a.bpf * a.n_frames;
  // persistent grid: one workgroup per CU (the LDS footprint allows no more), a multiple of 8 so that a workgroup's
  // blocks stay on one XCD's frames
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
  }
  // (the narrow fp16 x 2 chain -- signature 111, 122 registers per wave -- fits twice on a CU: two workgroups cover each
  // other's store / barrier / epilogue phases, which the one-workgroup shape leaves exposed)
  const int wg_per_cu = (arith == 1 && sig == 111 && 2 * (lds + 512) <= 160 * 1024) ? 2 : 1;
  int grid = min(a.n_blocks, cus * wg_per_cu);
  if (grid >= 8) grid &= ~7;
#define S3_GO2(SA, A0, A1, A2, PS, AR, TRL)                                                                            \
  do {                                                                                                                 \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(mlp_chain_s3_kernel<SA, A0, A1, A2, PS, AR, TRL>));            \
    hipLaunchKernelGGL((mlp_chain_s3_kernel<SA, A0, A1, A2, PS, AR, TRL>), dim3(grid), dim3(S3_THREADS), lds, st, a);  \
  } while (0)
#define S3_GO1(SA, A0, A1, A2, PS, AR) S3_GO2(SA, A0, A1, A2, PS, AR, false)
  // fp16 x 2 set abstraction with whole 16 / 32 / 64-column centres: the instantiation with the transposed last layer
  const bool trl = arith == 1 && a.is_sa && a.ns >= 16;
#define S3_GO(SA, A0, A1, A2, PS)                                                                                 \
  do {                                                                                                            \
    if (arith == 1 && SA && trl) S3_GO2(SA, A0, A1, A2, PS, 1, SA);                                               \
    else if (arith == 1) S3_GO1(SA, A0, A1, A2, PS, 1);                                                           \
    else S3_GO1(SA, A0, A1, A2, PS, 0);                                                                           \
  } while (0)
  if (a.is_sa && sig == 111 && arith == 1) S3_GO1(true, 1, 1, 1, 1, 1);
  else if (a.is_sa && sig == 122) S3_GO(true, 1, 2, 2, 1);
  else if (a.is_sa && sig == 224) S3_GO(true, 2, 2, 2, 2);
  else if (a.is_sa && sig == 234) S3_GO(true, 2, 3, 2, 2);
  else if (!a.is_sa && sig == 110) S3_GO(false, 1, 1, 0, 1);
  else if (!a.is_sa && sig == 220) S3_GO(false, 2, 2, 0, 1);
  else return -1;
#undef S3_GO
#undef S3_GO1
#undef S3_GO2
  PVN3D_LAUNCH_CHECK();
  return 0;
}

bool s3_fill(S3Args* a, int n_layers, const int* dims, const void* const* w, const float* const* bias) {
  if (n_layers < 1 || n_layers > S3_MAX_LAYERS) return false;
  a->n_layers = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    a->K[l] = dims[l];
    a->M[l] = dims[l + 1];
    a->W[l] = reinterpret_cast<const uint4*>(w[l]);
    a->bias[l] = bias[l];
    if (dims[l] <= 0 || dims[l + 1] <= 0 || !w[l] || !bias[l]) return false;
  }
  return true;
}

}  // namespace

#ifdef PVN3D_S3_TUNING
extern "C" int pvn3d_debug_s3_prof_read(unsigned long long* host256) {
  return (int)hipMemcpyFromSymbol(host256, HIP_SYMBOL(g_s3_prof), sizeof(unsigned long long) * 256);
}
#endif

// 1: the split-bf16 family takes this shape; 0: use pvn3d_sa_mlp_maxpool / pvn3d_fp_interp_mlp (fp32 MFMA).
// c_a: channels of the first row source (SA features / FP known points), c_b: FP skip channels.
static int s3_ok(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host, int arith, int no_narrow = 0) {
  if (!dims_host || n_layers < 2 || n_layers > 3) return 0;
  if (is_sa && arith == 1 && nw_signature(c_a, nsample, n_layers, dims_host, no_narrow) >= 0) return 1;      // narrow-chain kernel
  if (c_a <= 0 || (c_a % S3_KC) != 0) return 0;                       // whole 32-channel row-gather chunks
  if (is_sa) {
    if (nsample <= 0 || (nsample & (nsample - 1)) || nsample > 64) return 0;
  } else if ((c_b % S3_KC) > 8) {
    return 0;                                                         // the tail chunk holds <= 8 channels
  }
  if (s3_signature(is_sa, n_layers, dims_host, nsample, arith) < 0) return 0;
  S3Args a = {};
  a.n_layers = n_layers;
  a.is_sa = is_sa;
  for (int l = 0; l < n_layers; ++l) a.M[l] = dims_host[l + 1];
  return s3_plan(a, arith) ? 1 : 0;
}
extern "C" int pvn3d_mlp_split_ok(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host) {
  return s3_ok(is_sa, c_a, c_b, nsample, n_layers, dims_host, 0);
}
// the same question for the fp16 x 2 kernels (two pieces: smaller LDS footprint, one more chain signature)
extern "C" int pvn3d_mlp_split2_ok(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host, int flags) {
  return s3_ok(is_sa, c_a, c_b, nsample, n_layers, dims_host, 1, flags & PVN3D_MLP_NO_NARROW);
}

// Which kernel family pvn3d_*_split2 would run the chain on: 0 none (pvn3d_mlp_split2_ok == 0), 1 the 4 + 4-wave kernel,
// 2 a narrow-chain kernel (the chain's weights resident in LDS: 2-4 x faster on chains it takes).  The narrow kernels
// are instantiated for the backbone's widths only; a host that builds a different network can see here that a chain
// every layer of which is <= 128 wide does not get one (lib/pointnet2_utils/_ext.py warns once per shape).
extern "C" int pvn3d_mlp_split2_kernel(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host,
                                       int out_point_major, int flags) {
  if (!dims_host || n_layers < 1 || n_layers > S3_MAX_LAYERS) return 0;
  const int no_narrow = flags & PVN3D_MLP_NO_NARROW;
  if (is_sa && nw_signature(c_a, nsample, n_layers, dims_host, no_narrow) > 0) return 2;
  if (!is_sa && nwfp_ok(c_a, c_b, n_layers, dims_host, out_point_major, no_narrow)) return 2;
  return s3_ok(is_sa, c_a, c_b, nsample, n_layers, dims_host, 1, no_narrow) ? 1 : 0;
}

// fp16 x 2: per-layer (sw, ||W||_inf, max|b|) triples + the device-side input bounds -> S3Args; false when malformed
static bool s3_fill_scales(S3Args* a, int n_layers, const float* layer_meta, const float* bound_a, const float* bound_b,
                           float mul_b) {
  if (!layer_meta || !bound_a) return false;
  for (int l = 0; l < n_layers; ++l) {
    a->sw[l] = layer_meta[3 * l]; a->wnorm[l] = layer_meta[3 * l + 1]; a->bmax[l] = layer_meta[3 * l + 2];
    int e = 0;
    if (!(a->sw[l] > 0.f) || frexpf(a->sw[l], &e) != 0.5f || !(a->wnorm[l] >= 0.f) || !(a->bmax[l] >= 0.f)) return false;
  }
  a->bound_a = bound_a; a->bound_b = bound_b; a->mul_b = mul_b;
  return true;
}

static int s3_sa_entry(int arith, int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                       const float* features_pm, int ld_feat, const int* idx, int n_layers, const int* dims_host,
                       const void* const* w_split, const float* const* bias_padded, const float* layer_meta,
                       const float* bound_a, const float* bound_b, float mul_b, float* out_pm, int ld_out, int out_coff,
                       float* out_absmax, const float* out_row_mul, int flags, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (!xyz || !new_xyz || !idx || !out_pm || !features_pm || !dims_host || !w_split || !bias_padded)
    return (int)hipErrorInvalidValue;
  const int no_narrow = flags & PVN3D_MLP_NO_NARROW;
  // narrow chains (fp16 x 2): a kernel of their own; the six-feature instances (c == 6) take any row stride
  const int nsig = arith == 1 ? nw_signature(c, nsample, n_layers, dims_host, no_narrow) : -1;
  const bool narrow = nsig >= 0 && (nsig <= 2 || vec_ok(features_pm, ld_feat));
  if (!s3_ok(1, c, 0, nsample, n_layers, dims_host, arith, no_narrow) || dims_host[0] != c + 3 ||
      (!narrow && !vec_ok(features_pm, ld_feat)) || ld_feat < c || out_coff < 0 || ld_out < out_coff + dims_host[n_layers])
    return (int)hipErrorInvalidValue;
  S3Args a = {};
  if (!s3_fill(&a, n_layers, dims_host, w_split, bias_padded)) return (int)hipErrorInvalidValue;
  if (arith == 1 && !s3_fill_scales(&a, n_layers, layer_meta, bound_a, bound_b, mul_b)) return (int)hipErrorInvalidValue;
  a.out_absmax = (unsigned*)out_absmax;
  a.out_row_mul = out_row_mul; a.ident_a = ((flags & PVN3D_MLP_IDENTITY_A) && c == dims_host[1] && c % S3_KC == 0) ? 1 : 0;   // (a unit block needs a square, chunk-aligned table part)
  a.is_sa = 1;
  a.xyz = xyz; a.new_xyz = new_xyz; a.n = n; a.m = m; a.ns = nsample;
  a.idx = idx;
  a.tabA = features_pm; a.rowsA = n; a.ldA = ld_feat; a.nA = c / S3_KC;
  a.tail_w = 3;
  a.cols_total = m * nsample;
  a.n_frames = b;
  a.out = out_pm; a.point_major = 1; a.ld_out = ld_out; a.coff = out_coff;
  if (narrow) {
    const int rc = nw_launch(a, nsig, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  const int sig = s3_signature(1, n_layers, dims_host, nsample, arith);
  if (sig < 0 || (c % S3_KC) != 0 || !vec_ok(features_pm, ld_feat)) return (int)hipErrorInvalidValue;
  const int rc = s3_launch(a, sig, (hipStream_t)stream, arith);
  return rc < 0 ? (int)hipErrorInvalidValue : rc;
}
extern "C" int pvn3d_sa_mlp_maxpool_split(int b, int n, int m, int c, int nsample, const float* xyz,
                                          const float* new_xyz, const float* features_pm, int ld_feat, const int* idx,
                                          int n_layers, const int* dims_host, const void* const* w_split,
                                          const float* const* bias_padded, float* out_pm, int ld_out, int out_coff,
                                          void* stream) {
  return s3_sa_entry(0, b, n, m, c, nsample, xyz, new_xyz, features_pm, ld_feat, idx, n_layers, dims_host, w_split,
                     bias_padded, nullptr, nullptr, nullptr, 0.f, out_pm, ld_out, out_coff, nullptr, nullptr, 0, stream);
}
extern "C" int pvn3d_sa_mlp_maxpool_split2(int b, int n, int m, int c, int nsample, const float* xyz,
                                           const float* new_xyz, const float* features_pm, int ld_feat, const int* idx,
                                           int n_layers, const int* dims_host, const void* const* w_split2,
                                           const float* const* bias_padded, const float* layer_meta,
                                           const float* features_absmax, const float* xyz_absmax, float* out_pm,
                                           int ld_out, int out_coff, float* out_absmax, const float* out_row_mul,
                                           int flags, void* stream) {
  // |p - c| <= 2 max|xyz| bounds the relative coordinates whatever the index list holds
  return s3_sa_entry(1, b, n, m, c, nsample, xyz, new_xyz, features_pm, ld_feat, idx, n_layers, dims_host, w_split2,
                     bias_padded, layer_meta, features_absmax, xyz_absmax, 2.f, out_pm, ld_out, out_coff, out_absmax,
                     out_row_mul, flags, stream);
}

static int s3_fp_entry(int arith, int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                       const float* unknown_pm, int ld_unknown, const int* idx, const float* weight, int n_layers,
                       const int* dims_host, const void* const* w_split, const float* const* bias_padded,
                       const float* layer_meta, const float* bound_a, const float* bound_b, float* out,
                       int out_point_major, int ld_out, float* out_absmax, const float* out_row_mul, int flags,
                       void* stream) {
  if (b <= 0 || n <= 0) return 0;
  if (!known_pm || !idx || !weight || !out || !dims_host || !w_split || !bias_padded || (c1 > 0 && !unknown_pm))
    return (int)hipErrorInvalidValue;
  if (!s3_ok(0, c2, c1, 0, n_layers, dims_host, arith) || dims_host[0] != c2 + c1 || !vec_ok(known_pm, ld_known) ||
      ld_known < c2 || (c1 > 0 && ld_unknown < c1) || (c1 >= S3_KC && !vec_ok(unknown_pm, ld_unknown)) ||
      (out_point_major && ld_out < dims_host[n_layers]))
    return (int)hipErrorInvalidValue;
  S3Args a = {};
  if (!s3_fill(&a, n_layers, dims_host, w_split, bias_padded)) return (int)hipErrorInvalidValue;
  if (arith == 1 && (!s3_fill_scales(&a, n_layers, layer_meta, bound_a, bound_b, 1.f) || (c1 > 0 && !bound_b)))
    return (int)hipErrorInvalidValue;
  a.out_absmax = (unsigned*)out_absmax;
  a.out_row_mul = out_row_mul; a.ident_a = ((flags & PVN3D_MLP_IDENTITY_A) && c2 == dims_host[1] && c2 % S3_KC == 0) ? 1 : 0;
  a.is_sa = 0;
  a.idx = idx; a.weight = weight;
  a.tabA = known_pm; a.rowsA = m; a.ldA = ld_known; a.nA = c2 / S3_KC;
  a.tabB = unknown_pm; a.rowsB = n; a.ldB = ld_unknown; a.nB = c1 / S3_KC;
  a.tail_w = c1 % S3_KC;
  a.cols_total = n;
  a.n_frames = b;
  a.out = out; a.point_major = out_point_major ? 1 : 0; a.ld_out = ld_out; a.coff = 0;
  const int rc = s3_launch(a, s3_signature(0, n_layers, dims_host, 0, arith), (hipStream_t)stream, arith);
  return rc < 0 ? (int)hipErrorInvalidValue : rc;
}
extern "C" int pvn3d_fp_interp_mlp_split(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                                         const float* unknown_pm, int ld_unknown, const int* idx, const float* weight,
                                         int n_layers, const int* dims_host, const void* const* w_split,
                                         const float* const* bias_padded, float* out, int out_point_major, int ld_out,
                                         void* stream) {
  return s3_fp_entry(0, b, n, m, c2, c1, known_pm, ld_known, unknown_pm, ld_unknown, idx, weight, n_layers, dims_host,
                     w_split, bias_padded, nullptr, nullptr, nullptr, out, out_point_major, ld_out, nullptr, nullptr, 0, stream);
}
extern "C" int pvn3d_fp_interp_mlp_split2(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                                          const float* unknown_pm, int ld_unknown, const int* idx, const float* weight,
                                          int n_layers, const int* dims_host, const void* const* w_split2,
                                          const float* const* bias_padded, const float* layer_meta,
                                          const float* known_absmax, const float* unknown_absmax, float* out,
                                          int out_point_major, int ld_out, float* out_absmax, const float* out_row_mul,
                                          int flags, void* stream) {
  // (no narrow-chain kernel behind this entry point: of the flags only PVN3D_MLP_IDENTITY_A matters)
  // interpolation weights are non-negative and sum to 1: |interp(known)| <= max|known|
  return s3_fp_entry(1, b, n, m, c2, c1, known_pm, ld_known, unknown_pm, ld_unknown, idx, weight, n_layers, dims_host,
                     w_split2, bias_padded, layer_meta, known_absmax, unknown_absmax, out, out_point_major, ld_out, out_absmax,
                     out_row_mul, flags, stream);
}

// The pre-contracted form (DESIGN 4.7b''): the caller promises that the first c2 columns of layer 0's weights are the
// identity (PackedMLP.precontracted), i.e. layer 0 = relu(interp(known) + Wb.skip + b0) with known_pm holding the first
// conv's interpolated half per known point.  Same arguments and results as pvn3d_fp_interp_mlp_split2 (which computes
// the same thing by multiplying with that identity); the shapes the narrow kernel takes skip those MFMAs.
extern "C" int pvn3d_fp_interp_add_mlp_split2(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                                              const float* unknown_pm, int ld_unknown, const int* idx, const float* weight,
                                              int n_layers, const int* dims_host, const void* const* w_split2,
                                              const float* const* bias_padded, const float* layer_meta,
                                              const float* known_absmax, const float* unknown_absmax, float* out,
                                              int out_point_major, int ld_out, float* out_absmax, const float* out_row_mul,
                                              int flags, void* stream) {
  if (b > 0 && n > 0 && dims_host && known_pm && unknown_pm && idx && weight && out && w_split2 && bias_padded &&
      nwfp_ok(c2, c1, n_layers, dims_host, out_point_major, flags & PVN3D_MLP_NO_NARROW) && vec_ok(known_pm, ld_known) &&
      ld_known >= c2 &&
      ld_unknown >= c1 && unknown_absmax) {
    S3Args a = {};
    if (!s3_fill(&a, n_layers, dims_host, w_split2, bias_padded) ||
        !s3_fill_scales(&a, n_layers, layer_meta, known_absmax, unknown_absmax, 1.f))
      return (int)hipErrorInvalidValue;
    a.out_absmax = (unsigned*)out_absmax;
    a.out_row_mul = out_row_mul; a.ident_a = ((flags & PVN3D_MLP_IDENTITY_A) && c2 == dims_host[1] && c2 % S3_KC == 0) ? 1 : 0;
    a.is_sa = 0;
    a.idx = idx; a.weight = weight;
    a.tabA = known_pm; a.rowsA = m; a.ldA = ld_known;
    a.tabB = unknown_pm; a.rowsB = n; a.ldB = ld_unknown;
    a.cols_total = n;
    a.n_frames = b;
    a.out = out; a.point_major = 0; a.ld_out = ld_out; a.coff = 0;
    const int rc = nwfp_launch(a, (hipStream_t)stream);
    if (rc >= 0) return rc;
  }
  return pvn3d_fp_interp_mlp_split2(b, n, m, c2, c1, known_pm, ld_known, unknown_pm, ld_unknown, idx, weight, n_layers,
                                    dims_host, w_split2, bias_padded, layer_meta, known_absmax, unknown_absmax, out,
                                    out_point_major, ld_out, out_absmax, out_row_mul, flags, stream);
}

// max |x| over a point-major table [rows][ld] (channels [0, c)) -> *out (device float), as an atomic max on the bit
// pattern (non-negative floats order like unsigned integers).  *out must hold a value <= the result (0) beforehand.
namespace {
// thread -> (row slot ty, channel-group slot tx) with cx = 2^cx_log2 slots across a row: no division anywhere (a 64-bit
// `t / c4` per element made the first version of this kernel 4x slower than its memory traffic)
__global__ __launch_bounds__(256) void absmax_kernel(long long rows, int c, const float* __restrict__ src, int ld,
                                                      unsigned* __restrict__ out, int cx_log2, int vec) {
  const int cx = 1 << cx_log2, tx = threadIdx.x & (cx - 1), ty = threadIdx.x >> cx_log2, rpb = 256 >> cx_log2;
  float m = 0.f;
  if (vec) {
    const int c4 = c >> 2, rest = c & 3;
    for (long long r = (long long)blockIdx.x * rpb + ty; r < rows; r += (long long)gridDim.x * rpb) {
      const float* row = src + r * ld;
      for (int q = tx; q < c4; q += cx) {
        const float4 v = *reinterpret_cast<const float4*>(row + 4 * q);
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      }
      if (tx < rest) m = fmaxf(m, fabsf(row[4 * c4 + tx]));
    }
  } else {
    for (long long r = (long long)blockIdx.x * rpb + ty; r < rows; r += (long long)gridDim.x * rpb) {
      const float* row = src + r * ld;
      for (int q = tx; q < c; q += cx) m = fmaxf(m, fabsf(row[q]));
    }
  }
  // one atomic per workgroup (thousands of them on one address serialise at the L2: 12 ns each)
  __shared__ float s_m[4];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (m > 0.f) atomicMax(out, __float_as_uint(m));
  }
}
}  // namespace
extern "C" int pvn3d_absmax(long long rows, int c, const float* src, int ld_src, float* out_max, void* stream) {
  if (rows <= 0 || c <= 0) return 0;
  if (!src || !out_max || ld_src < c) return (int)hipErrorInvalidValue;
  const int vec = ((ld_src & 3) == 0 && ((uintptr_t)src & 15) == 0 && c >= 4) ? 1 : 0;
  const int per_row = vec ? (c + 3) / 4 : c;                 // work items across a row
  int cx_log2 = 0;
  while ((1 << cx_log2) < per_row && cx_log2 < 6) ++cx_log2;  // 1 .. 64 slots across
  const int rpb = 256 >> cx_log2;
  const long long want = (rows + rpb - 1) / rpb;
  const unsigned blocks = (unsigned)(want < 1024 ? want : 1024);
  hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, rows, c, src, ld_src, (unsigned*)out_max,
                     cx_log2, vec);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
