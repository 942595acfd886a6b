// sa_mlp.hip -- fused "group -> SharedMLP -> max-pool" (set abstraction) and
// "three_interpolate -> concat -> SharedMLP" (feature propagation) for gfx950, eval mode.
//
// Replaces, for inference, the data flow of _PointnetSAModuleBase.forward
// (pvn3d/lib/pointnet2_utils/pointnet2_modules.py:57-69): QueryAndGroup's gather + concat
// (pointnet2_utils.py:311-321), SharedMLP = [1x1 Conv2d -> BatchNorm2d -> ReLU] x L
// (lib/utils/etw_pytorch_utils/pytorch_utils.py:25-50) and F.max_pool2d over nsample; and of
// PointnetFPModule.forward (:183-206): three_interpolate, torch.cat, SharedMLP.
// The grouped (B, 3+C, npoint, nsample) tensor -- 69 MB per frame, the HBM-bound piece of the
// unfused path -- is never materialised: a workgroup gathers its 64 columns ((centre, sample)
// pairs, or 64 unknown points) channel-chunk by channel-chunk into LDS and runs the whole MLP
// chain on them.
//
// The contraction is the ONLY MFMA work of the hot path (north_star).  fp32 in / fp32 accumulate:
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the 157 TFLOP/s peak); the
// reference computes these layers in fp32 (cuDNN), so no reduced-precision shortcut is taken.
// BatchNorm (eval) is folded into the weights/bias on the host: W' = W * g/sqrt(var+eps),
// b' = beta - mean * g/sqrt(var+eps).
//
// Workgroup = 4 waves, 64 columns.  Layer l: D[M_l x 64] = relu(W_l[M_l x K_l] . H_{l-1}[K_l x 64] + b_l)
//   * A operand (weights): pre-packed on the host as [K/4][M/32][64 lanes][2] so that the
//     fragments of two consecutive 32x32x2 steps are one coalesced 512-byte load
//     (lane l, j: W[mt*32 + (l&31)][4*k4 + 2*j + (l>>5)]); K and M are zero padded to
//     multiples of 4 and 32; weights are shared by every workgroup and stay L2-resident.
//   * B operand (activations): LDS, H[k][64 cols] row-major -- a fragment read is two
//     conflict-free 128-byte rows (lane l: H[2*k2 + (l>>5)][ct*32 + (l&31)]).
//   * a wave owns row tiles mt = wave, wave+4, ... (<= 4) x 2 column tiles = <= 8 accumulator
//     tiles (128 VGPRs); after a layer the tile (bias, ReLU) is written back to the same LDS
//     buffer (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
//   * layer 0 reads its input in 32-channel chunks that the loader gathers into a
//     double-buffered LDS chunk; later layers read the full previous activation from LDS.
// Epilogue SA: max over the nsample columns of each centre (lanes), store (B, M, npoint).
// Epilogue FP: store (B, M, n).
#include <cstdlib>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int SM_COLS = 64;
constexpr int SM_KC = 32;            // input channels per layer-0 chunk
constexpr int SM_MAX_LAYERS = 4;
constexpr int SM_MAX_MT = 16;        // M <= 512

struct MlpDesc {
  int n_layers;
  int K[SM_MAX_LAYERS];              // true input width of layer l
  int M[SM_MAX_LAYERS];              // true output width
  const float* W[SM_MAX_LAYERS];     // packed [ceil(K/4)][ceil(M/32)][64][2]
  const float* bias[SM_MAX_LAYERS];  // [ceil(M/32)*32], zero padded
};

struct SaSrc {      // loader of the set-abstraction input: column = (centre j, sample s)
  const float* xyz;       // (B, n, 3)
  const float* new_xyz;   // (B, m, 3)
  const float* feat;      // (B, C, n) or null
  const int* idx;         // (B, m, ns)
  int n, m, ns, C, use_xyz;
};

struct FpSrc {      // loader of the feature-propagation input: column = unknown point j
  const float* known_feats;   // (B, C2, m)
  const float* unknow_feats;  // (B, C1, n) or null
  const int* idx;             // (B, n, 3)
  const float* weight;        // (B, n, 3)
  int n, m, C2, C1;
};

// ---- one layer on the matrix cores --------------------------------------------------------
// acc tiles [t][ct]; Hin in LDS [K][64] (layer >= 1) or streamed in chunks (layer 0).
// K is consumed in PAIRS of 32x32x2 steps (4 input channels): one 8-byte load per lane fetches
// the A fragments of both steps of a row tile.  The loop is branch-free (exact tile count NTC,
// clamped prefetch addresses, odd pair peeled after the loop) so that the compiler can count
// outstanding loads -- with conditional loads in the loop it falls back to s_waitcnt vmcnt(0)
// at the loop head and the prefetch is lost.  Two register sets, each refilled right after its
// MFMAs issue (two pairs = 4 steps ahead of its next use).
template <int NTC, int NT, int CT>
__device__ __forceinline__ void mma_pairs(f32x16 (&acc)[NT][CT], const float2* __restrict__ wp,
                                          size_t tstride, size_t pstride,
                                          const float* __restrict__ hr, int hstride, int pairs) {
  if (pairs <= 0) return;
  const int last = pairs - 1;
  float2 ac[NTC], an[NTC];
  float bc[2][CT], bn[2][CT];
#define SM_LD(A, B, P)                                                          \
  do {                                                                          \
    const float2* w_ = wp + (size_t)(P) * pstride;                              \
    _Pragma("unroll") for (int t = 0; t < NTC; ++t) A[t] = w_[(size_t)t * tstride]; \
    const float* h_ = hr + (size_t)(P) * 4 * hstride;                           \
    _Pragma("unroll") for (int c = 0; c < CT; ++c) {                            \
      B[0][c] = h_[c * 32];                                                     \
      B[1][c] = h_[2 * hstride + c * 32];                                       \
    }                                                                           \
  } while (0)
#define SM_MM(A, B)                                                                              \
  do {                                                                                           \
    _Pragma("unroll") for (int t = 0; t < NTC; ++t)                                              \
      _Pragma("unroll") for (int c = 0; c < CT; ++c)                                             \
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t].x, B[0][c], acc[t][c], 0, 0, 0);   \
    _Pragma("unroll") for (int t = 0; t < NTC; ++t)                                              \
      _Pragma("unroll") for (int c = 0; c < CT; ++c)                                             \
        acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[t].y, B[1][c], acc[t][c], 0, 0, 0);   \
  } while (0)
  SM_LD(ac, bc, 0);
  SM_LD(an, bn, min(1, last));
  int p = 0;
  for (; p + 1 < pairs; p += 2) {
    SM_MM(ac, bc);
    SM_LD(ac, bc, min(p + 2, last));
    SM_MM(an, bn);
    SM_LD(an, bn, min(p + 3, last));
  }
  if (p < pairs) SM_MM(ac, bc);
#undef SM_LD
#undef SM_MM
}

// dispatch on the (wave-uniform) number of row tiles this wave owns
template <int NT, int NW>
__device__ __forceinline__ void mma_chunk(f32x16 (&acc)[NT][2], const float* __restrict__ Wp,
                                          int mt_total, int wave, int nt, int pair_begin,
                                          int pair_end, const float* __restrict__ Hrows /*row 4*pair_begin*/,
                                          int lane) {
  const float2* wp = reinterpret_cast<const float2*>(Wp) + ((size_t)pair_begin * mt_total + wave) * 64 + lane;
  const float* hr = Hrows + (lane >> 5) * SM_COLS + (lane & 31);
  const size_t pstride = (size_t)mt_total * 64;
  const int pairs = pair_end - pair_begin;
  if (nt >= NT) mma_pairs<NT, NT, 2>(acc, wp, (size_t)NW * 64, pstride, hr, SM_COLS, pairs);
  else if constexpr (NT > 1) {
    if (nt == NT - 1) mma_pairs<NT - 1, NT, 2>(acc, wp, (size_t)NW * 64, pstride, hr, SM_COLS, pairs);
  }
}

// max over groups of ns (power of two <= 32) consecutive lanes with DPP row operations fused
// into v_max_f32 (a ds_bpermute butterfly costs ~5x the MFMA time of a narrow chain).
// ns <= 16: every lane of a group ends with the group max; ns == 32: lanes 16..31 / 48..63 do.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {
  const int o = __builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
  return fmaxf(v, __int_as_float(o));
}
__device__ __forceinline__ float seg_max(float v, int ns) {
  if (ns >= 2) v = dpp_max<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  if (ns >= 4) v = dpp_max<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  if (ns >= 8) v = dpp_max<0x141, 0xF>(v);    // row_half_mirror
  if (ns >= 16) v = dpp_max<0x140, 0xF>(v);   // row_mirror
  if (ns >= 32) v = dpp_max<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  return v;
}
__device__ __forceinline__ bool seg_leader(int col, int ns) {
  return ns >= 32 ? (col == 16) : ((col & (ns - 1)) == 0);
}

// accumulator tile <- bias of its rows (C/D map: reg r holds row (r&3) + 8*(r>>2) + 4*half)
__device__ __forceinline__ void acc_bias(f32x16& acc, const float* __restrict__ sb /*tile's 32 biases, LDS*/,
                                         int half) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(sb + 8 * g + 4 * half);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
}

// all biases of the chain -> LDS (layer l at offset sum_{i<l} roundup32(M_i)); caller syncs
__device__ __forceinline__ void stage_bias(const MlpDesc& d, float* __restrict__ sb, int tid, int nthreads) {
  int off = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    const int mp = ((d.M[l] + 31) >> 5) << 5;
    for (int i = tid; i < mp; i += nthreads) sb[off + i] = d.bias[l][i];
    off += mp;
  }
}

template <int NT, int NW>
__device__ __forceinline__ void store_act(const f32x16 (&acc)[NT][2], int wave, int nt,
                                          float* __restrict__ H, int lane) {
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
      const int mt = wave + NW * t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        H[row * SM_COLS + col] = fmaxf(acc[t][0][r], 0.f);
        H[row * SM_COLS + 32 + col] = fmaxf(acc[t][1][r], 0.f);
      }
    }
  }
}

// dynamic LDS: H [hrows][64] | chunk [2][32][64] | bias [sum roundup32(M_l)]
template <bool IS_SA, int NT, int NW>
__device__ __forceinline__ void mlp_chain_body(const MlpDesc& d, const SaSrc& sa, const FpSrc& fp, int hrows,
                                               int cols_total, float* __restrict__ out) {
  extern __shared__ float s_mem[];
  float* H = s_mem;
  float* chunk = s_mem + (size_t)hrows * SM_COLS;
  float* s_bias = chunk + 2 * SM_KC * SM_COLS;
  const int tid = threadIdx.x, lane = tid & 63;
  stage_bias(d, s_bias, tid, NW * 64);     // visible after the first barrier below
  // scalar wave index: keeps every "does this wave own row tile t" test a uniform branch
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = blockIdx.y;
  const int col0 = blockIdx.x * SM_COLS;

  // ---- per-thread loader state: this thread fills column lc, rows lr0 + 4*i of every chunk
  constexpr int LROWS = SM_KC / NW;   // chunk rows filled per thread
  const int lc = tid & 63, lr0 = tid >> 6;
  const int gcol = col0 + lc;
  const bool cvalid = gcol < cols_total;
  int id0 = 0, id1 = 0, id2 = 0;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f;
  size_t ctr = 0;      // offset of this column's centre in new_xyz
  if (IS_SA) {
    if (cvalid) {
      id0 = sa.idx[(size_t)bi * sa.m * sa.ns + gcol];
      ctr = ((size_t)bi * sa.m + gcol / sa.ns) * 3;
    }
  } else {
    if (cvalid) {
      const int* ip = fp.idx + ((size_t)bi * fp.n + gcol) * 3;
      const float* wp = fp.weight + ((size_t)bi * fp.n + gcol) * 3;
      id0 = ip[0]; id1 = ip[1]; id2 = ip[2];
      w0 = wp[0]; w1 = wp[1]; w2 = wp[2];
    }
  }
  // plain local copies: capturing the by-value kernel-argument structs by reference would pin
  // them in scratch memory and turn every field access of the loader into a scratch load
  const float* const sa_xyz = sa.xyz; const float* const sa_feat = sa.feat;
  const float* const sa_nxyz = sa.new_xyz;
  const int sa_n = sa.n, sa_C = sa.C, sa_c3 = sa.use_xyz ? 3 : 0;
  const float* const fp_kf = fp.known_feats; const float* const fp_uf = fp.unknow_feats;
  const int fp_n = fp.n, fp_m = fp.m, fp_C2 = fp.C2, fp_C1 = fp.C1;
  auto load_input = [=](int c) -> float {   // value of input channel c for this thread's column
#ifdef SM_EXP_NOGATHER
    if (c >= 32) return 1.0f;
#endif
    if (!cvalid) return 0.f;
    if (IS_SA) {
      if (c < sa_c3) {
        // grouped_xyz -= new_xyz (the centre is re-read: a select chain over three registers
        // is turned into a scratch-array lookup by the compiler)
        return sa_xyz[((size_t)bi * sa_n + id0) * 3 + c] - sa_nxyz[ctr + c];
      }
      const int cf = c - sa_c3;
      return cf < sa_C ? sa_feat[((size_t)bi * sa_C + cf) * sa_n + id0] : 0.f;
    } else {
      if (c < fp_C2) {
        const float* row = fp_kf + ((size_t)bi * fp_C2 + c) * fp_m;
        return row[id0] * w0 + row[id1] * w1 + row[id2] * w2;   // three_interpolate, unfused order
      }
      const int cu = c - fp_C2;
      return cu < fp_C1 ? fp_uf[((size_t)bi * fp_C1 + cu) * fp_n + gcol] : 0.f;
    }
  };

  f32x16 acc[NT][2];
  int boff = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    const int K = d.K[l], M = d.M[l];
    const int mt_total = (M + 31) >> 5;
    const int nt = (mt_total - wave + NW - 1) / NW;     // row tiles of this wave (<= NT)
    const int pairs_total = (K + 3) >> 2;
    if (l == 0) {
      const int n_chunks = (K + SM_KC - 1) / SM_KC;
      float stage[LROWS];
#pragma unroll
      for (int i = 0; i < LROWS; ++i) stage[i] = load_input(lr0 + NW * i);
#pragma unroll
      for (int i = 0; i < LROWS; ++i) chunk[(lr0 + NW * i) * SM_COLS + lc] = stage[i];
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) {
          acc_bias(acc[t][0], s_bias + boff + (wave + NW * t) * 32, lane >> 5);
          acc[t][1] = acc[t][0];
        }
      for (int ch = 0; ch < n_chunks; ++ch) {
        const int buf = ch & 1;
        const bool more = ch + 1 < n_chunks;
        if (more) {
#pragma unroll
          for (int i = 0; i < LROWS; ++i) stage[i] = load_input((ch + 1) * SM_KC + lr0 + NW * i);
        }
        const int pb = ch * (SM_KC / 4);
        const int pe = min(pb + SM_KC / 4, pairs_total);
        mma_chunk<NT, NW>(acc, d.W[l], mt_total, wave, nt, pb, pe, chunk + (size_t)buf * SM_KC * SM_COLS, lane);
        if (more) {
          float* cb = chunk + (size_t)(buf ^ 1) * SM_KC * SM_COLS;
#pragma unroll
          for (int i = 0; i < LROWS; ++i) cb[(lr0 + NW * i) * SM_COLS + lc] = stage[i];
        }
        __syncthreads();
      }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) {
          acc_bias(acc[t][0], s_bias + boff + (wave + NW * t) * 32, lane >> 5);
          acc[t][1] = acc[t][0];
        }
      mma_chunk<NT, NW>(acc, d.W[l], mt_total, wave, nt, 0, pairs_total, H, lane);
      __syncthreads();   // every wave has finished reading H_{l-1}
    }
    boff += mt_total * 32;
    if (l + 1 < d.n_layers) {
      store_act<NT, NW>(acc, wave, nt, H, lane);
      // rows [M, roundup32(M)) were written as relu(0 + 0) = 0 (zero-padded weights and bias),
      // which covers the next layer's K rounded up to a multiple of 4
      __syncthreads();
    }
  }

  // ---- epilogue on the last layer's accumulators
  const int L = d.n_layers - 1;
  const int M = d.M[L];
  const int mt_total = (M + 31) >> 5;
  const int nt = (mt_total - wave + NW - 1) / NW;
  const int half = lane >> 5, col = lane & 31;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
      const int mt = wave + NW * t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v0 = fmaxf(acc[t][0][r], 0.f), v1 = fmaxf(acc[t][1][r], 0.f);
        if (IS_SA) {
          // max over the nsample consecutive columns of each centre (columns beyond cols_total
          // hold relu(bias) of zero inputs and belong to centres >= m, never stored)
          const int ns = sa.ns;
          if (ns > 32) v0 = fmaxf(v0, v1);              // ns == 64: both tiles are one centre
          v0 = seg_max(v0, ns);
          if (ns <= 32) v1 = seg_max(v1, ns);
          if (row < M && seg_leader(col, ns)) {
            const int cb = ns >= 32 ? 0 : col;
            const int j0 = (col0 + cb) / ns, j1 = (col0 + 32 + cb) / ns;
            if (j0 < sa.m) out[((size_t)bi * M + row) * sa.m + j0] = v0;
            if (ns <= 32 && j1 < sa.m) out[((size_t)bi * M + row) * sa.m + j1] = v1;
          }
        } else {
          if (row < M) {
            const int g0 = col0 + col, g1 = col0 + 32 + col;
            if (g0 < cols_total) out[((size_t)bi * M + row) * fp.n + g0] = v0;
            if (g1 < cols_total) out[((size_t)bi * M + row) * fp.n + g1] = v1;
          }
        }
      }
    }
  }
}

template <bool IS_SA, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 8 ? (IS_SA ? 4 : 2) : 3)) void mlp_chain_kernel(
    MlpDesc d, SaSrc sa, FpSrc fp, int hrows, int cols_total, float* __restrict__ out) {
  mlp_chain_body<IS_SA, 1, NW>(d, sa, fp, hrows, cols_total, out);
}

// Two row tiles per wave (M > 256).
template <bool IS_SA>
__global__ __launch_bounds__(512) void mlp_chain_wide_kernel(
    MlpDesc d, SaSrc sa, FpSrc fp, int hrows, int cols_total, float* __restrict__ out) {
  mlp_chain_body<IS_SA, 2, 8>(d, sa, fp, hrows, cols_total, out);
}

// ---------------------------------------------------------------------------------------
// Narrow layers (every M <= 128): column-sliced variant.  A wave owns 32 columns and ALL row
// tiles (<= 4), keeps its own slice of the activations in LDS and gathers its own input
// chunk, so there is no workgroup barrier anywhere -- with M <= 64 the row-split kernel above
// leaves 2-3 of its 4 waves without a row tile.  Workgroup = 2 waves = 64 columns.
// dynamic LDS: bias [bias_floats] | per wave: H [hrows][32] | chunk [32][32].
// ---------------------------------------------------------------------------------------
template <bool IS_SA, int NTR>
__global__ __launch_bounds__(128, (!IS_SA ? 1 : (NTR == 1 ? 4 : (NTR == 2 ? 3 : 2)))) void mlp_chain_cols_kernel(MlpDesc d, SaSrc sa, FpSrc fp,
                                                             int hrows, int bias_floats,
                                                             int cols_total,
                                                             float* __restrict__ out) {
  extern __shared__ float s_mem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* s_bias = s_mem;
  float* H = s_mem + bias_floats + (size_t)wave * (hrows + SM_KC) * 32;
  float* chunk = H + (size_t)hrows * 32;
  const int bi = blockIdx.y;
  const int col0 = blockIdx.x * 64 + wave * 32;
  stage_bias(d, s_bias, tid, 128);
  __syncthreads();                         // the only barrier of this kernel
  if (col0 >= cols_total) return;          // wave-uniform
  const int half = lane >> 5, col = lane & 31;

  // loader: lane fills column `col`, rows half + 2*i (i < 16) of every 32-row chunk
  const int gcol = col0 + col;
  const bool cvalid = gcol < cols_total;
  int id0 = 0, id1 = 0, id2 = 0;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f;
  size_t ctr = 0;      // offset of this column's centre in new_xyz
  if (IS_SA) {
    if (cvalid) {
      id0 = sa.idx[(size_t)bi * sa.m * sa.ns + gcol];
      ctr = ((size_t)bi * sa.m + gcol / sa.ns) * 3;
    }
  } else if (cvalid) {
    const int* ip = fp.idx + ((size_t)bi * fp.n + gcol) * 3;
    const float* wp = fp.weight + ((size_t)bi * fp.n + gcol) * 3;
    id0 = ip[0]; id1 = ip[1]; id2 = ip[2];
    w0 = wp[0]; w1 = wp[1]; w2 = wp[2];
  }
  const float* const sa_xyz = sa.xyz; const float* const sa_feat = sa.feat;
  const float* const sa_nxyz = sa.new_xyz;
  const int sa_n = sa.n, sa_C = sa.C, sa_c3 = sa.use_xyz ? 3 : 0;
  const float* const fp_kf = fp.known_feats; const float* const fp_uf = fp.unknow_feats;
  const int fp_n = fp.n, fp_m = fp.m, fp_C2 = fp.C2, fp_C1 = fp.C1;
  auto load_input = [=](int c) -> float {
    if (!cvalid) return 0.f;
    if (IS_SA) {
      if (c < sa_c3) {
        return sa_xyz[((size_t)bi * sa_n + id0) * 3 + c] - sa_nxyz[ctr + c];
      }
      const int cf = c - sa_c3;
      return cf < sa_C ? sa_feat[((size_t)bi * sa_C + cf) * sa_n + id0] : 0.f;
    } else {
      if (c < fp_C2) {
        const float* row = fp_kf + ((size_t)bi * fp_C2 + c) * fp_m;
        return row[id0] * w0 + row[id1] * w1 + row[id2] * w2;
      }
      const int cu = c - fp_C2;
      return cu < fp_C1 ? fp_uf[((size_t)bi * fp_C1 + cu) * fp_n + gcol] : 0.f;
    }
  };

  f32x16 acc[NTR][1];
  auto mma = [&](const float* __restrict__ Wp, int mt_total, int pair_begin, int pair_end,
                 const float* __restrict__ rows) {
    const float2* wp = reinterpret_cast<const float2*>(Wp) + (size_t)pair_begin * mt_total * 64 + lane;
    const float* hr = rows + half * 32 + col;
    const size_t pstride = (size_t)mt_total * 64;
    const int pairs = pair_end - pair_begin;
    if (mt_total == 1) mma_pairs<1, NTR, 1>(acc, wp, 64, pstride, hr, 32, pairs);
    if constexpr (NTR >= 2) { if (mt_total == 2) mma_pairs<2, NTR, 1>(acc, wp, 64, pstride, hr, 32, pairs); }
    if constexpr (NTR >= 4) {
      if (mt_total == 3) mma_pairs<3, NTR, 1>(acc, wp, 64, pstride, hr, 32, pairs);
      if (mt_total == 4) mma_pairs<4, NTR, 1>(acc, wp, 64, pstride, hr, 32, pairs);
    }
  };

  int boff = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    const int K = d.K[l], M = d.M[l];
    const int mt_total = (M + 31) >> 5;
#pragma unroll
    for (int t = 0; t < NTR; ++t)
      if (t < mt_total) acc_bias(acc[t][0], s_bias + boff + t * 32, half);
    boff += mt_total * 32;
    const int pairs_total = (K + 3) >> 2;
    if (l == 0) {
      const int n_chunks = (K + SM_KC - 1) / SM_KC;
      float stage[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[i] = load_input(half + 2 * i);
      for (int ch = 0; ch < n_chunks; ++ch) {
#pragma unroll
        for (int i = 0; i < 16; ++i) chunk[(half + 2 * i) * 32 + col] = stage[i];
        if (ch + 1 < n_chunks) {   // gathers of the next chunk fly while this one is multiplied
#pragma unroll
          for (int i = 0; i < 16; ++i) stage[i] = load_input((ch + 1) * SM_KC + half + 2 * i);
        }
        const int pb = ch * (SM_KC / 4);
        mma(d.W[l], mt_total, pb, min(pb + SM_KC / 4, pairs_total), chunk);
      }
    } else {
      mma(d.W[l], mt_total, 0, pairs_total, H);
    }
    if (l + 1 < d.n_layers) {
#pragma unroll
      for (int t = 0; t < NTR; ++t) {
        if (t < mt_total) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            H[row * 32 + col] = fmaxf(acc[t][0][r], 0.f);
          }
        }
      }
    }
  }

  const int L = d.n_layers - 1;
  const int M = d.M[L];
  const int mt_total = (M + 31) >> 5;
#pragma unroll
  for (int t = 0; t < NTR; ++t) {
    if (t < mt_total) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = fmaxf(acc[t][0][r], 0.f);
        if (IS_SA) {
          const int ns = sa.ns;   // 2..32, power of two
          v = seg_max(v, ns);
          if (row < M && seg_leader(col, ns)) {
            const int j = (col0 + (ns >= 32 ? 0 : col)) / ns;
            if (j < sa.m) out[((size_t)bi * M + row) * sa.m + j] = v;
          }
        } else if (row < M && gcol < cols_total) {
          out[((size_t)bi * M + row) * fp.n + gcol] = v;
        }
      }
    }
  }
}

template <bool IS_SA>
int launch_chain(const MlpDesc& d, const SaSrc& sa, const FpSrc& fp, int b, int cols_total,
                 float* out, hipStream_t st) {
  int hrows = 2, max_mt = 1;
  for (int l = 0; l < d.n_layers; ++l) {
    if (l + 1 < d.n_layers) hrows = max(hrows, ((d.M[l] + 31) / 32) * 32 + 2);
    max_mt = max(max_mt, (d.M[l] + 31) / 32);
  }
  if (max_mt > SM_MAX_MT) return (int)hipErrorInvalidValue;
  const bool ns_ok = !IS_SA || sa.ns <= 32;
  // Measured (MI355X, 64 frames): the column-sliced kernel wins for M <= 64 (level 0:
  // 0.82 -> 0.36 ms and 2.66 -> 1.86 ms) and loses for M = 128 with K >= 99 (every wave
  // re-fetches all weight fragments): 1.4 -> 2.4 ms, so it is used for <= 2 row tiles only.
  static const int cols_max_mt = [] {
    const char* e = getenv("PVN3D_MLP_COLS_MAX_MT");     // tuning override
    return e ? atoi(e) : 2;
  }();
  if (max_mt <= cols_max_mt && max_mt <= 4 && ns_ok) {   // narrow chain: column-sliced kernel
    int hr = 2;
    for (int l = 0; l + 1 < d.n_layers; ++l) hr = max(hr, ((d.M[l] + 31) / 32) * 32 + 2);
    int bias_floats = 0;
    for (int l = 0; l < d.n_layers; ++l) bias_floats += ((d.M[l] + 31) / 32) * 32;
    const size_t lds2 = ((size_t)2 * (hr + SM_KC) * 32 + bias_floats) * sizeof(float);
    const dim3 grid2(pvn3d_ceil_div(cols_total, 64), b);
#define SM_LAUNCH_COLS(NTR)                                                                    \
  do {                                                                                         \
    auto kern = mlp_chain_cols_kernel<IS_SA, NTR>;                                             \
    if (lds2 > 48 * 1024)                                                                      \
      PVN3D_RETURN_IF_ERR(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                              (int)lds2));                                     \
    hipLaunchKernelGGL(kern, grid2, dim3(128), lds2, st, d, sa, fp, hr, bias_floats, cols_total, out); \
  } while (0)
    if (max_mt <= 1) SM_LAUNCH_COLS(1); else if (max_mt <= 2) SM_LAUNCH_COLS(2); else SM_LAUNCH_COLS(4);
#undef SM_LAUNCH_COLS
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
  int bias_all = 0;
  for (int l = 0; l < d.n_layers; ++l) bias_all += ((d.M[l] + 31) / 32) * 32;
  const size_t lds = ((size_t)hrows * SM_COLS + 2 * SM_KC * SM_COLS + bias_all) * sizeof(float);
  if (lds > 160 * 1024) return (int)hipErrorInvalidValue;
  const dim3 grid(pvn3d_ceil_div(cols_total, SM_COLS), b);
#define SM_LAUNCH(KERN, NW)                                                                      \
  do {                                                                                         \
    auto kern = KERN;                                                                          \
    if (lds > 48 * 1024)                                                                       \
      PVN3D_RETURN_IF_ERR(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),             \
                                              hipFuncAttributeMaxDynamicSharedMemorySize,      \
                                              (int)lds));                                      \
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, st, d, sa, fp, hrows, cols_total, out); \
  } while (0)
  // wide layers: 8 waves (two per SIMD hide each other's L2 / LDS waits), <= 2 row tiles each;
  // narrow layers (<= 4 row tiles): 4 waves, one row tile each
  if (max_mt <= 4) SM_LAUNCH((mlp_chain_kernel<IS_SA, 4>), 4);
  else if (max_mt <= 8) SM_LAUNCH((mlp_chain_kernel<IS_SA, 8>), 8);
  else SM_LAUNCH((mlp_chain_wide_kernel<IS_SA>), 8);
#undef SM_LAUNCH
  PVN3D_LAUNCH_CHECK();
  return 0;
}

bool fill_desc(MlpDesc* d, int n_layers, const int* dims, const float* const* W,
               const float* const* bias) {
  if (n_layers < 1 || n_layers > SM_MAX_LAYERS) return false;
  d->n_layers = n_layers;
  for (int l = 0; l < n_layers; ++l) {
    d->K[l] = dims[l];
    d->M[l] = dims[l + 1];
    d->W[l] = W[l];
    d->bias[l] = bias[l];
    if (dims[l] <= 0 || dims[l + 1] <= 0 || !W[l] || !bias[l]) return false;
  }
  return true;
}

}  // namespace

extern "C" int pvn3d_sa_mlp_maxpool(int b, int n, int m, int c, int nsample, int use_xyz,
                                    const float* xyz, const float* new_xyz,
                                    const float* features, const int* idx, int n_layers,
                                    const int* dims_host, const float* const* w_packed,
                                    const float* const* bias_padded, float* out, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (nsample <= 0 || (nsample & (nsample - 1)) || nsample > 64 || !xyz || !new_xyz || !idx ||
      !out || !dims_host || !w_packed || !bias_padded)
    return (int)hipErrorInvalidValue;
  MlpDesc d;
  if (!fill_desc(&d, n_layers, dims_host, w_packed, bias_padded)) return (int)hipErrorInvalidValue;
  if (dims_host[0] != (use_xyz ? 3 : 0) + (features ? c : 0)) return (int)hipErrorInvalidValue;
  SaSrc sa = {xyz, new_xyz, features, idx, n, m, nsample, features ? c : 0, use_xyz};
  FpSrc fp = {};
  return launch_chain<true>(d, sa, fp, b, m * nsample, out, (hipStream_t)stream);
}

extern "C" int pvn3d_fp_interp_mlp(int b, int n, int m, int c2, int c1, const float* known_feats,
                                   const float* unknow_feats, const int* idx,
                                   const float* weight, int n_layers, const int* dims_host,
                                   const float* const* w_packed, const float* const* bias_padded,
                                   float* out, void* stream) {
  if (b <= 0 || n <= 0) return 0;
  if (!known_feats || !idx || !weight || !out || !dims_host || !w_packed || !bias_padded ||
      (c1 > 0 && !unknow_feats))
    return (int)hipErrorInvalidValue;
  MlpDesc d;
  if (!fill_desc(&d, n_layers, dims_host, w_packed, bias_padded)) return (int)hipErrorInvalidValue;
  if (dims_host[0] != c2 + c1) return (int)hipErrorInvalidValue;
  SaSrc sa = {};
  FpSrc fp = {known_feats, unknow_feats, idx, weight, n, m, c2, c1};
  return launch_chain<false>(d, sa, fp, b, n, out, (hipStream_t)stream);
}
