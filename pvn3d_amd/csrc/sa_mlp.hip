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
// Feature tensors are POINT-MAJOR here ((B, points, channels), row stride `ld`): the gather of a
// neighbour is then one contiguous row, read 16 bytes per lane with 8 lanes per row (8 cache
// lines per wave instruction instead of 64 for a channel-major gather), and the pooled output
// of a centre is one contiguous row as well.  The host glue passes the reference's (B, C, n)
// tensors as transposed views, so nothing is copied between levels.
//
// Row-split kernel (mlp_chain_body): workgroup = NW waves, 64 columns.
// Layer l: D[M_l x 64] = relu(W_l[M_l x K_l] . H_{l-1}[K_l x 64] + b_l)
//   * A operand (weights): pre-packed on the host as [K/4][M/32][64 lanes][2] so that the
//     fragments of two consecutive 32x32x2 steps ("a pair" = 4 input channels) are one coalesced
//     512-byte load (lane l, j: W[mt*32 + (l&31)][4*k4 + 2*j + (l>>5)]); K and M are zero padded
//     to multiples of 4 and 32; weights are shared by every workgroup and stay L2-resident.
//     A ring of 8 register sets is refilled right after each pair's MFMAs issue, 8 pairs (one
//     input chunk) ahead, and runs on across chunk boundaries: in-order vmcnt means a weight
//     load queued behind the gathers of the next chunk completes after them, so those loads
//     must not be needed for a whole chunk.  The steady-state loops are branch-free (exact
//     tile count as template parameter, clamped prefetch addresses, loaders with multiplies
//     instead of branches) so that the compiler counts outstanding loads instead of falling
//     back to s_waitcnt vmcnt(0).
//   * B operand (activations): LDS.  Layer 0 streams its input in 32-channel chunks, double
//     buffered, rows of 96 floats shifted by 4*(row/4) columns (conflict-free for both the row-gather
//     writer and the fragment reader); later layers read the previous activation H[k][64].
//   * a wave owns row tiles mt = wave, wave+NW, ... (<= NT) x 2 column tiles; accumulators
//     start from the bias; after a layer relu(acc) goes back to H
//     (C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
// Epilogue SA: max over the nsample columns of each centre (through a wave-private LDS patch),
// store point-major.
// Epilogue FP: store point-major (intermediate levels) or (B, M, n) (the module's API layout).
#include <type_traits>

#include "common.h"

#ifdef SM_PROBE
// tuning builds only (tools/build_probe_lib.sh, tools/mlp_probe.py): cycles spent between consecutive
// stamps, summed over ALL workgroups of a launch (wave 0's view); slot 16+i counts the visits of stamp i
// (1024 independent slot sets indexed by workgroup id: atomics on ONE word serialise at ~11 ns each and
// the in-order vmcnt makes every later load of the stamping wave wait for them)
__device__ unsigned long long g_sm_probe[1024 * 32];
#define SM_PROBE_DECL unsigned long long sm_last_ = __builtin_readcyclecounter()
#define SM_STAMP(i)                                                                  \
  do {                                                                               \
    if (threadIdx.x == 0) {                                                          \
      const unsigned long long now_ = __builtin_readcyclecounter();                  \
      unsigned long long* slot_ = g_sm_probe + 32 * ((blockIdx.x + gridDim.x * blockIdx.y) & 1023u); \
      atomicAdd(slot_ + (i), now_ - sm_last_);                                       \
      atomicAdd(slot_ + 16 + (i), 1ull);                                             \
      sm_last_ = now_;                                                               \
    }                                                                                \
  } while (0)
extern "C" int pvn3d_debug_mlp_probe_read(unsigned long long* host32) {
  static unsigned long long all[1024 * 32];
  hipError_t e = hipMemcpyFromSymbol(all, HIP_SYMBOL(g_sm_probe), sizeof(all));
  if (e != hipSuccess) return (int)e;
  for (int i = 0; i < 32; ++i) {
    host32[i] = 0;
    for (int s = 0; s < 1024; ++s) host32[i] += all[s * 32 + i];
  }
  for (int i = 0; i < 1024 * 32; ++i) all[i] = 0;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sm_probe), all, sizeof(all));
}
#else
#define SM_PROBE_DECL do { } while (0)
#define SM_STAMP(i) do { } while (0)
#endif

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int SM_COLS = 64;
constexpr int SM_KC = 32;            // input channels per layer-0 chunk
constexpr int SM_CP = SM_KC / 4;     // pairs per chunk = depth of the weight prefetch ring
constexpr int SM_MAX_LAYERS = 4;
constexpr int SM_MAX_MT = 16;        // M <= 512
constexpr int SM_EPAD = 68;          // row stride of the max-pool patch (16-byte aligned, conflict-free)
constexpr int SM_CW = 96;            // row stride of a layer-0 chunk buffer: 64 columns + room for the rotation
constexpr int SM_TAIL_ROWS = 8;      // widest layer-0 tail the specialised (GEN = false) kernels take

struct MlpDesc {
  int n_layers;
  int K[SM_MAX_LAYERS];              // true input width of layer l
  int M[SM_MAX_LAYERS];              // true output width
  const float* W[SM_MAX_LAYERS];     // packed [ceil(K/4)][ceil(M/32)][64][2]
  const float* bias[SM_MAX_LAYERS];  // [ceil(M/32)*32], zero padded
};

struct RowSrc {       // point-major table: row r of frame b is tab[((size_t)b * rows + r) * ld ...]
  const float* tab;
  int rows, ld, width;   // width = channels taken from each row
};

struct SaSrc {        // set abstraction: column = (centre j, sample s); K = [feat.width][3 xyz]
  const float* xyz;       // (B, n, 3)
  const float* new_xyz;   // (B, m, 3)
  RowSrc feat;
  const int* idx;         // (B, m, ns)
  int n, m, ns, use_xyz;
};

struct FpSrc {        // feature propagation: column = unknown point; K = [known.width][unknown.width]
  RowSrc known, unknown;
  const int* idx;         // (B, n, 3)
  const float* weight;    // (B, n, 3)
  int n, m;
};

struct OutDesc {
  float* out;
  int point_major;        // 1: (B, points, ld) at channel offset coff; 0: (B, M, points)
  int ld, coff;
};

// ---- XCD-aware workgroup -> (frame, column block) mapping ---------------------------------
// Workgroups are dealt round-robin to the 8 XCDs in linear-id order, each XCD with its own
// 4 MB L2.  With the plain (x = column block, y = frame) grid every XCD touches every frame's
// feature rows; here XCD x works through frames x, x+8, x+16, ... one after the other.
__device__ __forceinline__ void xcd_frame_map(int& bi, int& bx) {
  const int nb = gridDim.x, nf = gridDim.y;
  bx = blockIdx.x;
  bi = blockIdx.y;
  if ((nf & 7) == 0) {
    const unsigned lin = blockIdx.x + (unsigned)nb * blockIdx.y;
    const unsigned q = lin >> 3;
    bi = (int)(lin & 7) + 8 * (int)(q / nb);
    bx = (int)(q % nb);
  }
}

// ---- small helpers ------------------------------------------------------------------------
// accumulator tile <- bias of its rows (C/D map: reg r holds row (r&3) + 8*(r>>2) + 4*half)
__device__ __forceinline__ void acc_bias(f32x16& acc, const float* __restrict__ sb /*tile's 32 biases, LDS*/,
                                         int half) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(sb + 8 * g + 4 * half);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
}

// all biases of the chain -> LDS (layer l at offset sum_{i<l} roundup32(M_i)); caller syncs.
// Every load is issued before the first store: one global round trip for the whole chain instead of
// one per layer (under load a round trip is several thousand cycles of a workgroup's prologue).
template <int NTHREADS>
__device__ __forceinline__ void stage_bias(const MlpDesc& d, float* __restrict__ sb, int tid) {
  constexpr int J = (SM_MAX_MT * 32 + NTHREADS - 1) / NTHREADS;
  float v[SM_MAX_LAYERS][J];
#pragma unroll
  for (int l = 0; l < SM_MAX_LAYERS; ++l) {
    const int mp = l < d.n_layers ? ((d.M[l] + 31) >> 5) << 5 : 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = tid + j * NTHREADS;
      v[l][j] = i < mp ? d.bias[l][i] : 0.f;
    }
  }
  int off = 0;
#pragma unroll
  for (int l = 0; l < SM_MAX_LAYERS; ++l) {
    const int mp = l < d.n_layers ? ((d.M[l] + 31) >> 5) << 5 : 0;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int i = tid + j * NTHREADS;
      if (i < mp) sb[off + i] = v[l][j];
    }
    off += mp;
  }
}

// ---- MFMA spans ---------------------------------------------------------------------------
// B fragment pair q of an LDS buffer: rows 4q + 2j + half, column tiles col, col+32.
// SWZ (layer-0 chunk buffers, row stride SM_CW): row r is stored shifted right by (r & ~3) columns --
// conflict-free for the row-gather writer (8 lanes x 4 rows per point) and for this reader -- and the
// rows are long enough that the shift never wraps, so every read of a span is the lane's base
// address plus a compile-time offset (a wrapping rotation needed two address registers per pair).
// !SWZ: activation buffer H[k][64].
// CS (column split): only column tile `ct` is fetched, into b[j][0].
template <bool SWZ, bool CS>
__device__ __forceinline__ void ld_b(float (&b)[2][2], const float* __restrict__ rows_half, int col, int q, int ct) {
  constexpr int RS = SWZ ? SM_CW : SM_COLS;
  const float* r0 = rows_half + q * 4 * RS + (SWZ ? 4 * q : 0);
  const int c0 = col + (CS ? ct * 32 : 0);
  b[0][0] = r0[c0];
  b[1][0] = r0[2 * RS + c0];
  if (!CS) { b[0][1] = r0[c0 + 32]; b[1][1] = r0[2 * RS + c0 + 32]; }
}

template <int NTC, int NT, bool CS>
__device__ __forceinline__ void mm_pair(f32x16 (&acc)[NT][2], const float2 (&a)[NTC > 0 ? NTC : 1],
                                        const float (&b)[2][2]) {
#pragma unroll
  for (int t = 0; t < NTC; ++t) {
    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[0][0], acc[t][0], 0, 0, 0);
    if (!CS) acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[0][1], acc[t][1], 0, 0, 0);
  }
#pragma unroll
  for (int t = 0; t < NTC; ++t) {
    acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[1][0], acc[t][0], 0, 0, 0);
    if (!CS) acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[1][1], acc[t][1], 0, 0, 0);
  }
}

struct WPtr {           // this lane's view of a layer's packed weights
  const float2* wp;     // + wave*64 + lane
  size_t pstride;       // float2 per pair = mt_total * 64
  size_t tstride;       // float2 between this wave's row tiles = NW * 64
  int last;             // last pair index of the layer
};

template <int NTC>
__device__ __forceinline__ void ring_load(float2 (&slot)[NTC > 0 ? NTC : 1], const WPtr& w, int p) {
  const float2* q = w.wp + (size_t)min(p, w.last) * w.pstride;
#pragma unroll
  for (int t = 0; t < NTC; ++t) slot[t] = q[(size_t)t * w.tstride];
}

// 8 pairs starting at global pair p0 (ring slot u holds pair p0+u), B rows from `rows_half`
// (local pair 0 = first row of the buffer); refills every slot with the pair 8 ahead.
template <int NTC, int NT, bool SWZ, bool CS>
__device__ __forceinline__ void span8(f32x16 (&acc)[NT][2], float2 (&ring)[SM_CP][NTC > 0 ? NTC : 1],
                                      const WPtr& w, int p0, const float* __restrict__ rows_half, int col,
                                      int ct) {
  if (NTC == 0) return;
  // B fragments one pair ahead (two pairs ahead measured no faster and costs 4 VGPRs, which
  // pushes the two-tile kernels over 240 -- the budget that leaves room for a 32-VGPR wave of a
  // concurrently running VALU kernel on the same SIMD)
  float b[2][2][2];
  ld_b<SWZ, CS>(b[0], rows_half, col, 0, ct);
#pragma unroll
  for (int u = 0; u < SM_CP; ++u) {
    // keep this order (fences): left alone, the scheduler sinks every LDS read to just before its
    // MFMA (exposing the LDS latency once per pair) and bunches the refills at the chunk end
    if (u + 1 < SM_CP) ld_b<SWZ, CS>(b[(u + 1) & 1], rows_half, col, u + 1, ct);
    __builtin_amdgcn_sched_barrier(0);
    mm_pair<NTC, NT, CS>(acc, ring[u], b[u & 1]);
    ring_load<NTC>(ring[u], w, p0 + SM_CP + u);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the last np (< 8, possibly 0) pairs of a layer: no refill
template <int NTC, int NT, bool SWZ, bool CS, int MAXP = SM_CP - 1>
__device__ __forceinline__ void span_tail(f32x16 (&acc)[NT][2], float2 (&ring)[SM_CP][NTC > 0 ? NTC : 1],
                                          int np, const float* __restrict__ rows_half, int col, int ct) {
  if (NTC == 0) return;
#pragma unroll
  for (int u = 0; u < MAXP; ++u) {
    if (u < np) {
      float b[2][2];
      ld_b<SWZ, CS>(b, rows_half, col, u, ct);
      mm_pair<NTC, NT, CS>(acc, ring[u], b);
    }
  }
}

// ---- layer-0 input loaders ------------------------------------------------------------------
// LDS scratch shared by the loaders of one workgroup
struct ColInfo {
  int* id;          // [3][64] neighbour index per column (SA: row 0 only)
  float* w;         // [3][64] interpolation weights (SA: row 0 = 1 for valid columns, else 0)
  float* aux;       // [3][64] SA: centre coordinates; FP: row 0 = 1 for valid columns, else 0
};

// P loader: point-major row gather, thread e -> column e>>3, channels 4*(e&7)..+3 of the chunk.
// NB neighbours, weights from LDS.  Issue = loads only; commit = combine + rotated LDS store.
template <int NB, int PIT>
struct PStage {
  float4 v[PIT][NB];
};

template <int NB, int PIT, int NTHR>
__device__ __forceinline__ void p_issue(PStage<NB, PIT>& st, const RowSrc& s, int bi, int cbase /*channel in source*/,
                                        const int* __restrict__ ids /*[3][64] or null = identity*/,
                                        int col0, int id_max, int tid) {
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int e = tid + NTHR * it;
    const int colx = e >> 3, g4 = (e & 7) * 4;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int id = ids ? ids[k * 64 + colx] : min(col0 + colx, id_max);
      st.v[it][k] = *reinterpret_cast<const float4*>(s.tab + ((size_t)bi * s.rows + id) * s.ld + cbase + g4);
    }
  }
}

template <int NB, int PIT, int NTHR>
__device__ __forceinline__ void p_commit(const PStage<NB, PIT>& st, float* __restrict__ cbuf,
                                         const float* __restrict__ ws /*[3][64] weights, LDS*/, int tid) {
#pragma unroll
  for (int it = 0; it < PIT; ++it) {
    const int e = tid + NTHR * it;
    const int colx = e >> 3, g = e & 7;
    float4 r;
    {
      const float w0 = ws[colx];
      r.x = st.v[it][0].x * w0; r.y = st.v[it][0].y * w0; r.z = st.v[it][0].z * w0; r.w = st.v[it][0].w * w0;
    }
#pragma unroll
    for (int k = 1; k < NB; ++k) {       // three_interpolate's order: p0*w0 + p1*w1 + p2*w2, unfused
      const float wk = ws[k * 64 + colx];
      r.x = r.x + st.v[it][k].x * wk; r.y = r.y + st.v[it][k].y * wk;
      r.z = r.z + st.v[it][k].z * wk; r.w = r.w + st.v[it][k].w * wk;
    }
    float* dst = cbuf + (4 * g) * SM_CW + colx + 4 * g;    // rows 4g..4g+3, shifted by 4g columns
    dst[0] = r.x; dst[SM_CW] = r.y; dst[2 * SM_CW] = r.z; dst[3 * SM_CW] = r.w;
  }
}

// C loader (generic, any channel): thread fills column lc = tid&63, rows lr0 + NW*i.
template <bool IS_SA>
__device__ __forceinline__ float c_load(const SaSrc& sa, const FpSrc& fp, const ColInfo& ci, int bi, int lc,
                                        int gcol, bool cvalid, int c) {
  if (!cvalid) return 0.f;
  if (IS_SA) {
    const int id = ci.id[lc];
    if (c < sa.feat.width) return sa.feat.tab[((size_t)bi * sa.feat.rows + id) * sa.feat.ld + c];
    const int k = c - sa.feat.width;
    if (sa.use_xyz && k < 3)     // grouped_xyz -= new_xyz
      return sa.xyz[((size_t)bi * sa.n + id) * 3 + k] - ci.aux[k * 64 + lc];
    return 0.f;
  } else {
    if (c < fp.known.width) {
      const float* t = fp.known.tab + (size_t)bi * fp.known.rows * fp.known.ld + c;
      return t[(size_t)ci.id[lc] * fp.known.ld] * ci.w[lc] + t[(size_t)ci.id[64 + lc] * fp.known.ld] * ci.w[64 + lc] +
             t[(size_t)ci.id[128 + lc] * fp.known.ld] * ci.w[128 + lc];
    }
    const int cu = c - fp.known.width;
    if (cu < fp.unknown.width) return fp.unknown.tab[((size_t)bi * fp.unknown.rows + gcol) * fp.unknown.ld + cu];
    return 0.f;
  }
}

__device__ __forceinline__ void c_store(float* __restrict__ cbuf, int row, int lc, float v) {
  cbuf[row * SM_CW + lc + (row & ~3)] = v;
}

// T loader (layer-0 tail of the specialised kernels, GEN = false): the <= 8 input channels that do
// not fill a 32-channel chunk -- SA: the 3 relative-xyz channels (grouped_xyz - new_xyz); FP: the last
// (width mod 32) <= 8 channels of the unknown points' own features, read with 4-byte loads (the table
// may be a strided view such as pc[..., 3:]).  Thread -> column tid & 63, rows tid>>6 + NW*i < 8.
template <int NW>
struct TStage {
  float v[SM_TAIL_ROWS / NW];
};

// ---- one workgroup's chain ------------------------------------------------------------------
// dynamic LDS: H [hrows][64] overlaid with chunk [2][32][64] | bias [bias_all] | id [3][64] | w [3][64] | aux [3][64]
// (the chunk buffers are dead once layer 0 has been multiplied; H is first written after the barrier
// that follows, so the two share their storage and a workgroup needs max(H, 16 KiB) instead of the sum)
// GEN = true keeps the generic per-element loader for layer-0 inputs of any width / alignment;
// GEN = false (chosen on the host, launch_chain) serves inputs made of whole 32-channel row-gather
// chunks plus at most one T-loader tail -- every shape of PVN3D's backbone -- without carrying the
// generic loader's address arithmetic in registers (it cost the 4-wave kernels 84 / 43 spilled VGPRs).
template <bool IS_SA, int NT, int NW, bool GEN>
struct Chain {
  static constexpr int NTHR = NW * 64;
  static constexpr int LROWS = SM_KC / NW;
  static constexpr int PIT = (SM_KC * SM_COLS / 4) / NTHR;   // float4 per thread per chunk
  static constexpr int NBA = IS_SA ? 1 : 3;                  // neighbours of the first row source
  static constexpr int TR = SM_TAIL_ROWS / NW;               // T-loader values per thread

  const MlpDesc& d;
  const SaSrc& sa;
  const FpSrc& fp;
  float* H;
  float* chunk;
  float* s_bias;
  ColInfo ci;
  int bi, col0, cols_total, tid, lane, wave;

  // chunk kinds: [0, nA) P-gather from source A; [nA, nAB) P-gather from source B (FP unknown);
  // the rest generic (GEN) or one T-loader tail of tail_w channels (!GEN)
  int nA, nAB, n_chunks;
  int tail_w;
#ifdef SM_PROBE
  unsigned long long sm_last_;
#endif

  __device__ __forceinline__ void t_issue(TStage<NW>& st, int tid) const {
    const int lc = tid & 63, r0 = tid >> 6;
    const bool cvalid = col0 + lc < cols_total;
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int r = r0 + NW * i;
      float v = 0.f;
      if (cvalid && r < tail_w) {
        if (IS_SA) v = sa.xyz[((size_t)bi * sa.n + ci.id[lc]) * 3 + r];
        else v = fp.unknown.tab[((size_t)bi * fp.unknown.rows + col0 + lc) * fp.unknown.ld + (nAB - nA) * SM_KC + r];
      }
      st.v[i] = v;
    }
  }
  __device__ __forceinline__ void t_commit(const TStage<NW>& st, float* __restrict__ cbuf, int tid) const {
    const int lc = tid & 63, r0 = tid >> 6;
#pragma unroll
    for (int i = 0; i < TR; ++i) {
      const int r = r0 + NW * i;
      float v = st.v[i];
      if (IS_SA) v = (col0 + lc < cols_total && r < tail_w) ? v - ci.aux[r * 64 + lc] : 0.f;   // grouped_xyz -= new_xyz
      c_store(cbuf, r, lc, v);
    }
  }

  __device__ __forceinline__ const RowSrc& srcA() const { return IS_SA ? sa.feat : fp.known; }

  // tile0: first row tile of this wave (tiles tile0, tile0 + NW, ...); CS: the wave owns only
  // column tile `ct` of row tile tile0 (layers with <= NW/2 row tiles, see run())
  template <int NTC, bool CS>
  __device__ __forceinline__ void layer0(f32x16 (&acc)[NT][2], const WPtr& w, int pairs_total, int tile0, int ct,
                                         int tid, int lane) {
    float2 ring[SM_CP][NTC > 0 ? NTC : 1];
#pragma unroll
    for (int u = 0; u < SM_CP; ++u) ring_load<NTC>(ring[u], w, u);
    const int lc = tid & 63, lr0 = tid >> 6;
    const int gcol = col0 + lc;
    const bool cvalid = gcol < cols_total;
    const int half = lane >> 5, col = lane & 31;
    const int id_max = IS_SA ? 0 : fp.n - 1;

    // ---- prologue: chunk 0 (and, from the first row source, chunk 1: its gathers are kept TWO chunks
    // ahead of the MFMAs -- under load a gather takes 5-6 k cycles (tools/mlp_probe.py) while a chunk of the
    // narrow layers is multiplied in 1-4 k, so one chunk of cover left the layer-0 loop latency-bound)
    PStage<NBA, PIT> sA0, sA1;
    if (nA > 0) {
      p_issue<NBA, PIT, NTHR>(sA0, srcA(), bi, 0, ci.id, col0, id_max, tid);
      if (nA > 1) p_issue<NBA, PIT, NTHR>(sA1, srcA(), bi, SM_KC, ci.id, col0, id_max, tid);
      p_commit<NBA, PIT, NTHR>(sA0, chunk, ci.w, tid);
    } else if (nAB > 0) {
      PStage<1, PIT> st;
      p_issue<1, PIT, NTHR>(st, fp.unknown, bi, 0, nullptr, col0, id_max, tid);
      p_commit<1, PIT, NTHR>(st, chunk, ci.aux, tid);
    } else if (GEN) {
#pragma unroll
      for (int i = 0; i < LROWS; ++i)
        c_store(chunk, lr0 + NW * i, lc, c_load<IS_SA>(sa, fp, ci, bi, lc, gcol, cvalid, lr0 + NW * i));
    } else {
      TStage<NW> st;
      t_issue(st, tid);
      t_commit(st, chunk, tid);
    }
    __syncthreads();
    SM_STAMP(7);
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      acc_bias(acc[t][0], s_bias + (tile0 + NW * t) * 32, half);
      if (!CS) acc[t][1] = acc[t][0];
    }

    // ---- steady state: chunk g-1 is multiplied while chunk g lands and chunk g+1 is requested.
    // Invariant at every step: chunks < g are staged, chunk g (if g < nA) is in flight in sA1 / sA0
    // alternately.  Unrolled by two so that the two register stages are addressed statically and every
    // load of the loop body is unconditional (the compiler then counts vmcnt exactly).
#define SM_SPAN_PREV(G) \
  span8<NTC, NT, true, CS>(acc, ring, w, ((G) - 1) * SM_CP, chunk + (((G) - 1) & 1) * SM_KC * SM_CW + half * SM_CW, col, ct)
    int g = 1;
    for (; g + 2 < nA; g += 2) {
      p_issue<NBA, PIT, NTHR>(sA0, srcA(), bi, (g + 1) * SM_KC, ci.id, col0, id_max, tid);
      SM_SPAN_PREV(g);
      p_commit<NBA, PIT, NTHR>(sA1, chunk + (g & 1) * SM_KC * SM_CW, ci.w, tid);
      __syncthreads();
      p_issue<NBA, PIT, NTHR>(sA1, srcA(), bi, (g + 2) * SM_KC, ci.id, col0, id_max, tid);
      SM_SPAN_PREV(g + 1);
      p_commit<NBA, PIT, NTHR>(sA0, chunk + ((g + 1) & 1) * SM_KC * SM_CW, ci.w, tid);
      __syncthreads();
    }
    if (g + 1 < nA) {            // two left: g in flight (sA1), g + 1 still to request
      p_issue<NBA, PIT, NTHR>(sA0, srcA(), bi, (g + 1) * SM_KC, ci.id, col0, id_max, tid);
      SM_SPAN_PREV(g);
      p_commit<NBA, PIT, NTHR>(sA1, chunk + (g & 1) * SM_KC * SM_CW, ci.w, tid);
      __syncthreads();
      SM_SPAN_PREV(g + 1);
      p_commit<NBA, PIT, NTHR>(sA0, chunk + ((g + 1) & 1) * SM_KC * SM_CW, ci.w, tid);
      __syncthreads();
      g += 2;
    } else if (g < nA) {         // one left, in flight (sA1)
      SM_SPAN_PREV(g);
      p_commit<NBA, PIT, NTHR>(sA1, chunk + (g & 1) * SM_KC * SM_CW, ci.w, tid);
      __syncthreads();
      g += 1;
    }
#undef SM_SPAN_PREV
    if (!IS_SA) {
      for (; g < nAB; ++g) {
        PStage<1, PIT> st;
        p_issue<1, PIT, NTHR>(st, fp.unknown, bi, (g - nA) * SM_KC, nullptr, col0, id_max, tid);
        span8<NTC, NT, true, CS>(acc, ring, w, (g - 1) * SM_CP, chunk + ((g - 1) & 1) * SM_KC * SM_CW + half * SM_CW, col, ct);
        p_commit<1, PIT, NTHR>(st, chunk + (g & 1) * SM_KC * SM_CW, ci.aux, tid);
        __syncthreads();
      }
    }
    if (GEN) {
      for (; g < n_chunks; ++g) {
        float stage[LROWS];
#pragma unroll
        for (int i = 0; i < LROWS; ++i)
          stage[i] = c_load<IS_SA>(sa, fp, ci, bi, lc, gcol, cvalid, g * SM_KC + lr0 + NW * i);
        span8<NTC, NT, true, CS>(acc, ring, w, (g - 1) * SM_CP, chunk + ((g - 1) & 1) * SM_KC * SM_CW + half * SM_CW, col, ct);
#pragma unroll
        for (int i = 0; i < LROWS; ++i) c_store(chunk + (g & 1) * SM_KC * SM_CW, lr0 + NW * i, lc, stage[i]);
        __syncthreads();
      }
    } else if (g < n_chunks) {          // the T-loader tail (always the last chunk)
      TStage<NW> st;
      t_issue(st, tid);
      span8<NTC, NT, true, CS>(acc, ring, w, (g - 1) * SM_CP, chunk + ((g - 1) & 1) * SM_KC * SM_CW + half * SM_CW, col, ct);
      t_commit(st, chunk + (g & 1) * SM_KC * SM_CW, tid);
      __syncthreads();
    }
    // ---- last chunk
    SM_STAMP(8);
    const int p0 = (n_chunks - 1) * SM_CP;
    const int np = pairs_total - p0;       // 1..8
    const float* rows_half = chunk + ((n_chunks - 1) & 1) * SM_KC * SM_CW + half * SM_CW;
    if (np == SM_CP)
      span8<NTC, NT, true, CS>(acc, ring, w, p0, rows_half, col, ct);
    else     // the specialised kernels' tail holds <= SM_TAIL_ROWS channels
      span_tail<NTC, NT, true, CS, (GEN ? SM_CP - 1 : SM_TAIL_ROWS / 4)>(acc, ring, np, rows_half, col, ct);
  }

  template <int NTC, bool CS>
  __device__ __forceinline__ void layerN(f32x16 (&acc)[NT][2], const WPtr& w, int pairs_total, int boff, int tile0,
                                         int ct, int lane) {
    float2 ring[SM_CP][NTC > 0 ? NTC : 1];
#pragma unroll
    for (int u = 0; u < SM_CP; ++u) ring_load<NTC>(ring[u], w, u);
    const int half = lane >> 5, col = lane & 31;
#pragma unroll
    for (int t = 0; t < NTC; ++t) {
      acc_bias(acc[t][0], s_bias + boff + (tile0 + NW * t) * 32, half);
      if (!CS) acc[t][1] = acc[t][0];
    }
    const float* rows_half = H + half * SM_COLS;
    int p0 = 0;
    for (; p0 + SM_CP <= pairs_total; p0 += SM_CP)
      span8<NTC, NT, false, CS>(acc, ring, w, p0, rows_half + (size_t)p0 * 4 * SM_COLS, col, ct);
    span_tail<NTC, NT, false, CS>(acc, ring, pairs_total - p0, rows_half + (size_t)p0 * 4 * SM_COLS, col, ct);
  }

  __device__ __forceinline__ void run(const OutDesc& od) {
    f32x16 acc[NT][2];
    int boff = 0;
    SM_STAMP(0);
    for (int l = 0; l < d.n_layers; ++l) {
      // Per-layer opaque copy of the thread id: every LDS / global address of the layer is derived from
      // it, so none of that arithmetic can be hoisted out of the layer loop and kept alive (or spilled)
      // across the other layers' MFMA loops; recomputing it costs a few dozen VALU per layer.
      int tid = this->tid;
      asm volatile("" : "+v"(tid));
      const int lane = tid & 63;
      const int K = d.K[l], M = d.M[l];
      const int mt_total = (M + 31) >> 5;
      // Column split: a non-final layer with <= NW/2 row tiles would leave half of the waves
      // without MFMA work; there wave u takes (row tile u % mt_total, column tile u / mt_total).
      // (one-tile SA kernels only: the two-tile kernels serve M > 128, and every extra
      // instantiation of the layer code in one kernel costs registers -- the FP level-0 kernel got
      // 7 % slower from the additional spills alone)
      const bool cs = (NT == 1) && IS_SA && (2 * mt_total <= NW) && (l + 1 < d.n_layers);
      const int tile0 = cs ? wave % mt_total : wave;
      const int ct = cs ? wave / mt_total : 0;
      const int nt = cs ? (ct < 2 ? 1 : 0) : (mt_total - wave + NW - 1) / NW;   // row tiles of this wave (<= NT)
      const int pairs_total = (K + 3) >> 2;
      WPtr w;
      w.wp = reinterpret_cast<const float2*>(d.W[l]) + (size_t)tile0 * 64 + lane;
      w.pstride = (size_t)mt_total * 64;
      w.tstride = (size_t)NW * 64;
      w.last = pairs_total - 1;
      if (l == 0) {
        if (cs) {
          if (nt) layer0<1, true>(acc, w, pairs_total, tile0, ct, tid, lane);
          else layer0<0, false>(acc, w, pairs_total, tile0, ct, tid, lane);
        } else if (nt >= NT) layer0<NT, false>(acc, w, pairs_total, tile0, ct, tid, lane);
        else if (NT > 1 && nt == NT - 1) layer0<(NT > 1 ? NT - 1 : 0), false>(acc, w, pairs_total, tile0, ct, tid, lane);
        else layer0<0, false>(acc, w, pairs_total, tile0, ct, tid, lane);
      } else {
        if (cs) {
          if (nt) layerN<1, true>(acc, w, pairs_total, boff, tile0, ct, lane);
        } else if (nt >= NT) layerN<NT, false>(acc, w, pairs_total, boff, tile0, ct, lane);
        else if (NT > 1 && nt == NT - 1) layerN<(NT > 1 ? NT - 1 : 0), false>(acc, w, pairs_total, boff, tile0, ct, lane);
      }
      __syncthreads();   // every wave has finished reading this layer's input
      SM_STAMP(1 + 2 * l);
      boff += mt_total * 32;
      if (l + 1 < d.n_layers) {
        // rows [M, roundup32(M)) come out as relu(0 + 0) = 0 (zero-padded weights and bias), which
        // covers the next layer's K rounded up to a multiple of 4
        const int half = lane >> 5, col = lane & 31;
        if (cs) {
          if (nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = tile0 * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              H[row * SM_COLS + ct * 32 + col] = fmaxf(acc[0][0][r], 0.f);
            }
          }
        } else {
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
        __syncthreads();
        SM_STAMP(2 + 2 * l);
      }
    }

    // ---- epilogue on the last layer's accumulators
    int tid = this->tid;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int L = d.n_layers - 1;
    const int M = d.M[L];
    const int mt_total = (M + 31) >> 5;
    const int nt = (mt_total - wave + NW - 1) / NW;
    const int half = lane >> 5, col = lane & 31;
    float* const out = od.out;
    if (IS_SA) {
      // Max-pool through LDS: the wave parks relu(tile) in a private [32][SM_EPAD] patch of the
      // (now free) activation buffer, then lane (row = lane&31, column half = lane>>5) reads
      // its 32 columns with 8 x 16-byte loads and reduces segments of nsample in registers.
      // Lanes 0..31 then hold 32 consecutive channels of a centre: one 128-byte store per
      // centre into the point-major output (a DPP butterfly per accumulator register plus
      // 4-byte scattered stores cost a quarter of a narrow chain's workgroup time).
      float* sc = H + (size_t)wave * (32 * SM_EPAD);
      const int ns = sa.ns;
      const int nout = ns >= 32 ? 1 : 32 / ns;                  // centres per lane
      const int jbase = ns >= 64 ? col0 / ns : (col0 + 32 * half) / ns;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) {
          const int mt = wave + NW * t;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
            sc[rr * SM_EPAD + col] = fmaxf(acc[t][0][r], 0.f);
            sc[rr * SM_EPAD + 32 + col] = fmaxf(acc[t][1][r], 0.f);
          }
          __builtin_amdgcn_wave_barrier();       // same wave: LDS operations complete in order
          int v[32];                              // post-ReLU floats compare like signed ints
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int4 q = *reinterpret_cast<const int4*>(sc + col * SM_EPAD + 32 * half + 4 * i);
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
          if (ns >= 16) {
            v[0] = max(v[0], v[1]);
            v[1] = max(v[2], v[3]);
          }
          if (ns >= 32) v[0] = max(v[0], v[1]);
          if (ns >= 64) v[0] = max(v[0], __shfl_xor(v[0], 32, 64));
          const int row = mt * 32 + col;
          if (row < M && (ns < 64 || half == 0)) {
            float* o = out + ((size_t)bi * sa.m + jbase) * od.ld + od.coff + row;
            if (nout <= 2) {
              if (jbase < sa.m) o[0] = __int_as_float(v[0]);
              if (nout == 2 && jbase + 1 < sa.m) o[od.ld] = __int_as_float(v[1]);
            } else {
#pragma unroll
              for (int q = 0; q < 32; ++q)
                if (q < nout && jbase + q < sa.m) o[(size_t)q * od.ld] = __int_as_float(v[q]);
            }
          }
        }
      }
      SM_STAMP(15);
      return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
        const int mt = wave + NW * t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = mt * 32 + 8 * g + 4 * half;      // regs 4g..4g+3 = rows row..row+3
          float v0[4], v1[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            v0[k] = fmaxf(acc[t][0][4 * g + k], 0.f);
            v1[k] = fmaxf(acc[t][1][4 * g + k], 0.f);
          }
          {
            const int g0 = col0 + col, g1 = col0 + 32 + col;
            if (od.point_major) {
              float* o0 = out + ((size_t)bi * fp.n + g0) * od.ld + od.coff + row;
              float* o1 = out + ((size_t)bi * fp.n + g1) * od.ld + od.coff + row;
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (row + k < M) {
                  if (g0 < cols_total) o0[k] = v0[k];
                  if (g1 < cols_total) o1[k] = v1[k];
                }
              }
            } else {
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                if (row + k < M) {
                  if (g0 < cols_total) out[((size_t)bi * M + row + k) * fp.n + g0] = v0[k];
                  if (g1 < cols_total) out[((size_t)bi * M + row + k) * fp.n + g1] = v1[k];
                }
              }
            }
          }
        }
      }
    }
    SM_STAMP(15);
  }
};

__device__ __forceinline__ bool row_src_vec_ok(const RowSrc& s) {
  return s.tab != nullptr && (s.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(s.tab) & 15) == 0;
}

template <bool IS_SA, int NT, int NW, bool GEN>
__device__ __forceinline__ void mlp_chain_body(const MlpDesc& d, const SaSrc& sa, const FpSrc& fp, int hrows,
                                               int bias_all, int cols_total, const OutDesc& od) {
  extern __shared__ float s_mem[];
  float* H = s_mem;
  float* chunk = H;                                   // overlaid (hrows * 64 >= 2 * SM_KC * SM_CW, launch_chain)
  float* s_bias = H + (size_t)hrows * SM_COLS;
  ColInfo ci;
  ci.id = reinterpret_cast<int*>(s_bias + bias_all);
  ci.w = reinterpret_cast<float*>(ci.id + 3 * 64);
  ci.aux = ci.w + 3 * 64;
  const int tid = threadIdx.x, lane = tid & 63;
  // scalar wave index: keeps every "does this wave own row tile t" test a uniform branch
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bi, bx;
  xcd_frame_map(bi, bx);
  const int col0 = bx * SM_COLS;

  SM_PROBE_DECL;
  SM_STAMP(14);
  stage_bias<NW * 64>(d, s_bias, tid);
  if (tid < 64) {     // per-column gather info
    const int gcol = col0 + tid;
    const bool cvalid = gcol < cols_total;
    if (IS_SA) {
      int id = 0;
      float cx = 0.f, cy = 0.f, cz = 0.f;
      if (cvalid) {
        id = sa.idx[(size_t)bi * sa.m * sa.ns + gcol];
        const float* c = sa.new_xyz + ((size_t)bi * sa.m + gcol / sa.ns) * 3;
        cx = c[0]; cy = c[1]; cz = c[2];
      }
      ci.id[tid] = id;
      ci.w[tid] = cvalid ? 1.f : 0.f;
      ci.aux[tid] = cx; ci.aux[64 + tid] = cy; ci.aux[128 + tid] = cz;
    } else {
      int i0 = 0, i1 = 0, i2 = 0;
      float w0 = 0.f, w1 = 0.f, w2 = 0.f;
      if (cvalid) {
        const int* ip = fp.idx + ((size_t)bi * fp.n + gcol) * 3;
        const float* wp = fp.weight + ((size_t)bi * fp.n + gcol) * 3;
        i0 = ip[0]; i1 = ip[1]; i2 = ip[2];
        w0 = wp[0]; w1 = wp[1]; w2 = wp[2];
      }
      ci.id[tid] = i0; ci.id[64 + tid] = i1; ci.id[128 + tid] = i2;
      ci.w[tid] = w0; ci.w[64 + tid] = w1; ci.w[128 + tid] = w2;
      ci.aux[tid] = cvalid ? 1.f : 0.f;       // "weight" of the unknown-feature rows
    }
  }
  __syncthreads();

  Chain<IS_SA, NT, NW, GEN> ch{d, sa, fp, H, chunk, s_bias, ci, bi, col0, cols_total, tid, lane, wave, 0, 0, 0, 0};
  ch.n_chunks = (d.K[0] + SM_KC - 1) / SM_KC;
  if (IS_SA) {
    ch.nA = row_src_vec_ok(sa.feat) ? sa.feat.width / SM_KC : 0;
    ch.nAB = ch.nA;
    ch.tail_w = sa.use_xyz ? 3 : 0;
  } else {
    ch.nA = row_src_vec_ok(fp.known) ? fp.known.width / SM_KC : 0;
    ch.nAB = ch.nA;
    if (ch.nA * SM_KC == fp.known.width && row_src_vec_ok(fp.unknown))
      ch.nAB = ch.nA + fp.unknown.width / SM_KC;
    ch.tail_w = fp.unknown.width - (ch.nAB - ch.nA) * SM_KC;
  }
#ifdef SM_PROBE
  ch.sm_last_ = sm_last_;
#endif
  ch.run(od);
}

// host mirror of the device-side chunk classification: can the specialised (GEN = false) kernels
// serve this layer-0 input?
bool host_vec_ok(const RowSrc& s) {
  return s.tab != nullptr && (s.ld & 3) == 0 && (reinterpret_cast<uintptr_t>(s.tab) & 15) == 0;
}
template <bool IS_SA>
bool specialised_ok(const SaSrc& sa, const FpSrc& fp) {
  if (IS_SA) return sa.feat.width % SM_KC == 0 && (sa.feat.width == 0 || host_vec_ok(sa.feat));
  if (fp.known.width % SM_KC != 0 || !host_vec_ok(fp.known)) return false;
  const int full = fp.unknown.width / SM_KC, rest = fp.unknown.width % SM_KC;
  // (a strided unknown table with >= 32 channels falls to the generic kernel: the device would
  // classify its whole chunks as generic too)
  return rest <= SM_TAIL_ROWS && (full == 0 || host_vec_ok(fp.unknown));
}

// One row tile per wave, 4 waves (M <= 128).  (The generic-loader variant is allowed the registers of
// two waves per SIMD: with three it spills.)
template <bool IS_SA, bool GEN>
__global__ __launch_bounds__(256, (GEN ? 2 : 3)) void mlp_chain_kernel(
    MlpDesc d, SaSrc sa, FpSrc fp, int hrows, int bias_all, int cols_total, OutDesc od) {
  mlp_chain_body<IS_SA, 1, 4, GEN>(d, sa, fp, hrows, bias_all, cols_total, od);
}

// Two row tiles per wave, 4 waves (128 < M <= 256): every wave owns a tile of a 128-wide layer,
// where an 8-wave workgroup would leave half of its waves without MFMA work.
template <bool IS_SA, bool GEN>
__global__ __launch_bounds__(256, 2) void mlp_chain_mid_kernel(
    MlpDesc d, SaSrc sa, FpSrc fp, int hrows, int bias_all, int cols_total, OutDesc od) {
  mlp_chain_body<IS_SA, 2, 4, GEN>(d, sa, fp, hrows, bias_all, cols_total, od);
}

// Two row tiles per wave, 8 waves (M > 256).
template <bool IS_SA, bool GEN>
__global__ __launch_bounds__(512, 2) void mlp_chain_wide_kernel(
    MlpDesc d, SaSrc sa, FpSrc fp, int hrows, int bias_all, int cols_total, OutDesc od) {
  mlp_chain_body<IS_SA, 2, 8, GEN>(d, sa, fp, hrows, bias_all, cols_total, od);
}

// ---------------------------------------------------------------------------------------
// Narrow chains (every M <= 64): column-sliced variant.  A wave owns 32 columns and ALL row
// tiles, keeps its own slice of the activations in LDS and gathers its own input chunk, so
// there is no workgroup barrier after the bias staging -- with M <= 64 the row-split kernel
// above leaves 2-3 of its 4 waves without a row tile.  Workgroup = 2 waves = 64 columns.
// dynamic LDS: bias [bias_floats] | per wave: H [hrows][32] | chunk [32][32].
// ---------------------------------------------------------------------------------------
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

template <bool IS_SA, int NTR>
__global__ __launch_bounds__(128, (!IS_SA ? 1 : (NTR <= 2 ? 4 : 2))) void mlp_chain_cols_kernel(
    MlpDesc d, SaSrc sa, FpSrc fp, int hrows, int bias_floats, int cols_total, OutDesc od) {
  extern __shared__ float s_mem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* s_bias = s_mem;
  float* H = s_mem + bias_floats + (size_t)wave * (hrows + SM_KC) * 32;
  float* chunk = H + (size_t)hrows * 32;
  int bi, bx;
  xcd_frame_map(bi, bx);
  const int col0 = bx * 64 + wave * 32;
  SM_PROBE_DECL;
  SM_STAMP(14);
  const int half = lane >> 5, col = lane & 31;

  // loader: lane fills column `col`, rows half + 2*i (i < 16) of every 32-row chunk.  The index /
  // weight loads are issued BEFORE the bias staging barrier so that the two latencies overlap.
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
  stage_bias<128>(d, s_bias, tid);
  __syncthreads();                         // the only barrier of this kernel
  if (col0 >= cols_total) return;          // wave-uniform
  SM_STAMP(0);
  // plain local copies: capturing the by-value kernel-argument structs by reference would pin
  // them in scratch memory and turn every field access of the loader into a scratch load
  const float* const sa_xyz = sa.xyz; const float* const sa_nxyz = sa.new_xyz;
  const float* const sa_row = sa.feat.tab ? sa.feat.tab + ((size_t)bi * sa.feat.rows + id0) * sa.feat.ld : nullptr;
  const int sa_n = sa.n, sa_C = sa.feat.width, sa_c3 = sa.use_xyz ? 3 : 0;
  const float* const kf = fp.known.tab ? fp.known.tab + (size_t)bi * fp.known.rows * fp.known.ld : nullptr;
  const float* const uf_row =
      fp.unknown.tab ? fp.unknown.tab + ((size_t)bi * fp.unknown.rows + min(gcol, fp.n - 1)) * fp.unknown.ld : nullptr;
  const int k_ld = fp.known.ld, fp_C2 = fp.known.width, fp_C1 = fp.unknown.width;
  auto load_input = [=](int c) -> float {
    if (!cvalid) return 0.f;
    if (IS_SA) {
      if (c < sa_C) return sa_row[c];
      const int k = c - sa_C;
      // grouped_xyz -= new_xyz (the centre is re-read: a select chain over three registers
      // is turned into a scratch-array lookup by the compiler)
      return k < sa_c3 ? sa_xyz[((size_t)bi * sa_n + id0) * 3 + k] - sa_nxyz[ctr + k] : 0.f;
    } else {
      if (c < fp_C2)
        return kf[(size_t)id0 * k_ld + c] * w0 + kf[(size_t)id1 * k_ld + c] * w1 + kf[(size_t)id2 * k_ld + c] * w2;
      const int cu = c - fp_C2;
      return cu < fp_C1 ? uf_row[cu] : 0.f;
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
        if (ch == 0) SM_STAMP(6);
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
    SM_STAMP(1 + 2 * l);
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
  float* const out = od.out;
  if (IS_SA) {
    // Max-pool through the wave's own (now free) H | chunk patch, as in the row-split kernel: park
    // relu(tile) as [32 rows][36], lane (row = lane&31, half) reads its 16 columns with four
    // 16-byte loads, reduces segments of nsample in registers, and lanes 0..31 store 32
    // consecutive channels of a centre (128 bytes).  The DPP butterfly + 4-byte scattered stores
    // this replaces took 30 % of a level-0 wave's time (SM_PROBE).
    constexpr int EP = 36;
    float* sc = H;                               // (hrows + 32) * 32 >= 32 * 36 floats (hrows >= 4)
    const int ns = sa.ns;                        // 1..32, power of two
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      if (t < mt_total) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = (r & 3) + 8 * (r >> 2) + 4 * half;
          sc[rr * EP + col] = fmaxf(acc[t][0][r], 0.f);
        }
        __builtin_amdgcn_wave_barrier();         // same wave: LDS operations complete in order
        int v[16];                               // post-ReLU floats compare like signed ints
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int4 q = *reinterpret_cast<const int4*>(sc + col * EP + 16 * half + 4 * i);
          v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        }
        __builtin_amdgcn_wave_barrier();
        if (ns >= 2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = max(v[2 * i], v[2 * i + 1]);
        }
        if (ns >= 4) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = max(v[2 * i], v[2 * i + 1]);
        }
        if (ns >= 8) { v[0] = max(v[0], v[1]); v[1] = max(v[2], v[3]); }
        if (ns >= 16) v[0] = max(v[0], v[1]);
        if (ns >= 32) v[0] = max(v[0], __shfl_xor(v[0], 32, 64));
        const int nout = ns >= 16 ? 1 : 16 / ns;                 // centres per lane
        const int jbase = ns >= 32 ? col0 / ns : (col0 + 16 * half) / ns;
        const int row = t * 32 + col;
        if (row < M && (ns < 32 || half == 0)) {
          float* o = out + ((size_t)bi * sa.m + jbase) * od.ld + od.coff + row;
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (q < nout && jbase + q < sa.m) o[(size_t)q * od.ld] = __int_as_float(v[q]);
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NTR; ++t) {
      if (t < mt_total) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const float v = fmaxf(acc[t][0][r], 0.f);
          if (row < M && gcol < cols_total) {
            if (od.point_major) out[((size_t)bi * fp.n + gcol) * od.ld + od.coff + row] = v;
            else out[((size_t)bi * M + row) * fp.n + gcol] = v;
          }
        }
      }
    }
  }
  SM_STAMP(15);
}

template <bool IS_SA>
int launch_chain(const MlpDesc& d, const SaSrc& sa, const FpSrc& fp, int b, int cols_total,
                 const OutDesc& od, hipStream_t st) {
  int hrows = 4, max_mt = 1, bias_all = 0;
  for (int l = 0; l < d.n_layers; ++l) {
    if (l + 1 < d.n_layers) hrows = max(hrows, ((d.M[l] + 31) / 32) * 32);
    max_mt = max(max_mt, (d.M[l] + 31) / 32);
    bias_all += ((d.M[l] + 31) / 32) * 32;
  }
  if (max_mt > SM_MAX_MT) return (int)hipErrorInvalidValue;
  const bool ns_ok = !IS_SA || sa.ns <= 32;
  // Measured (MI355X, 64 frames): the column-sliced kernel wins for M <= 64 (level 0) and
  // loses for M = 128 with K >= 99 (every wave re-fetches all weight fragments), so it is
  // used for <= 2 row tiles only.
  if (max_mt <= 2 && ns_ok) {   // narrow chain: column-sliced kernel
    const size_t lds2 = ((size_t)2 * (hrows + SM_KC) * 32 + bias_all) * sizeof(float);
    if (lds2 > 160 * 1024) return (int)hipErrorInvalidValue;
    const dim3 grid2(pvn3d_ceil_div(cols_total, 64), b);
    if (max_mt <= 1) {
      PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(mlp_chain_cols_kernel<IS_SA, 1>));
      hipLaunchKernelGGL((mlp_chain_cols_kernel<IS_SA, 1>), grid2, dim3(128), lds2, st, d, sa, fp, hrows, bias_all,
                         cols_total, od);
    } else {
      PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(mlp_chain_cols_kernel<IS_SA, 2>));
      hipLaunchKernelGGL((mlp_chain_cols_kernel<IS_SA, 2>), grid2, dim3(128), lds2, st, d, sa, fp, hrows, bias_all,
                         cols_total, od);
    }
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid(pvn3d_ceil_div(cols_total, SM_COLS), b);
  const bool spec = specialised_ok<IS_SA>(sa, fp);
  // H is overlaid with the two layer-0 chunk buffers and, in the SA epilogue, with NW wave-private
  // [32][SM_EPAD] max-pool patches
#define SM_LAUNCH(KERN, NW)                                                                                   \
  do {                                                                                                        \
    int hr = max(hrows, pvn3d_ceil_div(2 * SM_KC * SM_CW, SM_COLS)); \
    if (IS_SA) hr = max(hr, pvn3d_ceil_div(NW * 32 * SM_EPAD, SM_COLS));                                      \
    const size_t lds = ((size_t)hr * SM_COLS + bias_all + 9 * 64) * sizeof(float);                            \
    if (lds > 160 * 1024) return (int)hipErrorInvalidValue;                                                   \
    if (spec) {                                                                                               \
      PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(KERN<IS_SA, false>));                                     \
      hipLaunchKernelGGL((KERN<IS_SA, false>), grid, dim3(NW * 64), lds, st, d, sa, fp, hr, bias_all, cols_total, od); \
    } else {                                                                                                  \
      PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(KERN<IS_SA, true>));                                      \
      hipLaunchKernelGGL((KERN<IS_SA, true>), grid, dim3(NW * 64), lds, st, d, sa, fp, hr, bias_all, cols_total, od); \
    }                                                                                                         \
  } while (0)
  // <= 4 row tiles: 4 waves, one row tile each; <= 8: 4 waves, two row tiles each; wider: 8 waves
  // (two per SIMD hide each other's L2 / LDS waits), two row tiles each
  if (max_mt <= 4) SM_LAUNCH(mlp_chain_kernel, 4);
  else if (max_mt <= 8) SM_LAUNCH(mlp_chain_mid_kernel, 4);
  else SM_LAUNCH(mlp_chain_wide_kernel, 8);
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

// (B, C, n) -> (B, n, ld) tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void transpose_cn_kernel(int c, int n, const float* __restrict__ in,
                                                           float* __restrict__ out, int ld_out) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* src = in + (size_t)b * c * n;
  float* dst = out + (size_t)b * n * ld_out;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int cc = c0 + ty + i, nn = n0 + tx;
    tile[ty + i][tx] = (cc < c && nn < n) ? src[(size_t)cc * n + nn] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int nn = n0 + ty + i, cc = c0 + tx;
    if (nn < n && cc < c) dst[(size_t)nn * ld_out + cc] = tile[tx][ty + i];
  }
}

}  // namespace

extern "C" int pvn3d_sa_mlp_maxpool(int b, int n, int m, int c, int nsample, int use_xyz,
                                    const float* xyz, const float* new_xyz,
                                    const float* features_pm, int ld_feat, const int* idx,
                                    int n_layers, const int* dims_host,
                                    const float* const* w_packed, const float* const* bias_padded,
                                    float* out_pm, int ld_out, int out_coff, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (nsample <= 0 || (nsample & (nsample - 1)) || nsample > 64 || !xyz || !new_xyz || !idx ||
      !out_pm || !dims_host || !w_packed || !bias_padded)
    return (int)hipErrorInvalidValue;
  MlpDesc d;
  if (!fill_desc(&d, n_layers, dims_host, w_packed, bias_padded)) return (int)hipErrorInvalidValue;
  const int cf = features_pm ? c : 0;
  if (dims_host[0] != (use_xyz ? 3 : 0) + cf) return (int)hipErrorInvalidValue;
  if (cf > 0 && ld_feat < cf) return (int)hipErrorInvalidValue;
  if (out_coff < 0 || ld_out < out_coff + dims_host[n_layers]) return (int)hipErrorInvalidValue;
  SaSrc sa = {xyz, new_xyz, {cf > 0 ? features_pm : nullptr, n, ld_feat, cf}, idx, n, m, nsample, use_xyz};
  FpSrc fp = {};
  OutDesc od = {out_pm, 1, ld_out, out_coff};
  return launch_chain<true>(d, sa, fp, b, m * nsample, od, (hipStream_t)stream);
}

extern "C" int pvn3d_fp_interp_mlp(int b, int n, int m, int c2, int c1, const float* known_pm,
                                   int ld_known, const float* unknown_pm, int ld_unknown,
                                   const int* idx, const float* weight, int n_layers,
                                   const int* dims_host, const float* const* w_packed,
                                   const float* const* bias_padded, float* out,
                                   int out_point_major, int ld_out, void* stream) {
  if (b <= 0 || n <= 0) return 0;
  if (!known_pm || !idx || !weight || !out || !dims_host || !w_packed || !bias_padded ||
      (c1 > 0 && !unknown_pm) || ld_known < c2 || (c1 > 0 && ld_unknown < c1))
    return (int)hipErrorInvalidValue;
  MlpDesc d;
  if (!fill_desc(&d, n_layers, dims_host, w_packed, bias_padded)) return (int)hipErrorInvalidValue;
  if (dims_host[0] != c2 + c1) return (int)hipErrorInvalidValue;
  if (out_point_major && ld_out < dims_host[n_layers]) return (int)hipErrorInvalidValue;
  SaSrc sa = {};
  FpSrc fp = {{known_pm, m, ld_known, c2}, {c1 > 0 ? unknown_pm : nullptr, n, ld_unknown, c1}, idx, weight, n, m};
  OutDesc od = {out, out_point_major ? 1 : 0, ld_out, 0};
  return launch_chain<false>(d, sa, fp, b, n, od, (hipStream_t)stream);
}

extern "C" int pvn3d_transpose_bcn_to_bnc(int b, int c, int n, const float* in, float* out, int ld_out,
                                          void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  if (!in || !out || ld_out < c) return (int)hipErrorInvalidValue;
  const dim3 grid(pvn3d_ceil_div(n, 32), pvn3d_ceil_div(c, 32), b);
  hipLaunchKernelGGL(transpose_cn_kernel, grid, dim3(256), 0, (hipStream_t)stream, c, n, in, out, ld_out);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
