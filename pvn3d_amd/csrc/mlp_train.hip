// mlp_train.hip -- the training-mode grouped-point x MLP-weight contraction on bf16 MFMA (gfx950).
//
// Replaces, for the training step (BASELINE config 5), what the reference runs through cuDNN / ATen:
//   SharedMLP = [Conv2d 1x1 (bias-free) -> BatchNorm2d (batch statistics) -> ReLU] x L
//   (pvn3d/lib/utils/etw_pytorch_utils/pytorch_utils.py:25-50, 80-134) on the grouped tensor, then
//   F.max_pool2d over nsample (pvn3d/lib/pointnet2_utils/pointnet2_modules.py:58-71), and the FP modules'
//   interpolate -> concat -> SharedMLP (:188-206), forward AND backward.
//
// Layout: every activation is a POINT-MAJOR bf16 matrix [rows][ld] (row = one (cloud, centre, sample) column of
// the reference's (B, C, npoint, nsample) tensor, ld = channels rounded up to 16, pad columns zero).  A 1x1
// convolution is then a plain GEMM with K contiguous in both operands, a neighbour gather is one contiguous
// row, and nothing is ever transposed to NCHW.  Per layer:
//   forward : Y = H_prev . W^T            mt_gemm_nt (v_mfma_f32_32x32x16_bf16, fp32 accumulate), whose epilogue
//             also emits per-channel partial sums of y and y^2 (BatchNorm statistics, deterministic two-stage)
//             -> mt_bn_finalize (mean, 1/std, folded scale / shift, running statistics)
//             -> mt_bn_relu_apply (H = relu(a y + b))  [-> mt_pool_max with arg-indices after the last layer]
//   backward: dz = dH . [H > 0];  sums of dz and dz.yhat (mt_bn_bwd_reduce -> mt_bn_bwd_finalize: dgamma, dbeta)
//             dY = a dz + k1 y + k0  (mt_bn_bwd_apply: the whole BatchNorm backward is affine in dz and y)
//             dH_prev = dY . W       (mt_gemm_nt against the transposed weights)
//             dW = dY^T . H_prev     (mt_gemm_nt on transposed copies, K = rows, split-K with fp32 atomics)
// plus the gathers that build layer 0's input (SA: relative xyz ++ neighbour features; FP: three_interpolate ++
// skip features); their backwards hand a channel-major fp32 copy (mt_unpack_cm) to the row-owner scatter kernels of
// group_points.hip / interpolate.hip (global fp32 atomics measured 6x slower: 20 ms of a 50 ms step).
#include "common.h"

namespace {

typedef unsigned short bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16_t f2bf(float f) {      // round to nearest even: v_cvt_pk_bf16_f32
  return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
// two fp32 -> packed bf16 pair, round to nearest even: one v_cvt_pk_bf16_f32 (the software form is five VALU
// instructions per value, and the epilogues / elementwise kernels here are bound by VALU issue)
typedef float mt_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 mt_bf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  const mt_f2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, mt_bf2));
}
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}

// H > 0 without reading H: H = bf16(relu(a y + b)) (mt_bn_relu_apply), so the mask is recomputed from y -- the
// backward passes stream two matrices instead of three
// (a positive fp32 a y + b rounds to a positive bf16 unless it is a subnormal below 2^-134, where the derivative of
// the ReLU is 1 anyway)
__device__ __forceinline__ bool relu_on(float a, float y, float b) { return fmaf(a, y, b) > 0.f; }
// thread index -> (row, 8-channel chunk) and row -> (group, sample) in 32-bit arithmetic (the entry points refuse
// launches beyond 2^31 chunks; a 64-bit division costs more than the rest of these kernels)
__device__ __forceinline__ void split_idx(long long t, int cpr, long long& row, int& c0) {
  const unsigned u = (unsigned)t, r = u / (unsigned)cpr;
  row = r;
  c0 = (int)(u - r * (unsigned)cpr) * 8;
}
__device__ __forceinline__ void split_row(long long row, int ns, long long& g, int& s) {
  const unsigned u = (unsigned)row, q = u / (unsigned)ns;
  g = q;
  s = (int)(u - q * (unsigned)ns);
}
__device__ __forceinline__ void load8f(const float* __restrict__ p, float (&v)[8]) {
  const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
  v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
}

// ------------------------------------------------------------------------------------------------ GEMM
// C[M][N] = A[M][K] . B[N][K]^T, A / B bf16 with K contiguous (lda, ldb in elements, multiples of 8; K a multiple
// of 16; rows of B beyond N and rows of A beyond M read as zero).  Workgroup tile 128 x NT, four waves of 32 rows
// each, K in chunks of 32 through double-buffered LDS (row stride 40 elements = 80 B: sixteen consecutive rows
// land in sixteen different 16-byte bank groups for the ds_read_b128 fragment loads).
// EPI 0: C bf16 [M][ldc] (pad columns < ldc written as the zeros they accumulate) and, if stat_sum != nullptr,
//        stat_sum / stat_sq [row workers][stat_ld] = per-tile column sums of c and c^2 (fp32 accumulators).
// EPI 1: C fp32 [M][ldc], atomicAdd (split-K: blockIdx.z owns K range [z*klen, (z+1)*klen)).
constexpr int GEMM_LDS_STRIDE = 40;
constexpr int GEMM_MAX_GX = 1024;      // row-tile workgroups (each loops over its tiles; bounds the partial statistics)

template <int NT, int EPI>
__global__ __launch_bounds__(256) void mt_gemm_nt_kernel(int M, int N, int K, const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb, void* __restrict__ Cv,
                                                         int ldc, float* __restrict__ stat_sum,
                                                         float* __restrict__ stat_sq, int stat_ld, int klen) {
  // Wave layout inside the 128 x NT tile.  NT <= 128: four waves stacked along M, a wave = 32 rows x NT columns.
  // NT = 256: 2 x 2 waves, a wave = 64 rows x 128 columns -- per 32-wide K chunk a wave then reads 4 + 8 fragment
  // vectors from LDS instead of 2 + 16 (the B tile is no longer re-read by all four waves): the 256-column tile was
  // bound by LDS fragment reads (72 KB per chunk and workgroup against 512 MFMA cycles).
  constexpr bool W22 = NT == 256;
  constexpr int RB = W22 ? 2 : 1;             // 32-row blocks per wave
  constexpr int NB = W22 ? 4 : NT / 32;       // 32-column blocks per wave
  constexpr int WCOLS = 32 * NB;              // columns per wave
  constexpr int BL = (NT * 4 + 255) / 256;    // 16-byte B loads per thread and chunk
  // one allocation: [2][128 x 40] A chunks, [2][NT x 40] B chunks; the bf16 epilogue reuses it as four per-wave
  // [32 rows][EC + 8] patches (EC = min(NT, 128) columns per pass)
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (128 + NT) * GEMM_LDS_STRIDE];
  bf16_t (*sA)[128 * GEMM_LDS_STRIDE] = reinterpret_cast<bf16_t (*)[128 * GEMM_LDS_STRIDE]>(smem);
  bf16_t (*sB)[NT * GEMM_LDS_STRIDE] = reinterpret_cast<bf16_t (*)[NT * GEMM_LDS_STRIDE]>(smem + 2 * 128 * GEMM_LDS_STRIDE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wrow = W22 ? (wave & 1) * 64 : wave * 32;      // this wave's first row / column inside the tile
  const int wcol = W22 ? (wave >> 1) * 128 : 0;
  // (grid (row workers, column tiles, K slices); column tiles in x -- adjacent in launch order, to share the A rows
  // through L2 -- measured 4 % slower per step)
  const int n0 = blockIdx.y * NT;
  const int kb = blockIdx.z * klen, ke = min(K, kb + klen);
  const int nchunks = (ke - kb + 31) / 32;
  const int mtiles = (M + 127) / 128;
  float tsum[NB], tsq[NB];          // per-lane column statistics, accumulated over this workgroup's row tiles
#pragma unroll
  for (int i = 0; i < NB; ++i) { tsum[i] = 0.f; tsq[i] = 0.f; }

  // (requesting chunk 0 of the next row tile under the current tile's epilogue was measured: no gain for K <= 64 --
  // the co-resident workgroups already cover the round trip -- and 5-25 % slower for the 256-column tiles)
  // Two chunks in flight: chunk c + 2 is requested at the top of iteration c (into the register set c & 1) and moves to
  // LDS at the bottom of iteration c + 1 -- with one chunk ahead a K loop of 4 - 17 chunks waited for a cold global
  // round trip in every iteration (the 16 MFMAs of a chunk are 0.2 us).
  uint4 rA[2][2], rB[2][BL];
  // (every load is issued unconditionally at a clamped address and zeroed by a select afterwards: a guarded load
  // compiles to a branch, and across those branches the compiler falls back to s_waitcnt vmcnt(0) -- which waited for
  // the prefetched chunk in every iteration and made the loop a chain of exposed global round trips)
  auto gload = [&](int m0, int c, uint4 (&ra)[2], uint4 (&rb)[BL]) {
    const int k = kb + c * 32 + (tid & 3) * 8;
    const int kc = k < ke ? k : 0;
#pragma unroll
    for (int p = 0; p < 2; ++p)
      ra[p] = *reinterpret_cast<const uint4*>(A + (size_t)min(m0 + (tid >> 2) + p * 64, M - 1) * lda + kc);
#pragma unroll
    for (int p = 0; p < BL; ++p)
      rb[p] = *reinterpret_cast<const uint4*>(B + (size_t)min(n0 + ((tid + p * 256) >> 2), N - 1) * ldb + kc);
  };
  // out-of-range rows / k read a clamped address; they are zeroed HERE, when the registers move to LDS (an AND with a
  // mask: masking at load time would wait for the load at once, a select would be turned back into a guarded load)
  auto lstore = [&](int buf, int m0, int c, const uint4 (&ra)[2], const uint4 (&rb)[BL]) {
    const int seg = tid & 3;
    const bool kok = kb + c * 32 + seg * 8 < ke;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int row = (tid >> 2) + p * 64;
      const unsigned mk = (kok && m0 + row < M) ? 0xffffffffu : 0u;
      *reinterpret_cast<uint4*>(&sA[buf][row * GEMM_LDS_STRIDE + seg * 8]) =
          make_uint4(ra[p].x & mk, ra[p].y & mk, ra[p].z & mk, ra[p].w & mk);
    }
#pragma unroll
    for (int p = 0; p < BL; ++p) {
      const int row = (tid + p * 256) >> 2;
      const unsigned mk = (kok && n0 + row < N) ? 0xffffffffu : 0u;
      if (row < NT)
        *reinterpret_cast<uint4*>(&sB[buf][row * GEMM_LDS_STRIDE + seg * 8]) =
            make_uint4(rb[p].x & mk, rb[p].y & mk, rb[p].z & mk, rb[p].w & mk);
    }
  };
  for (int mt = blockIdx.x; mt < mtiles; mt += gridDim.x) {
  const int m0 = mt * 128;
  f32x16 acc[RB][NB];
#pragma unroll
  for (int j = 0; j < RB; ++j)
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  if (nchunks > 0) {
    gload(m0, 0, rA[0], rB[0]);
    if (nchunks > 1) gload(m0, 1, rA[1], rB[1]);
    __syncthreads();           // (the previous tile's patch / fragment reads are done)
    lstore(0, m0, 0, rA[0], rB[0]);
    __syncthreads();
  }
  auto chunk = [&](int c, uint4 (&ra_next2)[2], uint4 (&rb_next2)[BL], const uint4 (&ra_next)[2], const uint4 (&rb_next)[BL]) {
    // LDS holds chunk c in buffer c & 1; (ra_next, rb_next) hold chunk c + 1; (ra_next2, rb_next2) = the set chunk c
    // came from, free again: chunk c + 2 goes there
    const int buf = c & 1;
    if (c + 2 < nchunks) gload(m0, c + 2, ra_next2, rb_next2);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ko = ks * 16 + (lane >> 5) * 8;
      bf16x8 a[RB];
#pragma unroll
      for (int j = 0; j < RB; ++j)
        a[j] = __builtin_bit_cast(
            bf16x8, *reinterpret_cast<const uint4*>(&sA[buf][(wrow + j * 32 + (lane & 31)) * GEMM_LDS_STRIDE + ko]));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 b = __builtin_bit_cast(
            bf16x8, *reinterpret_cast<const uint4*>(&sB[buf][(wcol + nb * 32 + (lane & 31)) * GEMM_LDS_STRIDE + ko]));
#pragma unroll
        for (int j = 0; j < RB; ++j) acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b, acc[j][nb], 0, 0, 0);
      }
    }
    if (c + 1 < nchunks) lstore(buf ^ 1, m0, c + 1, ra_next, rb_next);
    __syncthreads();
  };
  for (int c = 0; c < nchunks; c += 2) {
    chunk(c, rA[0], rB[0], rA[1], rB[1]);
    if (c + 1 < nchunks) chunk(c + 1, rA[1], rB[1], rA[0], rB[0]);
  }

  // C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (EPI == 0) {
    // The accumulators hold one column per lane (2-byte stores, 128 B per wave instruction); the tile goes
    // through a per-wave LDS patch instead and leaves as 16-byte row-contiguous stores.
    bf16_t* C = reinterpret_cast<bf16_t*>(Cv);
    constexpr int EC = WCOLS < 128 ? WCOLS : 128; // columns per pass
    constexpr int PS = EC + 8;                    // patch row stride (elements): 16-byte aligned rows
    constexpr int SEG = EC / 8;                   // 16-byte segments per row
    static_assert(4 * 32 * PS <= 2 * (128 + NT) * GEMM_LDS_STRIDE, "patch fits the chunk buffers");
    bf16_t* patch = smem + wave * 32 * PS;
#pragma unroll
    for (int j = 0; j < RB; ++j) {
#pragma unroll
      for (int pass = 0; pass < WCOLS / EC; ++pass) {
#pragma unroll
        for (int nbl = 0; nbl < EC / 32; ++nbl) {
          const int nb = pass * (EC / 32) + nbl;
#pragma unroll
          for (int r = 0; r < 16; ++r)
            patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PS + nbl * 32 + (lane & 31)] = f2bf(acc[j][nb][r]);
          if (stat_sum) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { tsum[nb] += acc[j][nb][r]; tsq[nb] += acc[j][nb][r] * acc[j][nb][r]; }
          }
        }
        __syncthreads();
        const int cbase = n0 + wcol + pass * EC;
#pragma unroll
        for (int it = 0; it < (32 * SEG) / 64; ++it) {
          const int q = it * 64 + lane;
          const int rr = q / SEG, seg = q % SEG;
          const int row = m0 + wrow + j * 32 + rr, col = cbase + seg * 8;
          if (row < M && col < ldc)
            *reinterpret_cast<uint4*>(C + (size_t)row * ldc + col) = *reinterpret_cast<const uint4*>(patch + rr * PS + seg * 8);
        }
        __syncthreads();
      }
    }
  } else {
    float* C = reinterpret_cast<float*>(Cv);
#pragma unroll
    for (int j = 0; j < RB; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + wcol + nb * 32 + (lane & 31);
        if (col < N) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = m0 + wrow + j * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
            if (row < M) atomicAdd(&C[(size_t)row * ldc + col], acc[j][nb][r]);
          }
        }
      }
  }
  }   // row tiles

  if (EPI == 0 && stat_sum) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(&sA[0][0]);     // [2][4][WCOLS] floats <= 8 KiB
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float s = tsum[nb], q = tsq[nb];
      s += __shfl_xor(s, 32, 64);
      q += __shfl_xor(q, 32, 64);
      if (lane < 32) {
        red[(0 * 4 + wave) * WCOLS + nb * 32 + lane] = s;
        red[(1 * 4 + wave) * WCOLS + nb * 32 + lane] = q;
      }
    }
    __syncthreads();
    for (int t = tid; t < NT; t += 256) {
      if (n0 + t < stat_ld) {
        float s, q;
        if (W22) {                                        // column t belongs to the waves (0 | 1) + 2 * (t / 128)
          const int w0 = 2 * (t >> 7), c = t & 127;
          s = red[(0 * 4 + w0) * WCOLS + c] + red[(0 * 4 + w0 + 1) * WCOLS + c];
          q = red[(1 * 4 + w0) * WCOLS + c] + red[(1 * 4 + w0 + 1) * WCOLS + c];
        } else {
          s = (red[0 * NT + t] + red[1 * NT + t]) + (red[2 * NT + t] + red[3 * NT + t]);
          q = (red[4 * NT + t] + red[5 * NT + t]) + (red[6 * NT + t] + red[7 * NT + t]);
        }
        stat_sum[(size_t)blockIdx.x * stat_ld + n0 + t] = s;
        stat_sq[(size_t)blockIdx.x * stat_ld + n0 + t] = q;
      }
    }
  }
}

template <int EPI>
int launch_gemm_nt(int M, int N, int K, const bf16_t* A, int lda, const bf16_t* B, int ldb, void* C, int ldc,
                   float* ssum, float* ssq, int stat_ld, int ksplit, hipStream_t st) {
  const int ncols = EPI == 0 ? ldc : N;             // EPI 0 also writes the (zero) pad columns
  const int gx = min(pvn3d_ceil_div(M, 128), GEMM_MAX_GX);
  int klen = K;
  if (ksplit > 1) klen = pvn3d_ceil_div(pvn3d_ceil_div(K, ksplit), 32) * 32;
  const int gz = pvn3d_ceil_div(K, klen);
#define MT_GEMM(NT)                                                                                          \
  hipLaunchKernelGGL((mt_gemm_nt_kernel<NT, EPI>), dim3(gx, pvn3d_ceil_div(ncols, NT), gz), dim3(256), 0, st, \
                     M, N, K, A, lda, B, ldb, C, ldc, ssum, ssq, stat_ld, klen)
  if (ncols <= 32) MT_GEMM(32);
  else if (ncols <= 64) MT_GEMM(64);                 // (64-column tiles for wider outputs: 2 % slower per step)
  // bf16 outputs wider than 128 columns also take 128-column tiles: the 256-column tile keeps 128 accumulator
  // registers on top of ~170 others, i.e. ONE wave per SIMD, and nothing overlaps its load -> barrier -> MFMA -> store
  // phases (measured 10 - 20 % slower on every wide shape; A is re-read once per column tile, mostly from L2)
  else if (ncols <= 128 || EPI == 0) MT_GEMM(128);
  else MT_GEMM(256);
#undef MT_GEMM
  PVN3D_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------ wgrad, TN form
// dW[m][n] += sum over rows k of dY[k][m] * H[k][n]: the contraction runs over ROWS, which both operands store as
// their slow dimension.  The NT kernel above needs K contiguous, i.e. transposed copies of dY and H (a write and a
// re-read of both matrices per layer: 1.7 ms of a 20 ms step, plus 3.1 ms in split-K NT launches whose 128-row tiles
// are mostly padding when a layer has 16 - 64 channels).  This kernel reads the row-major matrices directly (layers of
// up to 512 x 544 channels; wider ones keep the NT path): every WAVE is an independent worker over its own range of 16-row steps -- it stages the
// step's [16 rows][channels] slices in a wave-private LDS patch with 16-byte stores and pulls the MFMA fragments out
// column-wise (eight 2-byte reads per fragment: the k index of an operand is the row).  No workgroup barrier in the
// loop; the work is HBM-bound (6 KB per step and wave against ~100 instructions), so the narrow LDS reads do not
// matter.  The four waves of a workgroup add their tiles in LDS, then one fp32 atomicAdd per element to dW.
// grid (tiles_m * tiles_n, workers / 4); MB x NB 32-blocks per tile, MB * NB <= 8.
template <int MB, int NB>
__global__ __launch_bounds__(256) void mt_wgrad_tn_kernel(long long rows, int M, int N, const bf16_t* __restrict__ dY,
                                                          int ldy, const bf16_t* __restrict__ H, int ldh,
                                                          float* __restrict__ dW, int ldw, int tiles_n) {
  constexpr int SA = 32 * MB + 8, SB = 32 * NB + 8;             // patch row strides (elements)
  constexpr int PATCH = 16 * (SA + SB);                         // one step of one wave
  constexpr int TILE_F = 32 * MB * 32 * NB;                     // floats of the output tile
  constexpr int LDS_E = (2 * 4 * PATCH * 2 > TILE_F * 4 ? 2 * 4 * PATCH : TILE_F * 2);   // elements (2 B)
  __shared__ __attribute__((aligned(16))) bf16_t smem[LDS_E];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * 32 * MB, n0 = tn * 32 * NB;
  // worker w takes the steps w, w + W, w + 2W, ... (W = all workers of this tile): at any moment the waves in flight
  // read one contiguous moving window of the matrices, like a streaming kernel, instead of W separate streams
  const long long total_steps = (rows + 15) >> 4;
  const long long W = (long long)gridDim.y * 4;
  const long long s_begin = (long long)blockIdx.y * 4 + wave;
  const long long s_end = total_steps;
  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16_t* const mine = smem + wave * 2 * PATCH;
  uint4 ra[MB], rb[NB];
  auto gload = [&](long long step) {
    const long long r0 = step << 4;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int q = i * 64 + lane, row = q / (4 * MB), seg = q - row * (4 * MB);
      const int ch = m0 + seg * 8;
      ra[i] = (r0 + row < rows && ch < ldy) ? *reinterpret_cast<const uint4*>(dY + (r0 + row) * ldy + ch) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = i * 64 + lane, row = q / (4 * NB), seg = q - row * (4 * NB);
      const int ch = n0 + seg * 8;
      rb[i] = (r0 + row < rows && ch < ldh) ? *reinterpret_cast<const uint4*>(H + (r0 + row) * ldh + ch) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
    bf16_t* pa = mine + buf * PATCH;
    bf16_t* pb = pa + 16 * SA;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int q = i * 64 + lane, row = q / (4 * MB), seg = q - row * (4 * MB);
      *reinterpret_cast<uint4*>(pa + row * SA + seg * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int q = i * 64 + lane, row = q / (4 * NB), seg = q - row * (4 * NB);
      *reinterpret_cast<uint4*>(pb + row * SB + seg * 8) = rb[i];
    }
  };
  // fragment = rows kb*8 .. kb*8+7 of column c, packed pairwise
  auto frag = [&](const bf16_t* base, int stride, int c) {
    const bf16_t* p = base + ((lane >> 5) * 8) * stride + c + (lane & 31);
    uint4 v;
    v.x = (unsigned)p[0] | ((unsigned)p[stride] << 16);
    v.y = (unsigned)p[2 * stride] | ((unsigned)p[3 * stride] << 16);
    v.z = (unsigned)p[4 * stride] | ((unsigned)p[5 * stride] << 16);
    v.w = (unsigned)p[6 * stride] | ((unsigned)p[7 * stride] << 16);
    return __builtin_bit_cast(bf16x8, v);
  };
  if (s_begin < s_end) {
    gload(s_begin);
    lstore(0);
    int buf = 0;
    for (long long st = s_begin; st < s_end; st += W, buf ^= 1) {
      if (st + W < s_end) gload(st + W);
      const bf16_t* pa = mine + buf * PATCH;
      const bf16_t* pb = pa + 16 * SA;
      bf16x8 b[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = frag(pb, SB, j * 32);
#pragma unroll
      for (int i = 0; i < MB; ++i) {
        const bf16x8 a = frag(pa, SA, i * 32);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[j], acc[i][j], 0, 0, 0);
      }
      if (st + W < s_end) lstore(buf ^ 1);
    }
  }
  // the four waves add their tiles in LDS one after the other (plain read-add-write: the wave whose turn it is owns the
  // tile; ds_add_f32 from four waves at once measured ~14 us per workgroup, more than the whole K loop), then one
  // global atomic per element
  float* tile = reinterpret_cast<float*>(smem);
  __syncthreads();
  for (int i = tid; i < TILE_F; i += 256) tile[i] = 0.f;
  __syncthreads();
  for (int w = 0; w < 4; ++w) {
    if (wave == w && s_begin < s_end) {
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), n = j * 32 + (lane & 31);
            tile[m * (32 * NB) + n] += acc[i][j][r];
          }
    }
    __syncthreads();
  }
  for (int i = tid; i < TILE_F; i += 256) {
    const int m = i / (32 * NB), n = i - m * (32 * NB);
    const float v = tile[i];
    if (m0 + m < M && n0 + n < N && v != 0.f) atomicAdd(&dW[(size_t)(m0 + m) * ldw + n0 + n], v);
  }
}

// ------------------------------------------------------------------------------------------------ layout helpers
// X [rows][ld] bf16 -> XT [ld][ldt] (ldt >= rows, both multiples of 8): (4096 / TC) x TC tiles through LDS, TC = 16 /
// 32 / 64 columns so that narrow matrices (ld = 16: the first SA level) still fill the tile.  16-byte global loads
// (8 channels of a row) and stores (8 rows of a channel).
template <int TC>
__global__ __launch_bounds__(256) void mt_transpose_kernel(int rows, int ld, const bf16_t* __restrict__ X,
                                                           bf16_t* __restrict__ XT, int ldt) {
  constexpr int TR = 4096 / TC;
  constexpr int LS = TR + 8;                 // LDS row stride: 16-byte aligned rows
  __shared__ __attribute__((aligned(16))) bf16_t tile[TC * LS];
  const int r0 = blockIdx.x * TR, c0 = blockIdx.y * TC;
  for (int t = threadIdx.x; t < TR * TC / 8; t += 256) {
    const int r = t / (TC / 8), c = (t % (TC / 8)) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < rows && c0 + c < ld) v = *reinterpret_cast<const uint4*>(X + (size_t)(r0 + r) * ld + c0 + c);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      tile[(c + 2 * i) * LS + r] = (bf16_t)(w[i] & 0xffffu);
      tile[(c + 2 * i + 1) * LS + r] = (bf16_t)(w[i] >> 16);
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < TR * TC / 8; t += 256) {
    const int c = t / (TR / 8), r = (t % (TR / 8)) * 8;
    if (c0 + c < ld && r0 + r < ldt)
      *reinterpret_cast<uint4*>(XT + (size_t)(c0 + c) * ldt + r0 + r) = *reinterpret_cast<const uint4*>(&tile[c * LS + r]);
  }
}

// fp32 [rows][cols] (row stride lds) -> bf16 [rows][ld] zero-padded; transpose != 0: out[c][r] = in[r][c]
__global__ void mt_pack_weight_kernel(int rows, int cols, const float* __restrict__ W, int lds, int transpose,
                                      bf16_t* __restrict__ out, int out_rows, int ld) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= out_rows * ld) return;
  const int r = i / ld, c = i % ld;
  float v = 0.f;
  if (!transpose) { if (r < rows && c < cols) v = W[(size_t)r * lds + c]; }
  else { if (c < rows && r < cols) v = W[(size_t)c * lds + r]; }
  out[i] = f2bf(v);
}

// ------------------------------------------------------------------------------------------------ gathers
// SA level input: X0[(b*m + j)*ns + s][c] = (c < 3: xyz[b, idx] - new_xyz[b, j]) | (c < 3 + C: feat[b, c - 3, idx])
// | 0.  feat element (b, c, n) at feat[b*fsb + c*fsc + n*fsn] (fp32: channel-major tensors and transposed views of
// point-major ones alike).  One thread = 8 consecutive channels of one row.
// eight consecutive floats from a 4-byte-aligned address: two 16-byte loads (global memory takes dword-aligned vectors)
struct __attribute__((packed, aligned(4))) mt_f4u { float x, y, z, w; };
__device__ __forceinline__ void load8u(const float* __restrict__ p, float (&v)[8]) {
  const mt_f4u a = *reinterpret_cast<const mt_f4u*>(p), c = *reinterpret_cast<const mt_f4u*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
}

__global__ void mt_gather_sa_kernel(int b, int n, int m, int ns, int C, int use_xyz, const float* __restrict__ xyz,
                                    const float* __restrict__ new_xyz, const float* __restrict__ feat, long long fsb,
                                    long long fsc, long long fsn, const int* __restrict__ idx,
                                    bf16_t* __restrict__ X0, int ld) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long rows = (long long)b * m * ns;
  if (t >= rows * cpr) return;
  long long row;
  int c0;
  split_idx(t, cpr, row, c0);
  const unsigned mns = (unsigned)m * (unsigned)ns;
  const int bi = (int)((unsigned)row / mns);
  const int j = (int)(((unsigned)row - (unsigned)bi * mns) / (unsigned)ns);
  const int k = idx[row];
  const int nx = use_xyz ? 3 : 0;
  float v[8];
  const float* fb = feat ? feat + bi * fsb + k * fsn : nullptr;
  if (fb && fsc == 1 && c0 >= nx && c0 - nx + 8 <= C) {
    // interior chunk of a point-major feature row: eight consecutive floats
    load8u(fb + (c0 - nx), v);
  } else {
    // every load is issued unconditionally at a clamped (valid) address and selected afterwards: a guarded load is a
    // branch with a round trip of its own
    float f[8], g[3];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int fc = c0 + i - nx;
      fc = fc < 0 ? 0 : (fc >= C ? C - 1 : fc);
      f[i] = (fb && C > 0) ? fb[fc * fsc] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) g[i] = xyz[((size_t)bi * n + k) * 3 + i] - new_xyz[((size_t)bi * m + j) * 3 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      v[i] = c < nx ? g[c < 3 ? c : 0] : (c < nx + C ? f[i] : 0.f);
    }
  }
  *reinterpret_cast<uint4*>(X0 + row * ld + c0) = pack8(v);
}

// dX0 [B*R][ld] bf16 -> out[b][c][r] fp32 for the channels [c_off, c_off + C): the layout the row-owner scatter
// kernels (group_points.hip, interpolate.hip: group_points_grad / three_interpolate_grad) stream.  64 x 64 tiles.
__global__ __launch_bounds__(256) void mt_unpack_cm_kernel(int R, int ld, int c_off, int C, const bf16_t* __restrict__ X,
                                                           float* __restrict__ out) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  X += (size_t)b * R * ld;
  out += (size_t)b * C * R;
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    const int r = t >> 6, c = t & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? bf2f(X[(size_t)(r0 + r) * ld + c_off + c0 + c]) : 0.f;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    const int c = t >> 6, r = t & 63;
    if (c0 + c < C && r0 + r < R) out[(size_t)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

// in[b][c][r] fp32 (channel-major, C channels) -> X [B*R][ld] bf16, zero in the pad columns: the gradient of a
// module that returned the reference's contiguous (B, C, n) layout
__global__ __launch_bounds__(256) void mt_pack_cm_kernel(int R, int ld, int C, const float* __restrict__ in,
                                                         bf16_t* __restrict__ X) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  in += (size_t)b * C * R;
  X += (size_t)b * R * ld;
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    const int c = t >> 6, r = t & 63;
    tile[c][r] = (c0 + c < C && r0 + r < R) ? in[(size_t)(c0 + c) * R + r0 + r] : 0.f;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < 64 * 64; t += 256) {
    const int r = t >> 6, c = t & 63;
    if (r0 + r < R && c0 + c < ld) X[(size_t)(r0 + r) * ld + c0 + c] = f2bf(tile[c][r]);
  }
}

// FP level input: X0[b*n + i][c] = (c < C2: sum_t w[b,i,t] * known[b, c, idx[b,i,t]]) | (c < C2 + C1:
// unknown[b, c - C2, i]) | 0   (pointnet2_modules.py:188-203: cat([interpolated, unknow_feats], dim=1))
__global__ void mt_gather_fp_kernel(int b, int n, int mk, int C2, int C1, const float* __restrict__ known,
                                    long long ksb, long long ksc, long long ksn, const float* __restrict__ unknown,
                                    long long usb, long long usc, long long usn, const int* __restrict__ idx,
                                    const float* __restrict__ w, bf16_t* __restrict__ X0, int ld) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long rows = (long long)b * n;
  if (t >= rows * cpr) return;
  long long row;
  int c0;
  split_idx(t, cpr, row, c0);
  const int bi = (int)((unsigned)row / (unsigned)n), i0 = (int)((unsigned)row - (unsigned)bi * (unsigned)n);
  const int k0 = idx[row * 3], k1 = idx[row * 3 + 1], k2 = idx[row * 3 + 2];
  const float w0 = w[row * 3], w1 = w[row * 3 + 1], w2 = w[row * 3 + 2];
  const float* kb = known + bi * ksb;
  float v[8];
  if (ksc == 1 && c0 + 8 <= C2) {
    // point-major known features: three rows, eight consecutive channels each (same expression as the generic path)
    float a0[8], a1[8], a2[8];
    load8u(kb + k0 * ksn + c0, a0);
    load8u(kb + k1 * ksn + c0, a1);
    load8u(kb + k2 * ksn + c0, a2);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = a0[i] * w0 + a1[i] * w1 + a2[i] * w2;
  } else {
    // unconditional loads at clamped addresses, selected afterwards (a guarded load is a branch with its own round trip)
    float a0[8], a1[8], a2[8], u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int kc = c0 + i;
      kc = kc >= C2 ? C2 - 1 : kc;
      const float* p = kb + kc * ksc;
      a0[i] = p[k0 * ksn]; a1[i] = p[k1 * ksn]; a2[i] = p[k2 * ksn];
      int uc = c0 + i - C2;
      uc = uc < 0 ? 0 : (uc >= C1 ? C1 - 1 : uc);
      u[i] = (unknown && C1 > 0) ? unknown[bi * usb + uc * usc + i0 * usn] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + i;
      v[i] = c < C2 ? a0[i] * w0 + a1[i] * w1 + a2[i] * w2 : (c < C2 + C1 ? u[i] : 0.f);
    }
  }
  *reinterpret_cast<uint4*>(X0 + row * ld + c0) = pack8(v);
}

// ---- backward of the layer-0 gathers, as gathers -------------------------------------------------------------
// The gradient of a gather is a scatter-add: dfeat[b, idx[b,e]] += dX0[row(e)].  Done with atomics it costs more
// than the GEMMs (round 3: 3.7 ms of a 26 ms step through the row-owner LDS-atomic kernels, 20 ms through global
// atomics).  The index lists are fixed per call, so they are inverted once (mt_csr_build: per cloud a counting sort of
// the entries by source row, in LDS) and the scatter becomes a gather again: one lane group per source row walks its
// list and sums contiguous bf16 row slices of dX0 (mt_inv_gather) -- no atomics on the data, point-major fp32 out.
// mt_csr_build: one workgroup per cloud; idx [b][E] in [0, n_src); start [b][n_src + 1]; ent [b][E] = the entry ids
// grouped by source row (order inside a group unspecified).
__global__ __launch_bounds__(1024) void mt_csr_build_kernel(int n_src, int E, const int* __restrict__ idx,
                                                           int* __restrict__ start, int* __restrict__ ent) {
  extern __shared__ int s_bin[];          // [n_src]: counts, then write cursors
  __shared__ int s_wsum[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* ib = idx + (size_t)b * E;
  for (int i = tid; i < n_src; i += 1024) s_bin[i] = 0;
  __syncthreads();
  for (int e = tid; e < E; e += 1024) {
    const int k = ib[e];
    if ((unsigned)k < (unsigned)n_src) atomicAdd(&s_bin[k], 1);
  }
  __syncthreads();
  const int per = (n_src + 1023) / 1024;
  const int lo = min(tid * per, n_src), hi = min(lo + per, n_src);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += s_bin[i];
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
  if (lane == 63) s_wsum[wave] = inc;
  __syncthreads();
  int run = inc - sum;
  for (int w = 0; w < wave; ++w) run += s_wsum[w];
  int* sb = start + (size_t)b * (n_src + 1);
  for (int i = lo; i < hi; ++i) {
    const int c = s_bin[i];
    s_bin[i] = run;
    sb[i] = run;
    run += c;
  }
  if (tid == 1023) sb[n_src] = run;
  __syncthreads();
  int* eb = ent + (size_t)b * E;
  for (int e = tid; e < E; e += 1024) {
    const int k = ib[e];
    if ((unsigned)k < (unsigned)n_src) eb[atomicAdd(&s_bin[k], 1)] = e;
  }
}

// out[(b * n_src + p) * out_ld + c] (+)= sum over the entries e of source row p of
// w[b][e] * dX[(b * (E / div) + e / div) * ld + c_off + c], c < C.  div = 1: SA (one dX0 row per entry), div = 3: FP
// (three weighted entries per dX0 row).  One wave per source row: LPP = 1 << lpp_shift lanes span the channels (lane
// l sums channels l, l + LPP, ...: contiguous 2-byte loads across the lanes) and the 64 / LPP lane groups take
// every (64 / LPP)-th entry of the list -- ball query pads a short neighbourhood with its first hit, so a few rows
// own lists hundreds of entries long, and those set the launch time when one lane group walks a list alone.
template <bool HAS_W>
__global__ __launch_bounds__(256) void mt_inv_gather_kernel(int n_src, int E, int div, int C, int c_off, int ld,
                                                            const bf16_t* __restrict__ dX, const int* __restrict__ start,
                                                            const int* __restrict__ ent, const float* __restrict__ w,
                                                            float* __restrict__ out, int out_ld, int accumulate,
                                                            int lpp_shift) {
  const int b = blockIdx.y;
  const int lpp = 1 << lpp_shift;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);                  // source row of this wave
  if (row >= n_src) return;
  const int lane = threadIdx.x & 63;
  const int gl = lane & (lpp - 1), slot = lane >> lpp_shift, nslot = 64 >> lpp_shift;
  const int* sb = start + (size_t)b * (n_src + 1);
  const int j0 = sb[row], j1 = sb[row + 1];
  const int* eb = ent + (size_t)b * E;
  const float* wb = HAS_W ? w + (size_t)b * E : nullptr;
  const bf16_t* base = dX + (size_t)b * (E / div) * ld + c_off + gl;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  int j = j0 + slot;
  for (; j + nslot < j1; j += 2 * nslot) {                              // two entries in flight per lane group
    const int e0 = eb[j], e1 = eb[j + nslot];
    const float w0 = HAS_W ? wb[e0] : 1.f, w1 = HAS_W ? wb[e1] : 1.f;
    const bf16_t* r0 = base + (size_t)(e0 / div) * ld;
    const bf16_t* r1 = base + (size_t)(e1 / div) * ld;
    float v0[8], v1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool on = gl + (k << lpp_shift) < C;
      v0[k] = on ? bf2f(r0[k << lpp_shift]) : 0.f;
      v1[k] = on ? bf2f(r1[k << lpp_shift]) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = fmaf(w1, v1[k], fmaf(w0, v0[k], acc[k]));
  }
  if (j < j1) {
    const int e0 = eb[j];
    const float w0 = HAS_W ? wb[e0] : 1.f;
    const bf16_t* r0 = base + (size_t)(e0 / div) * ld;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (gl + (k << lpp_shift) < C) acc[k] = fmaf(w0, bf2f(r0[k << lpp_shift]), acc[k]);
  }
  for (int o = lpp; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
  }
  if (slot == 0) {
    float* op = out + ((size_t)b * n_src + row) * out_ld + gl;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = gl + (k << lpp_shift);
      if (c < C) op[k << lpp_shift] = accumulate ? op[k << lpp_shift] + acc[k] : acc[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// partial [P][ld] x 2 -> per-channel mean, 1/std, folded scale a = gamma/std and shift b = beta - mean*a (zero in
// the pad channels), running statistics (momentum update, unbiased variance) like nn.BatchNorm2d in training mode.
// grid ceil(ld/32), block (32 channels x 32 partial lanes)
__global__ __launch_bounds__(1024) void mt_bn_finalize_kernel(int P, int ld, int C, double count,
                                                             const float* __restrict__ psum,
                                                             const float* __restrict__ psq,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, float momentum,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var,
                                                             float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                             float* __restrict__ a_out, float* __restrict__ b_out) {
  __shared__ double ss[32][32], sq[32][32];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, q = 0.0;
  if (c < ld)
    for (int p = pl; p < P; p += 32) { s += psum[(size_t)p * ld + c]; q += psq[(size_t)p * ld + c]; }
  ss[pl][cl] = s;
  sq[pl][cl] = q;
  __syncthreads();
  if (pl == 0 && c < ld) {
#pragma unroll
    for (int i = 1; i < 32; ++i) { s += ss[i][cl]; q += sq[i][cl]; }
    if (c < C) {
      const double mean = s / count;
      double var = q / count - mean * mean;
      if (var < 0.0) var = 0.0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float a = gamma[c] * invstd;
      mean_out[c] = (float)mean;
      invstd_out[c] = invstd;
      a_out[c] = a;
      b_out[c] = beta[c] - (float)mean * a;
      if (run_mean) {
        const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)unb;
      }
    } else {
      mean_out[c] = 0.f; invstd_out[c] = 0.f; a_out[c] = 0.f; b_out[c] = 0.f;
    }
  }
}

// H = relu(a y + b), 8 channels per thread
__global__ void mt_bn_relu_apply_kernel(long long rows, int ld, const bf16_t* __restrict__ Y,
                                        const float* __restrict__ a, const float* __restrict__ b,
                                        bf16_t* __restrict__ H) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cpr) return;
  long long row_;
  int c0;
  split_idx(t, cpr, row_, c0);
  float y[8];
  unpack8(*reinterpret_cast<const uint4*>(Y + t * 8), y);
  const float4 a0 = *reinterpret_cast<const float4*>(a + c0), a1 = *reinterpret_cast<const float4*>(a + c0 + 4);
  const float4 b0 = *reinterpret_cast<const float4*>(b + c0), b1 = *reinterpret_cast<const float4*>(b + c0 + 4);
  const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
  const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = fmaxf(fmaf(av[i], y[i], bv[i]), 0.f);
  *reinterpret_cast<uint4*>(H + t * 8) = pack8(y);
}

// max over the ns rows of every group: H [G*ns][ld] -> out[g*out_ld + c] (fp32, c < C), arg [G][ld] (uint8)
__global__ void mt_pool_max_kernel(long long G, int ns, int ld, int C, const bf16_t* __restrict__ H,
                                   float* __restrict__ out, long long out_ld, unsigned char* __restrict__ arg) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G * cpr) return;
  const long long g = t / cpr;
  const int c0 = (int)(t % cpr) * 8;
  float best[8];
  int bi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { best[i] = -__builtin_inff(); bi[i] = 0; }
  for (int s = 0; s < ns; ++s) {
    float h[8];
    unpack8(*reinterpret_cast<const uint4*>(H + (g * ns + s) * ld + c0), h);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (h[i] > best[i]) { best[i] = h[i]; bi[i] = s; }       // first maximum wins (as ATen's max_pool2d)
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (c0 + i < C) out[g * out_ld + c0 + i] = best[i];
    arg[g * ld + c0 + i] = (unsigned char)bi[i];
  }
}

// dH [G*ns][ld] = (s == arg ? dpool : 0)
__global__ void mt_pool_bwd_kernel(long long G, int ns, int ld, int C, const float* __restrict__ dout, long long out_ld,
                                   const unsigned char* __restrict__ arg, bf16_t* __restrict__ dH) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G * ns * cpr) return;
  const long long row = t / cpr;
  const int c0 = (int)(t % cpr) * 8;
  const long long g = row / ns;
  const int s = (int)(row % ns);
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
    v[i] = (c0 + i < C && arg[g * ld + c0 + i] == s) ? dout[g * out_ld + c0 + i] : 0.f;
  *reinterpret_cast<uint4*>(dH + t * 8) = pack8(v);
}

// fp32 gradient of a point-major output (rows, C) with row stride gld -> bf16 [rows][ld] zero-padded
__global__ void mt_pack_grad_kernel(long long rows, int ld, int C, const float* __restrict__ g, long long gld,
                                    bf16_t* __restrict__ dH) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cpr) return;
  const long long row = t / cpr;
  const int c0 = (int)(t % cpr) * 8;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = c0 + i < C ? g[row * gld + c0 + i] : 0.f;
  *reinterpret_cast<uint4*>(dH + t * 8) = pack8(v);
}

// H [rows][ld] bf16 -> out[row*out_ld + c] fp32 (c < C)
__global__ void mt_unpack_out_kernel(long long rows, int ld, int C, const bf16_t* __restrict__ H, float* __restrict__ out,
                                     long long out_ld) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cpr) return;
  const long long row = t / cpr;
  const int c0 = (int)(t % cpr) * 8;
  float h[8];
  unpack8(*reinterpret_cast<const uint4*>(H + t * 8), h);
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (c0 + i < C) out[row * out_ld + c0 + i] = h[i];
}

// partial sums of dz = dH.[H > 0] and dz.yhat (yhat = (y - mean)/std).  Block = 256 threads = (ld/8 channel chunks)
// x (row lanes); each block covers `rows_per_block` rows and writes one partial row.
__global__ __launch_bounds__(256) void mt_bn_bwd_reduce_kernel(long long rows, int ld, int rows_per_block,
                                                               const bf16_t* __restrict__ dH,
                                                               const bf16_t* __restrict__ Y,
                                                               const float* __restrict__ av, const float* __restrict__ bv,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               float* __restrict__ p1, float* __restrict__ p2) {
  extern __shared__ float s_red[];          // [2][rl][ld]
  const int cpr = ld >> 3;
  const int rl = 256 / cpr;                 // row lanes (>= 3 for ld <= 528)
  const int chunk = threadIdx.x % cpr, rlane = threadIdx.x / cpr;
  const int c0 = chunk * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (rlane < rl) {
    float mu[8], is[8], aa[8], bb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = mean[c0 + i]; is[i] = invstd[c0 + i]; aa[i] = av[c0 + i]; bb[i] = bv[c0 + i]; }
    // Block b takes the row groups b, b + P, b + 2P, ... of 4 rl rows (P = gridDim.x): the blocks in flight read one
    // moving window of the matrices instead of P separate streams (mt_wgrad_tn: 2.6x on cold data).  Four rows in
    // flight per thread (eight 16-byte loads): one row at a time is latency-bound at a third of the HBM rate.
    (void)rows_per_block;
    const long long group = 4LL * rl;
    for (long long base = (long long)blockIdx.x * group; base < rows; base += (long long)gridDim.x * group) {
      uint4 vg[4], vy[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long long r = base + rlane + (long long)u * rl;
        ok[u] = r < rows;
        const long long o = (ok[u] ? r : rows - 1) * ld + c0;          // (clamped: the load itself is unconditional)
        vg[u] = *reinterpret_cast<const uint4*>(dH + o);
        vy[u] = *reinterpret_cast<const uint4*>(Y + o);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float g[8], y[8];
        unpack8(vg[u], g);
        unpack8(vy[u], y);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float dz = (ok[u] & relu_on(aa[i], y[i], bb[i])) ? g[i] : 0.f;
          s1[i] += dz;
          s2[i] += dz * ((y[i] - mu[i]) * is[i]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s_red[(0 * rl + rlane) * ld + c0 + i] = s1[i];
      s_red[(1 * rl + rlane) * ld + c0 + i] = s2[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ld; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rl; ++r) { a += s_red[(0 * rl + r) * ld + c]; b += s_red[(1 * rl + r) * ld + c]; }
    p1[(size_t)blockIdx.x * ld + c] = a;
    p2[(size_t)blockIdx.x * ld + c] = b;
  }
}

// The same partial sums for the LAST layer of a set-abstraction chain, whose dH is the max-pool backward: one
// nonzero per (group, channel), at row g*ns + arg[g][c], of value bf16(dpool[g][c]).  Reads the pooled gradient,
// the arg indices and one y per (group, channel) instead of two dense (rows x ld) matrices (and the dense dH is never
// written).  Block = (ld/8 channel chunks) x (group lanes); a block covers `groups_per_block` groups.
__global__ __launch_bounds__(256) void mt_bn_bwd_reduce_pooled_kernel(long long G, int ns, int ld, int C,
                                                                      int groups_per_block, const float* __restrict__ dout,
                                                                      long long out_ld, const unsigned char* __restrict__ arg,
                                                                      const bf16_t* __restrict__ Y,
                                                                      const float* __restrict__ av, const float* __restrict__ bv,
                                                                      const float* __restrict__ mean,
                                                                      const float* __restrict__ invstd,
                                                                      float* __restrict__ p1, float* __restrict__ p2) {
  extern __shared__ float s_red[];          // [2][rl][ld]
  const int cpr = ld >> 3;
  const int rl = 256 / cpr;
  const int chunk = threadIdx.x % cpr, rlane = threadIdx.x / cpr;
  const int c0 = chunk * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  if (rlane < rl) {
    float mu[8], is[8], aa[8], bb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { mu[i] = mean[c0 + i]; is[i] = invstd[c0 + i]; aa[i] = av[c0 + i]; bb[i] = bv[c0 + i]; }
    const long long g0 = (long long)blockIdx.x * groups_per_block;
    const long long g1 = min(G, g0 + groups_per_block);
    for (long long g = g0 + rlane; g < g1; g += rl) {
      const uint2 ar = *reinterpret_cast<const uint2*>(arg + g * ld + c0);
      float d[8];
      bf16_t yv[8];
      if (c0 + 8 <= C && (out_ld & 3) == 0) {
        const float4 d0 = *reinterpret_cast<const float4*>(dout + g * out_ld + c0);
        const float4 d1 = *reinterpret_cast<const float4*>(dout + g * out_ld + c0 + 4);
        d[0] = d0.x; d[1] = d0.y; d[2] = d0.z; d[3] = d0.w; d[4] = d1.x; d[5] = d1.y; d[6] = d1.z; d[7] = d1.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = c0 + i < C ? dout[g * out_ld + c0 + i] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {           // all eight (in-bounds: the pad channels' arg is 0) requested together
        const int sidx = (int)(((i < 4 ? ar.x : ar.y) >> (8 * (i & 3))) & 0xff);
        yv[i] = Y[(g * ns + sidx) * ld + c0 + i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float y = bf2f(yv[i]);
        const float dz = (c0 + i < C && relu_on(aa[i], y, bb[i])) ? bf2f(f2bf(d[i])) : 0.f;
        s1[i] += dz;
        s2[i] += dz * ((y - mu[i]) * is[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      s_red[(0 * rl + rlane) * ld + c0 + i] = s1[i];
      s_red[(1 * rl + rlane) * ld + c0 + i] = s2[i];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < ld; c += 256) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rl; ++r) { a += s_red[(0 * rl + r) * ld + c]; b += s_red[(1 * rl + r) * ld + c]; }
    p1[(size_t)blockIdx.x * ld + c] = a;
    p2[(size_t)blockIdx.x * ld + c] = b;
  }
}

// dY = a.dz + k1.y + k0 for that layer: dz = (s == arg[g][c] and H > 0) ? bf16(dpool[g][c]) : 0
__global__ void mt_bn_bwd_apply_pooled_kernel(long long G, int ns, int ld, int C, const float* __restrict__ dout,
                                              long long out_ld, const unsigned char* __restrict__ arg,
                                              const bf16_t* __restrict__ Y, const float* __restrict__ a,
                                              const float* __restrict__ b, const float* __restrict__ k1,
                                              const float* __restrict__ k0, bf16_t* __restrict__ dY) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G * ns * cpr) return;
  long long row, g;
  int c0, s;
  split_idx(t, cpr, row, c0);
  split_row(row, ns, g, s);
  const uint2 ar = *reinterpret_cast<const uint2*>(arg + g * ld + c0);
  float y[8], o[8], d[8], av[8], bv[8], k1v[8], k0v[8];
  unpack8(*reinterpret_cast<const uint4*>(Y + t * 8), y);
  load8f(a + c0, av); load8f(b + c0, bv); load8f(k1 + c0, k1v); load8f(k0 + c0, k0v);
  // everything is read unconditionally, up front: a guarded load is a round trip of its own
  if (c0 + 8 <= C && (out_ld & 3) == 0) {
    load8f(dout + g * out_ld + c0, d);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = c0 + i < C ? dout[g * out_ld + c0 + i] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int sidx = (int)(((i < 4 ? ar.x : ar.y) >> (8 * (i & 3))) & 0xff);
    const bool on = (c0 + i < C) & (sidx == s) & relu_on(av[i], y[i], bv[i]);
    const float dz = on ? bf2f(f2bf(d[i])) : 0.f;
    o[i] = fmaf(av[i], dz, fmaf(k1v[i], y[i], k0v[i]));
  }
  *reinterpret_cast<uint4*>(dY + t * 8) = pack8(o);
}

// last layer forward: out[g][c] = max over the ns rows of bf16(relu(a y + b)) and its arg index, straight from Y
// (the post-ReLU matrix of that layer is never written: nothing else reads it)
__global__ void mt_bn_relu_pool_kernel(long long G, int ns, int ld, int C, const bf16_t* __restrict__ Y,
                                       const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                       long long out_ld, unsigned char* __restrict__ arg) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G * cpr) return;
  const long long g = t / cpr;
  const int c0 = (int)(t % cpr) * 8;
  float aa[8], bb[8], best[8];
  int bi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { aa[i] = a[c0 + i]; bb[i] = b[c0 + i]; best[i] = -__builtin_inff(); bi[i] = 0; }
  for (int s = 0; s < ns; ++s) {
    float y[8];
    unpack8(*reinterpret_cast<const uint4*>(Y + (g * ns + s) * ld + c0), y);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float h = bf2f(f2bf(fmaxf(fmaf(aa[i], y[i], bb[i]), 0.f)));
      if (h > best[i]) { best[i] = h; bi[i] = s; }             // first maximum wins (as ATen's max_pool2d)
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (c0 + i < C) out[g * out_ld + c0 + i] = best[i];
    arg[g * ld + c0 + i] = (unsigned char)bi[i];
  }
}

// -> dgamma = sum dz.yhat, dbeta = sum dz, and the affine form of the BatchNorm backward
//    dY = a.(dz - mean(dz) - yhat.mean(dz.yhat)) = a.dz + k1.y + k0
__global__ __launch_bounds__(1024) void mt_bn_bwd_finalize_kernel(int P, int ld, int C, double count,
                                                                 const float* __restrict__ p1,
                                                                 const float* __restrict__ p2,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ a,
                                                                 float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 float* __restrict__ k1, float* __restrict__ k0) {
  __shared__ double ss[32][32], sq[32][32];
  const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s = 0.0, q = 0.0;
  if (c < ld)
    for (int p = pl; p < P; p += 32) { s += p1[(size_t)p * ld + c]; q += p2[(size_t)p * ld + c]; }
  ss[pl][cl] = s;
  sq[pl][cl] = q;
  __syncthreads();
  if (pl == 0 && c < ld) {
#pragma unroll
    for (int i = 1; i < 32; ++i) { s += ss[i][cl]; q += sq[i][cl]; }
    if (c < C) {
      dbeta[c] = (float)s;
      dgamma[c] = (float)q;
      const double c1 = s / count, c2 = q / count;
      const double kk1 = -(double)a[c] * c2 * (double)invstd[c];
      k1[c] = (float)kk1;
      k0[c] = (float)(-(double)a[c] * c1 - kk1 * (double)mean[c]);
    } else {
      k1[c] = 0.f; k0[c] = 0.f;
    }
  }
}

// dY = a.dz + k1.y + k0 with dz = dH.[H > 0]
__global__ void mt_bn_bwd_apply_kernel(long long rows, int ld, const bf16_t* __restrict__ dH,
                                       const bf16_t* __restrict__ Y, const float* __restrict__ a,
                                       const float* __restrict__ b, const float* __restrict__ k1,
                                       const float* __restrict__ k0, bf16_t* __restrict__ dY) {
  const int cpr = ld >> 3;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * cpr) return;
  long long row_;
  int c0;
  split_idx(t, cpr, row_, c0);
  float g[8], y[8], o[8], av[8], bv[8], k1v[8], k0v[8];
  unpack8(*reinterpret_cast<const uint4*>(dH + t * 8), g);
  unpack8(*reinterpret_cast<const uint4*>(Y + t * 8), y);
  load8f(a + c0, av); load8f(b + c0, bv); load8f(k1 + c0, k1v); load8f(k0 + c0, k0v);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float dz = relu_on(av[i], y[i], bv[i]) ? g[i] : 0.f;
    o[i] = fmaf(av[i], dz, fmaf(k1v[i], y[i], k0v[i]));
  }
  *reinterpret_cast<uint4*>(dY + t * 8) = pack8(o);
}

inline unsigned grid1(long long work, int block) { return (unsigned)((work + block - 1) / block); }

}  // namespace

// ---------------------------------------------------------------------------------------------------- C ABI
#define MT_ST ((hipStream_t)stream)

extern "C" int pvn3d_mt_gemm_nt(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                float* stat_sum, float* stat_sq, int stat_ld, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || (K & 15) || (lda & 7) || (ldb & 7) || !A || !B || !C) return (int)hipErrorInvalidValue;
  return launch_gemm_nt<0>(M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, stat_sum, stat_sq, stat_ld, 1,
                           MT_ST);
}

extern "C" int pvn3d_mt_gemm_nt_stat_rows(int M) { return min(pvn3d_ceil_div(M, 128), GEMM_MAX_GX); }

// C fp32 [M][ldc] += A . B^T over K split `ksplit` ways (atomics); C must hold the values to add to (zeros)
extern "C" int pvn3d_mt_gemm_nt_splitk(int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                                       int ldc, int ksplit, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || (K & 15) || (lda & 7) || (ldb & 7) || !A || !B || !C || ksplit < 1) return (int)hipErrorInvalidValue;
  return launch_gemm_nt<1>(M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc, nullptr, nullptr, 0, ksplit,
                           MT_ST);
}

extern "C" int pvn3d_mt_wgrad_tn_ok(int M, int N) { return M > 0 && N > 0 && M <= 512 && N <= 544; }

// dW (M, N) fp32 += dY^T . H over `rows` rows; dY [rows][ldy], H [rows][ldh] bf16 row-major (ld multiples of 8,
// pad channels zero); M, N <= 128.  dW must be initialised by the caller (it is accumulated into).
extern "C" int pvn3d_mt_wgrad_tn(long long rows, int M, int N, const void* dY, int ldy, const void* H, int ldh, float* dW,
                                 int ldw, void* stream) {
  if (rows <= 0 || M <= 0 || N <= 0) return 0;
  if (!pvn3d_mt_wgrad_tn_ok(M, N) || (ldy & 7) || (ldh & 7) || ldy < M || ldh < N || !dY || !H || !dW)
    return (int)hipErrorInvalidValue;
  const int mb_all = pvn3d_ceil_div(M, 32), nb_all = pvn3d_ceil_div(N, 32);
  // tile = MB x NB 32-blocks per wave, MB * NB <= 8 (128 accumulator registers); wider layers are cut into several
  // tiles, whose workers sweep the rows at the same pace, so the re-read operand slices are L2 hits
  int MB, NB;
  if (nb_all <= 2) { MB = mb_all >= 3 ? 4 : mb_all; NB = nb_all; }
  else { MB = mb_all >= 2 ? 2 : 1; NB = 4; }
  if (MB == 1 && NB == 3) NB = 4;
  const int tiles_m = pvn3d_ceil_div(M, 32 * MB), tiles_n = pvn3d_ceil_div(N, 32 * NB);
  const long long steps = (rows + 15) >> 4;
  // ~3000 workers (waves) over the chip (every workgroup ends with a tile reduction + atomics: 6000 measured slower), at least 8 steps each
  long long workers = 3072 / (tiles_m * tiles_n);
  if (workers < 4) workers = 4;
  long long spw = (steps + workers - 1) / workers;
  if (spw < 8) spw = 8;
  workers = (steps + spw - 1) / spw;
  const dim3 grid(tiles_m * tiles_n, (unsigned)((workers + 3) / 4));
#define MT_WG(MB_, NB_)                                                                                             \
  hipLaunchKernelGGL((mt_wgrad_tn_kernel<MB_, NB_>), grid, dim3(256), 0, MT_ST, rows, M, N, (const bf16_t*)dY, ldy,  \
                     (const bf16_t*)H, ldh, dW, ldw, tiles_n)
  if (MB == 1 && NB == 1) MT_WG(1, 1);
  else if (MB == 1 && NB == 2) MT_WG(1, 2);
  else if (MB == 1 && NB == 4) MT_WG(1, 4);
  else if (MB == 2 && NB == 1) MT_WG(2, 1);
  else if (MB == 2 && NB == 2) MT_WG(2, 2);
  else if (MB == 2 && NB == 4) MT_WG(2, 4);
  else if (MB == 4 && NB == 1) MT_WG(4, 1);
  else if (MB == 4 && NB == 2) MT_WG(4, 2);
  else return (int)hipErrorInvalidValue;
#undef MT_WG
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_transpose(long long rows, int ld, const void* X, void* XT, long long ldt, void* stream) {
  if (rows <= 0 || ld <= 0) return 0;
  if (rows > 0x7fffffffLL || ldt > 0x7fffffffLL) return (int)hipErrorInvalidValue;
  // the kernel writes 8 rows per 16-byte store at XT[c * ldt + r]: anything else is misaligned and runs into the next
  // channel row (and past the buffer in the last one)
  if ((rows & 7) || (ldt & 7) || ldt < rows) return (int)hipErrorInvalidValue;
  if (ld <= 16)
    hipLaunchKernelGGL(mt_transpose_kernel<16>, dim3(pvn3d_ceil_div((int)rows, 256), pvn3d_ceil_div(ld, 16)), dim3(256), 0,
                       MT_ST, (int)rows, ld, (const bf16_t*)X, (bf16_t*)XT, (int)ldt);
  else if (ld <= 32)
    hipLaunchKernelGGL(mt_transpose_kernel<32>, dim3(pvn3d_ceil_div((int)rows, 128), pvn3d_ceil_div(ld, 32)), dim3(256), 0,
                       MT_ST, (int)rows, ld, (const bf16_t*)X, (bf16_t*)XT, (int)ldt);
  else
    hipLaunchKernelGGL(mt_transpose_kernel<64>, dim3(pvn3d_ceil_div((int)rows, 64), pvn3d_ceil_div(ld, 64)), dim3(256), 0,
                       MT_ST, (int)rows, ld, (const bf16_t*)X, (bf16_t*)XT, (int)ldt);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_pack_weight(int rows, int cols, const float* W, int lds, int transpose, void* out, int out_rows,
                                    int ld, void* stream) {
  if (out_rows <= 0 || ld <= 0) return 0;
  hipLaunchKernelGGL(mt_pack_weight_kernel, dim3(grid1((long long)out_rows * ld, 256)), dim3(256), 0, MT_ST, rows, cols,
                     W, lds, transpose, (bf16_t*)out, out_rows, ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_gather_sa(int b, int n, int m, int ns, int C, int use_xyz, const float* xyz,
                                  const float* new_xyz, const float* feat, long long fsb, long long fsc, long long fsn,
                                  const int* idx, void* X0, int ld, void* stream) {
  const long long rows = (long long)b * m * ns;
  if (rows <= 0) return 0;
  if ((ld & 15) || ld < (use_xyz ? 3 : 0) + C || rows * (ld >> 3) >= 0x7fffffffLL || (long long)m * ns >= 0x7fffffffLL)
    return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_gather_sa_kernel, dim3(grid1(rows * (ld >> 3), 256)), dim3(256), 0, MT_ST, b, n, m, ns, C,
                     use_xyz, xyz, new_xyz, feat, fsb, fsc, fsn, idx, (bf16_t*)X0, ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_unpack_cm(int b, int R, int ld, int c_off, int C, const void* X, float* out, void* stream) {
  if (b <= 0 || R <= 0 || C <= 0) return 0;
  if (c_off < 0 || c_off + C > ld) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_unpack_cm_kernel, dim3(pvn3d_ceil_div(R, 64), pvn3d_ceil_div(C, 64), b), dim3(256), 0, MT_ST, R,
                     ld, c_off, C, (const bf16_t*)X, out);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_csr_build(int b, int n_src, int E, const int* idx, int* start, int* ent, void* stream) {
  if (b <= 0 || n_src <= 0) return 0;
  if (E < 0 || n_src > 32768) return (int)hipErrorInvalidValue;          // the bins live in LDS
  auto k = mt_csr_build_kernel;
  PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(k));
  hipLaunchKernelGGL(k, dim3(b), dim3(1024), (size_t)n_src * sizeof(int), MT_ST, n_src, E, idx, start, ent);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_inv_gather(int b, int n_src, int E, int div, int C, int c_off, int ld, const void* dX,
                                   const int* start, const int* ent, const float* w, float* out, int out_ld,
                                   int accumulate, void* stream) {
  if (b <= 0 || n_src <= 0 || C <= 0) return 0;
  if (div <= 0 || E % div || C > 512 || c_off < 0 || c_off + C > ld || out_ld < C) return (int)hipErrorInvalidValue;
  int shift = 3;                                                         // 8 .. 64 lanes across the channels
  while (shift < 6 && (8 << shift) < C) ++shift;
  const dim3 grid(pvn3d_ceil_div(n_src, 4), b);
  if (w)
    hipLaunchKernelGGL(mt_inv_gather_kernel<true>, grid, dim3(256), 0, MT_ST, n_src, E, div, C, c_off, ld,
                       (const bf16_t*)dX, start, ent, w, out, out_ld, accumulate, shift);
  else
    hipLaunchKernelGGL(mt_inv_gather_kernel<false>, grid, dim3(256), 0, MT_ST, n_src, E, div, C, c_off, ld,
                       (const bf16_t*)dX, start, ent, w, out, out_ld, accumulate, shift);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_pack_cm(int b, int R, int ld, int C, const float* in, void* X, void* stream) {
  if (b <= 0 || R <= 0 || ld <= 0) return 0;
  if (C > ld) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_pack_cm_kernel, dim3(pvn3d_ceil_div(R, 64), pvn3d_ceil_div(ld, 64), b), dim3(256), 0, MT_ST, R, ld,
                     C, in, (bf16_t*)X);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_gather_fp(int b, int n, int mk, int C2, int C1, const float* known, long long ksb, long long ksc,
                                  long long ksn, const float* unknown, long long usb, long long usc, long long usn,
                                  const int* idx, const float* w, void* X0, int ld, void* stream) {
  const long long rows = (long long)b * n;
  if (rows <= 0) return 0;
  if ((ld & 15) || ld < C2 + C1 || C2 <= 0 || rows * (ld >> 3) >= 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_gather_fp_kernel, dim3(grid1(rows * (ld >> 3), 256)), dim3(256), 0, MT_ST, b, n, mk, C2, C1,
                     known, ksb, ksc, ksn, unknown, usb, usc, usn, idx, w, (bf16_t*)X0, ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_finalize(int P, int ld, int C, double count, const float* psum, const float* psq,
                                    const float* gamma, const float* beta, float eps, float momentum, float* run_mean,
                                    float* run_var, float* mean, float* invstd, float* a, float* b, void* stream) {
  if (ld <= 0) return 0;
  hipLaunchKernelGGL(mt_bn_finalize_kernel, dim3(pvn3d_ceil_div(ld, 32)), dim3(1024), 0, MT_ST, P, ld, C, count, psum,
                     psq, gamma, beta, eps, momentum, run_mean, run_var, mean, invstd, a, b);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_relu_apply(long long rows, int ld, const void* Y, const float* a, const float* b, void* H,
                                      void* stream) {
  if (rows <= 0) return 0;
  if ((ld & 7) || rows * (ld >> 3) >= 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_bn_relu_apply_kernel, dim3(grid1(rows * (ld >> 3), 256)), dim3(256), 0, MT_ST, rows, ld,
                     (const bf16_t*)Y, a, b, (bf16_t*)H);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_pool_max(long long G, int ns, int ld, int C, const void* H, float* out, long long out_ld,
                                 unsigned char* arg, void* stream) {
  if (G <= 0) return 0;
  if (ns < 1 || ns > 255) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_pool_max_kernel, dim3(grid1(G * (ld >> 3), 256)), dim3(256), 0, MT_ST, G, ns, ld, C,
                     (const bf16_t*)H, out, out_ld, arg);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_pool_bwd(long long G, int ns, int ld, int C, const float* dout, long long out_ld,
                                 const unsigned char* arg, void* dH, void* stream) {
  if (G <= 0) return 0;
  hipLaunchKernelGGL(mt_pool_bwd_kernel, dim3(grid1(G * ns * (ld >> 3), 256)), dim3(256), 0, MT_ST, G, ns, ld, C, dout,
                     out_ld, arg, (bf16_t*)dH);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_pack_grad(long long rows, int ld, int C, const float* g, long long gld, void* dH, void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(mt_pack_grad_kernel, dim3(grid1(rows * (ld >> 3), 256)), dim3(256), 0, MT_ST, rows, ld, C, g, gld,
                     (bf16_t*)dH);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_unpack_out(long long rows, int ld, int C, const void* H, float* out, long long out_ld,
                                   void* stream) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(mt_unpack_out_kernel, dim3(grid1(rows * (ld >> 3), 256)), dim3(256), 0, MT_ST, rows, ld, C,
                     (const bf16_t*)H, out, out_ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

// rows per workgroup of the reduction: about 1024 workgroups, at least 64 rows each
static int mt_bwd_rows_per_block(long long rows) {
  long long rpb = (rows + 1023) / 1024;
  rpb = (rpb + 63) / 64 * 64;
  return (int)(rpb < 64 ? 64 : rpb);
}
extern "C" int pvn3d_mt_bn_bwd_partials(long long rows) {
  const int rpb = mt_bwd_rows_per_block(rows);
  return (int)((rows + rpb - 1) / rpb);
}

extern "C" int pvn3d_mt_bn_bwd_reduce(long long rows, int ld, const void* dH, const void* Y, const float* a,
                                      const float* b, const float* mean, const float* invstd, float* p1, float* p2,
                                      void* stream) {
  if (rows <= 0) return 0;
  if (ld > 2048 || (ld & 7)) return (int)hipErrorInvalidValue;
  const int rl = 256 / (ld >> 3);
  hipLaunchKernelGGL(mt_bn_bwd_reduce_kernel, dim3(pvn3d_mt_bn_bwd_partials(rows)), dim3(256),
                     (size_t)2 * rl * ld * sizeof(float), MT_ST, rows, ld, mt_bwd_rows_per_block(rows), (const bf16_t*)dH,
                     (const bf16_t*)Y, a, b, mean, invstd, p1, p2);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_bwd_reduce_pooled(long long G, int ns, int ld, int C, const float* dout, long long out_ld,
                                             const void* arg, const void* Y, const float* a, const float* b,
                                             const float* mean, const float* invstd, float* p1, float* p2, void* stream) {
  if (G <= 0) return 0;
  if (ld > 2048 || (ld & 7) || ns <= 0 || ns > 256) return (int)hipErrorInvalidValue;
  const int rl = 256 / (ld >> 3);
  hipLaunchKernelGGL(mt_bn_bwd_reduce_pooled_kernel, dim3(pvn3d_mt_bn_bwd_partials(G)), dim3(256),
                     (size_t)2 * rl * ld * sizeof(float), MT_ST, G, ns, ld, C, mt_bwd_rows_per_block(G), dout, out_ld,
                     (const unsigned char*)arg, (const bf16_t*)Y, a, b, mean, invstd, p1, p2);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_bwd_apply_pooled(long long G, int ns, int ld, int C, const float* dout, long long out_ld,
                                            const void* arg, const void* Y, const float* a, const float* b,
                                            const float* k1, const float* k0, void* dY, void* stream) {
  if (G <= 0) return 0;
  if ((ld & 7) || ns <= 0 || ns > 256 || G * ns * (ld >> 3) >= 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_bn_bwd_apply_pooled_kernel, dim3(grid1(G * ns * (ld >> 3), 256)), dim3(256), 0, MT_ST, G, ns, ld,
                     C, dout, out_ld, (const unsigned char*)arg, (const bf16_t*)Y, a, b, k1, k0, (bf16_t*)dY);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_relu_pool(long long G, int ns, int ld, int C, const void* Y, const float* a, const float* b,
                                     float* out, long long out_ld, void* arg, void* stream) {
  if (G <= 0) return 0;
  if ((ld & 7) || ns <= 0 || ns > 256) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_bn_relu_pool_kernel, dim3(grid1(G * (ld >> 3), 256)), dim3(256), 0, MT_ST, G, ns, ld, C,
                     (const bf16_t*)Y, a, b, out, out_ld, (unsigned char*)arg);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_bwd_finalize(int P, int ld, int C, double count, const float* p1, const float* p2,
                                        const float* mean, const float* invstd, const float* a, float* dgamma,
                                        float* dbeta, float* k1, float* k0, void* stream) {
  if (ld <= 0) return 0;
  hipLaunchKernelGGL(mt_bn_bwd_finalize_kernel, dim3(pvn3d_ceil_div(ld, 32)), dim3(1024), 0, MT_ST, P, ld, C, count, p1,
                     p2, mean, invstd, a, dgamma, dbeta, k1, k0);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_mt_bn_bwd_apply(long long rows, int ld, const void* dH, const void* Y, const float* a,
                                     const float* b, const float* k1, const float* k0, void* dY, void* stream) {
  if (rows <= 0) return 0;
  if ((ld & 7) || rows * (ld >> 3) >= 0x7fffffffLL) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(mt_bn_bwd_apply_kernel, dim3(grid1(rows * (ld >> 3), 256)), dim3(256), 0, MT_ST, rows, ld,
                     (const bf16_t*)dH, (const bf16_t*)Y, a, b, k1, k0, (bf16_t*)dY);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
