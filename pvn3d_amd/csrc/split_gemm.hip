// split_gemm.hip -- layer-by-layer form of the split-bf16 SharedMLP for chains whose hidden layer is too wide for the
// fused kernel of sa_mlp_split.hip (gfx950).  FP levels 2 and 3 of PVN3D's backbone (lib/pvn3d.py:114-118:
// [768 -> 512 -> 512] on 1024 <- 512 points, [1536 -> 512 -> 512] on 512 <- 128): 64 columns of a 512-wide hidden
// layer are 196 KB as three bf16 pieces, so the chain cannot stay in one CU's LDS.
//
// Arithmetic: the same as sa_mlp_split.hip -- every fp32 operand is the exact sum of three bf16 pieces (weights
// rounded to nearest on the host, activations by truncation, 8 + 8 + 8 mantissa bits), a product is the six partial
// products with piece indices i + j <= 2 accumulated in fp32 on v_mfma_f32_32x32x16_bf16, smallest terms first.
//
// Layout ("s16"): a matrix [rows][K] as rows x ceil(K/16) slabs x 3 pieces x 16 bf16 -- the three pieces of a
// 16-k slab side by side, 96 B, so that a 32-k chunk of a row is 192 contiguous bytes in HBM and in LDS.
//
// What the reference computes (pointnet2_modules.py:188-206): h = relu(bn(conv([interp(known); skip]))), then one
// more conv -> bn -> relu.  interp is linear in the features and the first conv is linear, so
//     W . [interp(known); skip] = interp(Wa . known) + Wb . skip        (W = [Wa | Wb], BatchNorm folded)
// and Wa . known runs over the m KNOWN points (2-4x fewer than the unknown ones).  Three launches of one kernel:
//     Z = Wa . known                              (fp32 out, no bias / relu)
//     H = relu(Wb . skip + interp(Z) + b1)        (interp gathered in the epilogue; s16 out)
//     Y = relu(W2 . H + b2)                       (fp32 point-major out)
// The regrouping changes the fp32 rounding sequence (the reference interpolates first): results agree with an fp64
// evaluation of the reference's formula to the same 1e-6 as the fused kernels (tests/test_gpu_ops.py).
//
// Kernel: workgroup tile 128 output channels x 128 points, 2 x 2 waves of 64 x 64, K in chunks of 32 through one LDS
// buffer per operand (row stride 208 B: the 16-byte fragment reads of 8 consecutive rows fall into 8 distinct 4-bank
// groups), the next chunk held in registers meanwhile; 53 KB of LDS -> up to three workgroups per CU cover each
// other's load / barrier phases.  The weights are the MFMA "A" operand, so a lane ends up with four consecutive
// channels of one point: 16-byte stores into a point-major fp32 row, 8-byte stores per piece into an s16 row.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) _Float16 sg_h2;
typedef __attribute__((ext_vector_type(2))) float sg_f2;

// Arithmetic (template parameter AR), as in sa_mlp_split.hip:
//   0  bf16 x 3 ("s16" matrices: rows x slabs x 3 pieces x 16 bf16 = 96 B per slab), six partial products;
//   1  fp16 x 2 ("h16" matrices: rows x slabs x 2 pieces x 16 fp16 = 64 B per slab), three partial products on
//      v_mfma_f32_32x32x16_f16; every h16 matrix holds x * s with s the power of two that puts a bound B on |x| (a DEVICE
//      float the caller names: pvn3d_absmax of the source, or pvn3d_bound_affine of such bounds) at 2^14; the weights carry
//      a host-side power-of-two scale; both are undone exactly on the accumulators.  Two thirds of the operand bytes
//      (the kernel is bound by L2 -> LDS traffic) and half of the MFMAs of (0).
#ifndef PVN3D_SG_DBG
#define PVN3D_SG_DBG 0      // tuning builds only (tools/sg_variants.sh): 1 no MFMAs, 2 no stores, 4 no operand loads,
#endif                      // 8 cycle stamps per phase of the LDS-DMA kernel, summed over waves (pvn3d_sg_prof_read)
#if PVN3D_SG_DBG & 8
constexpr int SG_PROF_WAVES = 32768;
__device__ unsigned sg_prof_buf[8 * SG_PROF_WAVES];   // per wave, plain stores (atomics on eight hot words serialise at the L2)
__device__ unsigned sg_prof_life[4 * SG_PROF_WAVES];  // per wave: start, end (s_memrealtime, 100 MHz), HW_ID, XCC_ID
#define SG_PROF_DECL unsigned sg_pa[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long sg_pt = __builtin_readcyclecounter(); \
  const unsigned sg_t0 = (unsigned)__builtin_amdgcn_s_memrealtime()
#define SG_PROF(P)                                                       \
  do {                                                                   \
    const unsigned long long t_ = __builtin_readcyclecounter();          \
    sg_pa[P] += (unsigned)(t_ - sg_pt);                                  \
    sg_pt = t_;                                                          \
  } while (0)
#define SG_PROF_FLUSH                                                                             \
  if ((threadIdx.x & 63) == 0) {                                                                  \
    const unsigned w_ = ((blockIdx.x + gridDim.x * blockIdx.y) * 4 + (threadIdx.x >> 6)) % SG_PROF_WAVES; \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) sg_prof_buf[8 * w_ + q_] = sg_pa[q_];        \
    unsigned hw_, xcc_;                                                                           \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                             \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                           \
    sg_prof_life[4 * w_] = sg_t0; sg_prof_life[4 * w_ + 1] = (unsigned)__builtin_amdgcn_s_memrealtime(); \
    sg_prof_life[4 * w_ + 2] = hw_; sg_prof_life[4 * w_ + 3] = xcc_;                              \
  }
#else
#define SG_PROF_DECL do { } while (0)
#define SG_PROF(P) do { } while (0)
#define SG_PROF_FLUSH do { } while (0)
#endif
constexpr int SG_T = 128;                 // tile edge
constexpr int sg_np(int ar) { return ar == 1 ? 2 : 3; }
constexpr int sg_slab(int ar) { return sg_np(ar) * 32; }            // bytes per 16-k slab of a row: 96 / 64
constexpr int sg_rowb(int ar) { return 2 * sg_slab(ar) + 16; }      // LDS bytes per tile row (32 k): 208 / 144 = 16 x odd
constexpr int sg_parts(int ar) { return 2 * sg_slab(ar) / 16; }     // 16-byte parts per row and chunk: 12 / 8
constexpr int sg_opb(int ar) { return SG_T * sg_rowb(ar); }         // one operand's chunk in LDS

struct SgArgs {
  int P, N, S;                 // points, real output channels, 16-k slabs of the contraction (even)
  const char* X;               // s16 [P][S]
  const char* W;               // s16 [ceil(N/128)*128][S], rows >= N zero
  const float* bias;           // [ceil(N/128)*128] or nullptr
  int relu;
  const float* Z;              // gathered add: fp32 [frames * zm][ldz] or nullptr
  int ldz, zn, zm;             // points per frame of this launch (zn) and rows per frame of Z (zm)
  const int* idx;              // [P][3] row of Z inside the point's frame
  const float* wgt;            // [P][3]
  float* out_f; int ld_out;    // fp32 [P][ld_out], channels < N
  char* out_s; int S_out;      // s16 [P][S_out]: every channel < 16 * S_out is written (pad channels are exact zeros)
  // fp16 x 2 only
  const float* x_bound;        // device: bound on |X| (the scale X was written with)
  float w_scale;               // host: power-of-two scale of W
  const float* w_row_mul;      // device [ceil(N/128)*128] or nullptr: per-output-channel multiplier of the accumulators
  const float* out_bound;      // device: bound on |out| for the h16 output (nullptr without out_s)
  unsigned* out_absmax;        // device or nullptr: atomic max of |out_f| (bit pattern), for the consumer's bound
};

__device__ __forceinline__ float sg_pow2_scale(float bound) {      // largest power of two s with bound * s <= 2^14
  int e;
  (void)frexpf(fmaxf(bound, 1e-30f), &e);
  return ldexpf(1.f, 14 - e);
}
// two fp16 pieces (round to nearest) of four scaled values
__device__ __forceinline__ void sg_split4h(const float (&x)[4], uint2& h, uint2& l) {
  pvn3d_split2_f16(x[0], x[1], h.x, l.x);            // (common.h: three instructions per pair)
  pvn3d_split2_f16(x[2], x[3], h.y, l.y);
}

// exact 3-way split of four fp32 values (consecutive channels) into three packed bf16x4
__device__ __forceinline__ void sg_split4(const float (&x)[4], uint2& h, uint2& m, uint2& l) {
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

// Epilogue of both GEMM kernels: a wave's accumulators are 2 x TJ blocks of 32 channels x 32 points at channel
// c0 + 64 wr, point p0 + 32 TJ wc.
template <int AR, int TJ>
__device__ __forceinline__ void sg_epilogue(const SgArgs& a, f32x16 (&acc)[2][TJ], const int c0, const int p0, const int wr,
                                            const int wc, const int lane, unsigned* sg_pa = nullptr,
                                            unsigned long long* sg_ptp = nullptr) {
#if PVN3D_SG_DBG & 8
#define SG_EPROF(P)                                                                                   \
  if (sg_pa) {                                                                                        \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
    const unsigned long long t_ = __builtin_readcyclecounter();                                       \
    sg_pa[P] += (unsigned)(t_ - *sg_ptp);                                                             \
    *sg_ptp = t_;                                                                                     \
  }
#else
#define SG_EPROF(P)
#endif
  constexpr int SLAB = sg_slab(AR);
  // epilogue.  C/D layout of the 32x32 MFMA: column (point) = lane & 31, row (channel) = 4 * (lane >> 5) + 8 * g + e
  // for register 4 g + e.  Two passes per point block: first every value is finished in its accumulator register
  // (gathered rows, bias, relu), then the stores follow -- loads and stores are not interleaved.  (Interleaved, the
  // compiler's code returned a zero for the first word of a gathered row in a few hundred of 3e7 values per launch --
  // last 16 lanes of a wave, timing dependent; the same loads behind an explicit s_waitcnt vmcnt(0) were always
  // right.  Root cause not established; tools/sg_check.py exercises the case at the failing size.)
  const int half = lane >> 5;
  float s_out = 1.f, amax = 0.f;
  if (AR == 1) {
    // accumulators carry w_scale * s_x: undo (an exact power of two) before anything is added
    const float inv = 1.f / (a.w_scale * sg_pow2_scale(*a.x_bound));
    // ... and the caller's per-row weight scales with it (powers of two as well: still exact)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float m4[4] = {inv, inv, inv, inv};
        if (a.w_row_mul) {
          const float4 rm = *reinterpret_cast<const float4*>(a.w_row_mul + c0 + wr * 64 + i * 32 + 8 * g + 4 * half);
          m4[0] = rm.x * inv; m4[1] = rm.y * inv; m4[2] = rm.z * inv; m4[3] = rm.w * inv;
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] *= m4[e];
      }
    if (a.out_s) s_out = sg_pow2_scale(*a.out_bound);
  }
  SG_EPROF(5)
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int p = p0 + wc * (32 * TJ) + j * 32 + (lane & 31);
    const bool live = p < a.P;
    const int pc = live ? p : a.P - 1;
    if (a.Z) {
      // three_interpolate of the Z rows, in the reference's order p0*w0 + p1*w1 + p2*w2 (pointnet2_utils.py:136-170)
      const int f = pc / a.zn;
      const float* zbase = a.Z + (size_t)f * a.zm * a.ldz;
      const float* zr[3];
      float zw[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        zr[t] = zbase + (size_t)a.idx[(size_t)pc * 3 + t] * a.ldz + c0 + wr * 64 + 4 * half;
        zw[t] = a.wgt[(size_t)pc * 3 + t];
      }
      constexpr int GZ = TJ > 2 ? 2 : 4;             // rows in flight per pass: 24 or 48 registers beside the accumulators
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g0 = 0; g0 < 4; g0 += GZ) {
          float4 z[GZ][3];
#pragma unroll
          for (int g = 0; g < GZ; ++g)
#pragma unroll
            for (int t = 0; t < 3; ++t) z[g][t] = *reinterpret_cast<const float4*>(zr[t] + i * 32 + 8 * (g0 + g));
          __builtin_amdgcn_s_waitcnt(0x0f70);        // vmcnt(0): every row is in its registers before the first use
#pragma unroll
          for (int g = 0; g < GZ; ++g) {
            acc[i][j][4 * (g0 + g) + 0] += z[g][0].x * zw[0] + z[g][1].x * zw[1] + z[g][2].x * zw[2];
            acc[i][j][4 * (g0 + g) + 1] += z[g][0].y * zw[0] + z[g][1].y * zw[1] + z[g][2].y * zw[2];
            acc[i][j][4 * (g0 + g) + 2] += z[g][0].z * zw[0] + z[g][1].z * zw[1] + z[g][2].z * zw[2];
            acc[i][j][4 * (g0 + g) + 3] += z[g][0].w * zw[0] + z[g][1].w * zw[1] + z[g][2].w * zw[2];
          }
        }
    }
    if (a.bias) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + c0 + wr * 64 + i * 32 + 8 * g + 4 * half);
          acc[i][j][4 * g + 0] += b.x; acc[i][j][4 * g + 1] += b.y; acc[i][j][4 * g + 2] += b.z; acc[i][j][4 * g + 3] += b.w;
        }
    }
    if (a.relu) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaxf(acc[i][j][r], 0.f);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    SG_EPROF(6)
    if (!live || ((PVN3D_SG_DBG & 2) && acc[0][j][0] != 12345.f)) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = c0 + wr * 64 + i * 32 + 8 * g + 4 * half;       // four consecutive channels
        const float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (a.out_f) {
          float* o = a.out_f + (size_t)p * a.ld_out + ch;
          if (ch + 3 < a.N) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ch + e < a.N) { o[e] = v[e]; amax = fmaxf(amax, fabsf(v[e])); }
          }
        }
        if (a.out_s && ch < 16 * a.S_out) {
          char* o = a.out_s + ((size_t)p * a.S_out + (ch >> 4)) * SLAB + (ch & 15) * 2;
          if (AR == 1) {
            const float y[4] = {v[0] * s_out, v[1] * s_out, v[2] * s_out, v[3] * s_out};
            uint2 h, l;
            sg_split4h(y, h, l);
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + 32) = l;
          } else {
            uint2 h, m, l;
            sg_split4(v, h, m, l);
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + 32) = m;
            *reinterpret_cast<uint2*>(o + 64) = l;
          }
        }
      }
  }
  SG_EPROF(7)
  if (a.out_absmax) {           // the consumer's bound on |out_f| (pvn3d_absmax without a second pass over the table)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    // (a plain read first: once the running maximum is above this wave's, no atomic is issued -- thousands of atomics
    // on one address serialise at the L2)
    if (lane == 0 && amax > 0.f && __float_as_uint(amax) > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(a.out_absmax, __float_as_uint(amax));
  }
}

// grid (channel tiles, point tiles), remapped per XCD below.
template <int AR>
__global__ __launch_bounds__(256, 2) void sg_gemm_kernel(SgArgs a) {
  constexpr int NP = sg_np(AR), SLAB = sg_slab(AR), ROWB = sg_rowb(AR), PARTS = sg_parts(AR), OPB = sg_opb(AR);
  constexpr int LD = SG_T * PARTS / 256;             // 16-byte chunk loads per thread and operand: 6 / 4
  __shared__ __attribute__((aligned(16))) char smem[2 * OPB];
  char* sW = smem;
  char* sX = smem + OPB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = wave >> 1;           // 64-channel / 64-point quadrant of this wave
  // (channel tile, point tile) of this workgroup.  Workgroups are dealt to the 8 XCDs round-robin in dispatch order, each
  // XCD with its own L2: in the plain (x = channel tile, y = point tile) reading the channel tiles of one point tile land
  // on different XCDs and every one of them pulls the X tile from HBM.  Here XCD x works through point tiles x, x + 8,
  // ..., all channel tiles of a point tile back to back on the same XCD: the X tile comes from HBM once.
  int ct = blockIdx.x, pt = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned q = lin >> 3;
    ct = (int)(q % gridDim.x);
    pt = (int)(q / gridDim.x) * 8 + (int)(lin & 7);
  }
  const int c0 = ct * SG_T, p0 = pt * SG_T;
  const size_t rowb = (size_t)a.S * SLAB;            // bytes per s16 / h16 row
  const int nch = a.S >> 1;

  // chunk loads: PARTS x 16 B per row and operand; thread -> (row, part) = ((tid + 256 j) / PARTS, (tid + 256 j) % PARTS)
  const char* gW[LD];
  const char* gX[LD];
  int lofs[LD];
#pragma unroll
  for (int j = 0; j < LD; ++j) {
    const int id = tid + 256 * j, row = id / PARTS, part = id - row * PARTS;
    gW[j] = a.W + (size_t)(c0 + row) * rowb + part * 16;
    gX[j] = a.X + (size_t)min(p0 + row, a.P - 1) * rowb + part * 16;
    lofs[j] = row * ROWB + part * 16;
  }
  u32x4 rW[LD], rX[LD];
#define SG_GLOAD(C)                                                             \
  _Pragma("unroll") for (int j = 0; j < LD; ++j) {                              \
    rW[j] = *reinterpret_cast<const u32x4*>(gW[j] + (size_t)(C) * (2 * SLAB));  \
    rX[j] = *reinterpret_cast<const u32x4*>(gX[j] + (size_t)(C) * (2 * SLAB));  \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (lane & 31) of a 32-row block, k half (lane >> 5); + slab * SLAB + piece * 32
  const char* fW = sW + (wr * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;
  const char* fX = sX + (wc * 64 + (lane & 31)) * ROWB + (lane >> 5) * 16;

  SG_GLOAD(0)
  for (int c = 0; c < nch; ++c) {
    __syncthreads();                                  // the previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < LD; ++j) {
      *reinterpret_cast<u32x4*>(sW + lofs[j]) = rW[j];
      *reinterpret_cast<u32x4*>(sX + lofs[j]) = rX[j];
    }
    __syncthreads();
    if (c + 1 < nch) { SG_GLOAD(c + 1) }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4 fa[2][NP], fb[2][NP];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          fa[i][p] = *reinterpret_cast<const u32x4*>(fW + i * 32 * ROWB + s * SLAB + p * 32);
          fb[i][p] = *reinterpret_cast<const u32x4*>(fX + i * 32 * ROWB + s * SLAB + p * 32);
        }
      if (AR == 1) {
#define SG_MM(PA, PB)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][PA]),     \
                                                         __builtin_bit_cast(f16x8, fb[j][PB]), acc[i][j], 0, 0, 0)
        SG_MM(0, 1); SG_MM(1, 0); SG_MM(0, 0);
#undef SG_MM
      } else {
#define SG_MM(PA, PB)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[i][PA]),   \
                                                          __builtin_bit_cast(bf16x8, fb[j][PB]), acc[i][j], 0, 0, 0)
        SG_MM(0, 2); SG_MM(2, 0); SG_MM(1, 1);
        SG_MM(0, 1); SG_MM(1, 0); SG_MM(0, 0);
#undef SG_MM
      }
    }
  }
#undef SG_GLOAD

  sg_epilogue<AR, 2>(a, acc, c0, p0, wr, wc, lane);
}

// Epilogue of the LDS-DMA kernel.  Same arithmetic per value as sg_epilogue<1, .> (accumulator x multiplier, + the
// interpolated rows in the reference's order, + bias, relu -- bit-identical results), organised around latency:
//   * multipliers and bias come from LDS (s_const, staged at kernel start);
//   * pass A finishes every value in its accumulator register; with an interpolated table the rows of a batch (one
//     32-channel block half of one 32-point block: 2 row groups x 3 neighbours = 6 loads of 16 B) are requested two
//     batches ahead of their use -- only loads are in flight in this pass, and loads return in order;
//   * pass B stores (fp32 rows and / or h16 rows); nothing waits for a store.
// (The waves of the two workgroups on a CU reach their epilogues together: whatever the epilogue waits for, the matrix
// pipe waits for as well.  Round-trip chains here were 25 % of a pure GEMM launch and 60 % of an interpolating one.)
template <int TJ, bool HZ>
__device__ __forceinline__ void sg_epilogue_dma(const SgArgs& a, f32x16 (&acc)[2][TJ], const float* s_const, const int c0,
                                                const int p0, const int wr, const int wc, const int lane,
                                                unsigned* sg_pa = nullptr, unsigned long long* sg_ptp = nullptr) {
  constexpr int SLAB = sg_slab(1);
  const int half = lane >> 5;
  const float* s_mul = s_const + wr * 64 + 4 * half;
  const float* s_bias = s_const + SG_T + wr * 64 + 4 * half;
  const float s_out = a.out_s ? sg_pow2_scale(*a.out_bound) : 1.f;
  float amax = 0.f;
  SG_EPROF(5)
  if constexpr (HZ) {
    // per point block: the three rows' addresses and weights (one round trip for all of them)
    // (byte offsets into the table in 32 bits: the launcher refuses tables of 4 GB or more)
    const char* zt = reinterpret_cast<const char*>(a.Z + c0 + wr * 64 + 4 * half);
    unsigned zr[TJ][3];
    float zw[TJ][3];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      const int pc = min(p0 + wc * (32 * TJ) + j * 32 + (lane & 31), a.P - 1);
      const unsigned zf = (unsigned)(pc / a.zn) * (unsigned)a.zm;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        zr[j][t] = (zf + (unsigned)a.idx[(size_t)pc * 3 + t]) * (unsigned)(a.ldz * 4);
        zw[j][t] = a.wgt[(size_t)pc * 3 + t];
      }
    }
    // batches k = (j, i, g pair): rows z[g][t], g in {2 gp, 2 gp + 1}
    constexpr int NB = TJ * 4;
    float4 zb[2][2][3];
#define SG_ZLOAD(K, SLOT)                                                                            \
  {                                                                                                  \
    const int j_ = (K) >> 2, i_ = ((K) >> 1) & 1, gp_ = (K) & 1;                                      \
    _Pragma("unroll") for (int g = 0; g < 2; ++g) _Pragma("unroll") for (int t = 0; t < 3; ++t)      \
        zb[SLOT][g][t] = *reinterpret_cast<const float4*>(zt + zr[j_][t] + (i_ * 32 + 8 * (2 * gp_ + g)) * 4); \
  }
    __builtin_amdgcn_sched_barrier(0);
    SG_ZLOAD(0, 0)
    SG_ZLOAD(1, 1)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int j = k >> 2, i = (k >> 1) & 1, gp = k & 1, slot = k & 1;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int gg = 2 * gp + g;
        const float4 m = *reinterpret_cast<const float4*>(s_mul + i * 32 + 8 * gg);
        const float4 b = *reinterpret_cast<const float4*>(s_bias + i * 32 + 8 * gg);
        const float4(&z)[3] = zb[slot][g];
        float v0 = acc[i][j][4 * gg + 0] * m.x, v1 = acc[i][j][4 * gg + 1] * m.y;
        float v2 = acc[i][j][4 * gg + 2] * m.z, v3 = acc[i][j][4 * gg + 3] * m.w;
        // three_interpolate in the reference's order p0*w0 + p1*w1 + p2*w2 (pointnet2_utils.py:136-170)
        v0 += z[0].x * zw[j][0] + z[1].x * zw[j][1] + z[2].x * zw[j][2];
        v1 += z[0].y * zw[j][0] + z[1].y * zw[j][1] + z[2].y * zw[j][2];
        v2 += z[0].z * zw[j][0] + z[1].z * zw[j][1] + z[2].z * zw[j][2];
        v3 += z[0].w * zw[j][0] + z[1].w * zw[j][1] + z[2].w * zw[j][2];
        if (a.bias) { v0 += b.x; v1 += b.y; v2 += b.z; v3 += b.w; }
        if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        acc[i][j][4 * gg + 0] = v0; acc[i][j][4 * gg + 1] = v1; acc[i][j][4 * gg + 2] = v2; acc[i][j][4 * gg + 3] = v3;
      }
      __builtin_amdgcn_sched_barrier(0);             // (batches stay in this order: hoisted loads cost registers)
      if (k + 2 < NB) {
        if (slot == 0) SG_ZLOAD(k + 2, 0) else SG_ZLOAD(k + 2, 1)
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef SG_ZLOAD
  } else {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) {
        const float4 m = *reinterpret_cast<const float4*>(s_mul + i * 32 + 8 * gg);
        const float4 b = *reinterpret_cast<const float4*>(s_bias + i * 32 + 8 * gg);
        const float m4[4] = {m.x, m.y, m.z, m.w}, b4[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = acc[i][j][4 * gg + e] * m4[e];
            if (a.bias) v += b4[e];
            acc[i][j][4 * gg + e] = a.relu ? fmaxf(v, 0.f) : v;
          }
      }
  }
  SG_EPROF(6)
  // pass B: stores
#pragma unroll
  for (int j = 0; j < TJ; ++j) {
    const int p = p0 + wc * (32 * TJ) + j * 32 + (lane & 31);
    if (p >= a.P || ((PVN3D_SG_DBG & 2) && acc[0][j][0] != 12345.f)) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = c0 + wr * 64 + i * 32 + 8 * g + 4 * half;       // four consecutive channels
        const float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (a.out_f) {
          float* o = a.out_f + (size_t)p * a.ld_out + ch;
          if (ch + 3 < a.N) {
            *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
            amax = fmaxf(amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (ch + e < a.N) { o[e] = v[e]; amax = fmaxf(amax, fabsf(v[e])); }
          }
        }
        if (a.out_s && ch < 16 * a.S_out) {
          char* o = a.out_s + ((size_t)p * a.S_out + (ch >> 4)) * SLAB + (ch & 15) * 2;
          const float y[4] = {v[0] * s_out, v[1] * s_out, v[2] * s_out, v[3] * s_out};
          uint2 h, l;
          sg_split4h(y, h, l);
          *reinterpret_cast<uint2*>(o) = h;
          *reinterpret_cast<uint2*>(o + 32) = l;
        }
      }
  }
  SG_EPROF(7)
  if (a.out_absmax) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    if (lane == 0 && amax > 0.f && __float_as_uint(amax) > __hip_atomic_load(a.out_absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(a.out_absmax, __float_as_uint(amax));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16 x 2, LDS-DMA form.  The kernel above moves every operand byte global -> VGPR -> LDS (ds_write_b128: 13 cycles of
// the CU's store path per wave-instruction, 416 cycles per 32-k chunk against 768 cycles of MFMA) and crosses two
// barriers per chunk with one LDS buffer.  Here:
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction = one 16-k slab of 16
//     rows, no staging registers, no ds_write);
//   * a stage is ONE 16-k slab of the tile's rows (64 B per row: both pieces), three stages in a ring: while a stage
//     is multiplied the next one has landed or is landing and the one after is being issued -- one barrier per stage;
//   * workgroup tile 128 channels x 64 TJ points (TJ = 4: 256 points, 24 KB per stage, two workgroups per CU), a wave
//     holds 2 x TJ accumulator blocks: 24 MFMAs per 12 fragment reads at TJ = 4 (12 per 8 before);
//   * an LDS row is the slab's 64 bytes without padding; the DMA's destination is lane-linear, so the swizzle that keeps
//     the 16-byte fragment reads conflict-free sits on the SOURCE side: 16-byte part q of row r lands in column
//     q ^ ((r >> 2) & 3), and the reader applies the same XOR.  (ds_read_b128 serves lanes in groups {0-3, 12-15, 20-27}
//     and {4-11, 16-19, 28-31}: per group the four rows that share r mod 4 -- the 16-bank quarter -- differ in
//     (r >> 2) & 3, so the sixteen lanes cover all 64 banks once.)
// The MFMA order per accumulator is the same as above (per slab: hi x lo, lo x hi, hi x hi): results are bit-identical
// to sg_gemm_kernel<1>, which stays as the cross-check (pvn3d_split_gemm2_tile128) and for launches too small to fill
// the chip with 256-point tiles.
//
// Ordering of the DMA (MI355X guide, LDS-DMA rules): a wave's own `s_waitcnt vmcnt(n)` retires its DMA writes, the
// barrier after it publishes them to the other waves, and only then is the stage read.  The stage overwritten by the
// DMA issued after that barrier was read one iteration earlier: every wave has issued the MFMAs that consumed those
// reads before it arrived at the barrier.  The asm statements are invisible to the compiler's own vmcnt bookkeeping;
// the loop contains no compiler-issued VMEM operation, and the last stage waits for vmcnt(0) before the epilogue's loads.
__device__ __forceinline__ void sg_glds16(const char* sbase, unsigned voff, unsigned lds_dst) {
  if (PVN3D_SG_DBG & 4) return;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int TJ, bool HZ>        // HZ: with the interpolated table (its own instantiation: the gather's registers)
__global__ __launch_bounds__(256, 2) void sg_gemm_dma_kernel(SgArgs a) {
  constexpr int PT = 64 * TJ;                        // points per workgroup tile
  constexpr int WB = SG_T * 64, XB = PT * 64;        // bytes of the two operands in a stage
  constexpr int STAGE = WB + XB;
  extern __shared__ __attribute__((aligned(1024))) char sg_dyn[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = wave >> 1;
  int ct = blockIdx.x, pt = blockIdx.y;
  if ((gridDim.y & 7) == 0) {                        // XCD x works through point tiles x, x + 8, ... (see above)
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    const unsigned q = lin >> 3;
    ct = (int)(q % gridDim.x);
    pt = (int)(q / gridDim.x) * 8 + (int)(lin & 7);
  }
  SG_PROF_DECL;
  const int c0 = ct * SG_T, p0 = pt * PT;
  const unsigned rowb = (unsigned)a.S * 64u;
  const int S = a.S;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)sg_dyn);
  // the tile's per-channel constants behind the ring: multiplier of the accumulators (the scales undone: exact powers
  // of two) and bias.  Published by the first barrier of the loop; the epilogue reads them with LDS latency instead of
  // two global round trips per point block.
  float* s_const = reinterpret_cast<float*>(sg_dyn + 3 * STAGE);
  if (tid < SG_T) {
    const float inv = 1.f / (a.w_scale * sg_pow2_scale(*a.x_bound));
    s_const[tid] = a.w_row_mul ? a.w_row_mul[c0 + tid] * inv : inv;
    s_const[SG_T + tid] = a.bias ? a.bias[c0 + tid] : 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // DMA sources: instruction t of this wave covers 16 rows; lane -> row (lane >> 2), LDS column (lane & 3)
  unsigned voW[2], voX[TJ];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int r = wave * 32 + t * 16 + (lane >> 2);
    voW[t] = (unsigned)r * rowb + (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 16);
  }
#pragma unroll
  for (int t = 0; t < TJ; ++t) {
    const int r = wave * (16 * TJ) + t * 16 + (lane >> 2);
    const int rg = min(p0 + r, a.P - 1) - p0;        // rows past the last point repeat it (never stored)
    voX[t] = (unsigned)rg * rowb + (unsigned)(((lane & 3) ^ ((r >> 2) & 3)) * 16);
  }
#if PVN3D_SG_DBG & 16
  // timing experiment (wrong data): eight lanes cover 128 contiguous bytes of a row, as a 32-k stage would
#pragma unroll
  for (int t = 0; t < 2; ++t) voW[t] = (unsigned)(wave * 32 + t * 8 + (lane >> 3)) * rowb + (lane & 7) * 16;
#pragma unroll
  for (int t = 0; t < TJ; ++t) voX[t] = (unsigned)(wave * (16 * TJ) + t * 8 + (lane >> 3)) * rowb + (lane & 7) * 16;
#endif
  const char* gWt = a.W + (size_t)c0 * rowb;
  const char* gXt = a.X + (size_t)p0 * rowb;
  const unsigned ldW = lds0 + wave * 2048, ldX = lds0 + WB + wave * (1024 * TJ);
#define SG2_ISSUE(SLABI, BOFF)                                                                     \
  {                                                                                                \
    const char* bw = gWt + (size_t)(SLABI) * 64;                                                   \
    const char* bx = gXt + (size_t)(SLABI) * 64;                                                   \
    _Pragma("unroll") for (int t = 0; t < 2; ++t) sg_glds16(bw, voW[t], ldW + (BOFF) + t * 1024);  \
    _Pragma("unroll") for (int t = 0; t < TJ; ++t) sg_glds16(bx, voX[t], ldX + (BOFF) + t * 1024); \
  }

  f32x16 acc[2][TJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads: row (lane & 31) of a 32-row block, k half (lane >> 5), piece p: column ((2 p) | half) ^ swizzle
  const int frow = lane & 31;
  const int fcol = (lane >> 5) ^ ((frow >> 2) & 3);
  const char* fW = sg_dyn + (wr * 64 + frow) * 64 + fcol * 16;
  const char* fX = sg_dyn + WB + (wc * (32 * TJ) + frow) * 64 + fcol * 16;
  const int px = 32 - 2 * (fcol & 2) * 16;           // piece 1 = column ^ 2: + 32 bytes or - 32 bytes

  SG2_ISSUE(0, 0)
  if (S > 1) SG2_ISSUE(1, STAGE)
  unsigned rb = 0, ib = 2 * STAGE;                   // ring offsets (wave-uniform): stage read now, stage issued now
  SG_PROF(0);
#pragma unroll 1
  for (int s = 0; s < S; ++s) {
    if (s + 1 < S) {
      if constexpr (TJ == 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    SG_PROF(1);
    asm volatile("s_barrier" ::: "memory");
    SG_PROF(2);
    if (s + 2 < S) SG2_ISSUE(s + 2, ib)
    SG_PROF(3);
    const char* w0 = fW + rb;
    const char* x0 = fX + rb;
    u32x4 fa[2][2], fb[TJ][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      fa[i][0] = *reinterpret_cast<const u32x4*>(w0 + i * 2048);
      fa[i][1] = *reinterpret_cast<const u32x4*>(w0 + i * 2048 + px);
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
      fb[j][0] = *reinterpret_cast<const u32x4*>(x0 + j * 2048);
      fb[j][1] = *reinterpret_cast<const u32x4*>(x0 + j * 2048 + px);
    }
#define SG2_MM(PA, PB)                                                                             \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j)     \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[i][PA]),     \
                                                         __builtin_bit_cast(f16x8, fb[j][PB]), acc[i][j], 0, 0, 0)
    if (!(PVN3D_SG_DBG & 1)) { SG2_MM(0, 1); SG2_MM(1, 0); SG2_MM(0, 0); }
    else { _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < TJ; ++j) acc[i][j][0] += __uint_as_float(fa[i][0][0] ^ fa[i][1][1] ^ fb[j][0][2] ^ fb[j][1][3]); }
#undef SG2_MM
    rb = rb == 2 * STAGE ? 0 : rb + STAGE;
    ib = ib == 2 * STAGE ? 0 : ib + STAGE;
#if PVN3D_SG_DBG & 8
    asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[1][TJ - 1][15]));   // the stage's MFMAs are issued before the stamp
#endif
    SG_PROF(4);
  }
#undef SG2_ISSUE
#if PVN3D_SG_DBG & 8
  sg_epilogue_dma<TJ, HZ>(a, acc, s_const, c0, p0, wr, wc, lane, sg_pa, &sg_pt);
#else
  sg_epilogue_dma<TJ, HZ>(a, acc, s_const, c0, p0, wr, wc, lane);
#endif
  SG_PROF(0);
  SG_PROF_FLUSH;
}
#if PVN3D_SG_DBG & 8
extern "C" int pvn3d_sg_prof_life(unsigned* out, int n_waves) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sg_prof_life), (size_t)16 * (n_waves < SG_PROF_WAVES ? n_waves : SG_PROF_WAVES));
}
extern "C" int pvn3d_sg_prof_read(unsigned long long* out8, int n_waves) {
  static unsigned host[8 * SG_PROF_WAVES];
  hipError_t e = hipMemcpyFromSymbol(host, HIP_SYMBOL(sg_prof_buf), sizeof(host));
  if (e != hipSuccess) return (int)e;
  for (int q = 0; q < 8; ++q) out8[q] = 0;
  for (int w = 0; w < n_waves && w < SG_PROF_WAVES; ++w)
    for (int q = 0; q < 8; ++q) out8[q] += host[8 * w + q];
  return 0;
}
#endif

// fp32 rows [rows][ld] (channels [0, c)) -> s16 [rows][S] by truncation split; channels >= c are zeros.
// One thread per (row, 4 channels).
template <int AR>
__global__ __launch_bounds__(256) void sg_split_rows_kernel(long long rows, int c, const float* __restrict__ src, int ld,
                                                            char* __restrict__ dst, int S, const float* __restrict__ bound) {
  const int q_per_row = S * 4;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= rows * q_per_row) return;
  const long long row = t / q_per_row;
  const int q = (int)(t - row * q_per_row), ch = q * 4;
  float x[4] = {0.f, 0.f, 0.f, 0.f};
  const float* s = src + row * ld + ch;
  if (ch + 3 < c && (ld & 3) == 0) {
    const float4 v = *reinterpret_cast<const float4*>(s);
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (ch + e < c) x[e] = s[e];
  }
  char* o = dst + ((size_t)row * S + (ch >> 4)) * sg_slab(AR) + (ch & 15) * 2;
  if (AR == 1) {
    const float s = sg_pow2_scale(*bound);
    const float y[4] = {x[0] * s, x[1] * s, x[2] * s, x[3] * s};
    uint2 h, l;
    sg_split4h(y, h, l);
    *reinterpret_cast<uint2*>(o) = h;
    *reinterpret_cast<uint2*>(o + 32) = l;
  } else {
    uint2 h, m, l;
    sg_split4(x, h, m, l);
    *reinterpret_cast<uint2*>(o) = h;
    *reinterpret_cast<uint2*>(o + 32) = m;
    *reinterpret_cast<uint2*>(o + 64) = l;
  }
}

// out = ca * *a + cb * *b + c0 (b may be null): the rigorous bound of a layer's output from the bounds of its inputs
__global__ void sg_bound_affine_kernel(float* out, const float* a, float ca, const float* b, float cb, float c0) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = (ca * *a + (b ? cb * *b : 0.f) + c0) * 1.01f;
}

}  // namespace

static int sg_split_rows_any(int ar, long long rows, int c, const float* src, int ld_src, void* dst, int slabs,
                             const float* bound, void* stream) {
  if (rows <= 0) return 0;
  if (!src || !dst || c <= 0 || slabs <= 0 || c > 16 * slabs || ld_src < c || (ar == 1 && !bound) ||
      ((uintptr_t)src & 15) != 0 || ((uintptr_t)dst & 15) != 0)
    return (int)hipErrorInvalidValue;
  const long long n = rows * (long long)slabs * 4;
  if (n > 0x7fffffffLL * 256) return (int)hipErrorInvalidValue;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (ar == 1)
    hipLaunchKernelGGL(sg_split_rows_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, rows, c, src, ld_src, (char*)dst, slabs, bound);
  else
    hipLaunchKernelGGL(sg_split_rows_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, rows, c, src, ld_src, (char*)dst, slabs, bound);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
extern "C" int pvn3d_split_rows(long long rows, int c, const float* src, int ld_src, void* dst_s16, int slabs,
                                void* stream) {
  return sg_split_rows_any(0, rows, c, src, ld_src, dst_s16, slabs, nullptr, stream);
}
extern "C" int pvn3d_split_rows2(long long rows, int c, const float* src, int ld_src, const float* src_bound, void* dst_h16,
                                 int slabs, void* stream) {
  return sg_split_rows_any(1, rows, c, src, ld_src, dst_h16, slabs, src_bound, stream);
}

extern "C" int pvn3d_bound_affine(float* out, const float* a, float ca, const float* b, float cb, float c0, void* stream) {
  if (!out || !a || !(ca >= 0.f) || !(cb >= 0.f) || !(c0 >= 0.f)) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sg_bound_affine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out, a, ca, b, cb, c0);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

static int sg_gemm_any(int ar, int form, int n_points, int n_out, int slabs, const void* x, const void* w, const float* bias_padded,
                       int relu, const float* z, int ldz, int z_points_per_frame, int z_rows_per_frame, const int* idx,
                       const float* weight, float* out_f32, int ld_out, void* out_s, int slabs_out, const float* x_bound,
                       float w_scale, const float* w_row_mul, const float* out_bound, float* out_absmax, void* stream) {
  if (n_points <= 0 || n_out <= 0) return 0;
  if (!x || !w || slabs <= 0 || (slabs & 1) || (!out_f32 && !out_s) || (out_f32 && ld_out < n_out) ||
      (out_s && (slabs_out <= 0 || 16 * slabs_out > pvn3d_ceil_div(n_out, SG_T) * SG_T)) ||
      (z && (!idx || !weight || (ldz & 3) || ldz < pvn3d_ceil_div(n_out, SG_T) * SG_T || z_points_per_frame <= 0 ||
             z_rows_per_frame <= 0 || ((uintptr_t)z & 15) != 0)) ||
      ((uintptr_t)x & 15) != 0 || ((uintptr_t)w & 15) != 0 || (out_f32 && (((uintptr_t)out_f32 & 15) != 0 || (ld_out & 3))) ||
      (out_s && ((uintptr_t)out_s & 15) != 0) || (bias_padded && ((uintptr_t)bias_padded & 15) != 0) ||
      (w_row_mul && ((uintptr_t)w_row_mul & 15) != 0))
    return (int)hipErrorInvalidValue;
  if (ar == 1) {
    int e = 0;
    if (!x_bound || !(w_scale > 0.f) || frexpf(w_scale, &e) != 0.5f || (out_s && !out_bound)) return (int)hipErrorInvalidValue;
  } else if (out_absmax) {
    return (int)hipErrorInvalidValue;
  }
  SgArgs a = {};
  a.P = n_points; a.N = n_out; a.S = slabs;
  a.X = (const char*)x; a.W = (const char*)w; a.bias = bias_padded; a.relu = relu;
  a.Z = z; a.ldz = ldz; a.zn = z_points_per_frame; a.zm = z_rows_per_frame; a.idx = idx; a.wgt = weight;
  a.out_f = out_f32; a.ld_out = ld_out; a.out_s = (char*)out_s; a.S_out = slabs_out;
  a.x_bound = x_bound; a.w_scale = w_scale; a.w_row_mul = w_row_mul; a.out_bound = out_bound;
  a.out_absmax = (unsigned*)out_absmax;
  const int ct = pvn3d_ceil_div(n_out, SG_T);
  const dim3 grid(ct, pvn3d_ceil_div(n_points, SG_T));
  if (ar == 1 && form == 0 && z &&
      (double)pvn3d_ceil_div(n_points, z_points_per_frame) * z_rows_per_frame * ldz * 4.0 >= 4294967296.0)
    form = 1;                                        // (the LDS-DMA kernel addresses the table with 32-bit byte offsets)
  if (ar == 1 && form == 0) {
    // LDS-DMA form: 256-point tiles when they still give every CU its two workgroups, 128-point tiles otherwise
    const bool wide = (long long)ct * pvn3d_ceil_div(n_points, 256) >= 512;
    const dim3 gw(ct, pvn3d_ceil_div(n_points, 256));
    const size_t lw = 3 * (SG_T + 256) * 64 + 1024, ln = 3 * (SG_T + 128) * 64 + 1024;
    if (wide) {
      int e = z ? pvn3d_allow_big_lds(sg_gemm_dma_kernel<4, true>) : pvn3d_allow_big_lds(sg_gemm_dma_kernel<4, false>);
      if (e) return e;
      if (z) hipLaunchKernelGGL((sg_gemm_dma_kernel<4, true>), gw, dim3(256), lw, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((sg_gemm_dma_kernel<4, false>), gw, dim3(256), lw, (hipStream_t)stream, a);
    } else {
      if (z) hipLaunchKernelGGL((sg_gemm_dma_kernel<2, true>), grid, dim3(256), ln, (hipStream_t)stream, a);
      else hipLaunchKernelGGL((sg_gemm_dma_kernel<2, false>), grid, dim3(256), ln, (hipStream_t)stream, a);
    }
  } else if (ar == 1) {
    hipLaunchKernelGGL(sg_gemm_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, a);
  } else {
    hipLaunchKernelGGL(sg_gemm_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}
extern "C" int pvn3d_split_gemm(int n_points, int n_out, int slabs, const void* x_s16, const void* w_s16,
                                const float* bias_padded, int relu, const float* z, int ldz, int z_points_per_frame,
                                int z_rows_per_frame, const int* idx, const float* weight, float* out_f32, int ld_out,
                                void* out_s16, int slabs_out, void* stream) {
  return sg_gemm_any(0, 0, n_points, n_out, slabs, x_s16, w_s16, bias_padded, relu, z, ldz, z_points_per_frame, z_rows_per_frame,
                     idx, weight, out_f32, ld_out, out_s16, slabs_out, nullptr, 1.f, nullptr, nullptr, nullptr, stream);
}
extern "C" int pvn3d_split_gemm2(int n_points, int n_out, int slabs, const void* x_h16, const float* x_bound,
                                 const void* w_h16, float w_scale, const float* w_row_mul, const float* bias_padded,
                                 int relu, const float* z, int ldz, int z_points_per_frame, int z_rows_per_frame,
                                 const int* idx, const float* weight, float* out_f32, int ld_out, float* out_absmax,
                                 void* out_h16, int slabs_out, const float* out_bound, void* stream) {
  return sg_gemm_any(1, 0, n_points, n_out, slabs, x_h16, w_h16, bias_padded, relu, z, ldz, z_points_per_frame,
                     z_rows_per_frame, idx, weight, out_f32, ld_out, out_h16, slabs_out, x_bound, w_scale, w_row_mul, out_bound,
                     out_absmax, stream);
}
// The same product on the register-staged 128 x 128-tile kernel (bit-identical results; the cross-check of the LDS-DMA form).
extern "C" int pvn3d_split_gemm2_tile128(int n_points, int n_out, int slabs, const void* x_h16, const float* x_bound,
                                         const void* w_h16, float w_scale, const float* w_row_mul, const float* bias_padded,
                                         int relu, const float* z, int ldz, int z_points_per_frame, int z_rows_per_frame,
                                         const int* idx, const float* weight, float* out_f32, int ld_out, float* out_absmax,
                                         void* out_h16, int slabs_out, const float* out_bound, void* stream) {
  return sg_gemm_any(1, 1, n_points, n_out, slabs, x_h16, w_h16, bias_padded, relu, z, ldz, z_points_per_frame,
                     z_rows_per_frame, idx, weight, out_f32, ld_out, out_h16, slabs_out, x_bound, w_scale, w_row_mul, out_bound,
                     out_absmax, stream);
}
