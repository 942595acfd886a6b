// sg_fault_repro.hip -- stand-alone reproducer (no torch) for the split-GEMM epilogue fault of round 4
// (pvn3d_amd/csrc/split_gemm.hip, "epilogue" note).  The GEMM main loop is the library's; the
// epilogue is a template over variants of the gathered-add + store loop:
//   V0  gathers, arithmetic and stores of one (i, g) group interleaved -- what the compiler schedules by itself
//       (the form that returned wrong values in round 4)
//   V1  V0 + s_waitcnt vmcnt(0) between the gathers and their first use   (the round-4 work-around)
//   V2  V0 + s_waitcnt vmcnt(0) BEFORE the gathers are issued (stores drained first), compiler's waits after
//   V3  gathers as inline-asm loads into registers that do NOT overlap the pending stores' data / address registers
//       (kept alive across the loads), partial waits vmcnt(2/1/0) placed by hand like the compiler's
//   V4  V3 without the keep-alive (the register allocator is free to reuse the store operands)
//   V5  V1 with the three rows requested in the order 2, 1, 0 (does the fault follow the FIRST request or row 0?)
//   V6  V1 with a dummy request of row 0 ahead of the three real ones; the two copies of row 0 compared in the kernel
//   V7  V1 + a second, later read of row 0 compared with the first in the kernel (raw bits of both recorded)
//   V8  V1 with the launch configuration as compile-time constants (no uniform branches around the gathers / stores)
//   V10 V1 with dead points clamped instead of skipped (no divergent `continue`)
//   V11..V16  V1 with the row-0 product of channels 0 / 1 as one hand-written v_pk_mul_f32 directly behind the wait:
//        11 in place, op_sel:[0,1] (low half takes src1's HIGH register -- the compiler's form in V0..V4)
//        12 same, destination distinct from the sources      13 same as 11 with 16 idle cycles between wait and multiply
//        14 in place, no op_sel (src1 = {w0, w0})            15 in place, op_sel_hi:[1,0] (src1 = {w0, w2})
//        16 same as 11 with s_nop 1 behind it
//   V9  the library's two-pass epilogue (reference result)
// Every variant computes the same fp32 expression, so a correct run is bit-identical to V9.  For every mismatch the
// host decodes (lane, i, j, g, word) and checks which candidate explains the value (gathered row t read as zero, ...).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/sg_fault_repro.hip -o tools/sg_fault_repro.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
static inline int pvn3d_ceil_div(int a, int b) { return (a + b - 1) / b; }
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int SG_T = 128;                 // tile edge
constexpr int SG_ROWB = 208;              // LDS bytes per tile row: 2 slabs x 96 B + 16 B pad
constexpr int SG_OPB = SG_T * SG_ROWB;    // one operand's chunk

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
  unsigned* dbg;               // [0] = count, records of 12 words from word 16 on (variants 6, 7)
  int skip_loop;               // context sweep: no K loop at all (accumulators stay 0)
  int delay;                   // context sweep: idle time between the K loop and the epilogue, in units of s_sleep 127 (~8k cycles)
};

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

// grid (channel tiles, point tiles), remapped per XCD below.
template <int V>
__global__ __launch_bounds__(256, 2) void sg_repro_kernel(SgArgs a) {
  __shared__ __attribute__((aligned(16))) char smem[2 * SG_OPB];
  char* sW = smem;
  char* sX = smem + SG_OPB;
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
  const size_t rowb = (size_t)a.S * 96;              // bytes per s16 row
  const int nch = a.skip_loop ? 0 : (a.S >> 1);

  // chunk loads: 12 x 16 B per row and operand; thread -> (row, part) = ((tid + 256 j) / 12, (tid + 256 j) % 12)
  const char* gW[6];
  const char* gX[6];
  int lofs[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int id = tid + 256 * j, row = id / 12, part = id - row * 12;
    gW[j] = a.W + (size_t)(c0 + row) * rowb + part * 16;
    gX[j] = a.X + (size_t)min(p0 + row, a.P - 1) * rowb + part * 16;
    lofs[j] = row * SG_ROWB + part * 16;
  }
  u32x4 rW[6], rX[6];
#define SG_GLOAD(C)                                                        \
  _Pragma("unroll") for (int j = 0; j < 6; ++j) {                          \
    rW[j] = *reinterpret_cast<const u32x4*>(gW[j] + (size_t)(C) * 192);    \
    rX[j] = *reinterpret_cast<const u32x4*>(gX[j] + (size_t)(C) * 192);    \
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: row (lane & 31) of a 32-row block, k half (lane >> 5); + slab * 96 + piece * 32
  const char* fW = sW + (wr * 64 + (lane & 31)) * SG_ROWB + (lane >> 5) * 16;
  const char* fX = sX + (wc * 64 + (lane & 31)) * SG_ROWB + (lane >> 5) * 16;

  if (nch > 0) { SG_GLOAD(0) }
  for (int c = 0; c < nch; ++c) {
    __syncthreads();                                  // the previous chunk's fragment reads are done
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      *reinterpret_cast<u32x4*>(sW + lofs[j]) = rW[j];
      *reinterpret_cast<u32x4*>(sX + lofs[j]) = rX[j];
    }
    __syncthreads();
    if (c + 1 < nch) { SG_GLOAD(c + 1) }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 fa[2][3], fb[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          fa[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(fW + i * 32 * SG_ROWB + s * 96 + p * 32));
          fb[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(fX + i * 32 * SG_ROWB + s * 96 + p * 32));
        }
#define SG_MM(PA, PB)                                                                              \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)      \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA], fb[j][PB], acc[i][j], 0, 0, 0)
      SG_MM(0, 2); SG_MM(2, 0); SG_MM(1, 1);
      SG_MM(0, 1); SG_MM(1, 0); SG_MM(0, 0);
#undef SG_MM
    }
  }
#undef SG_GLOAD


  if (nch > 0 || a.skip_loop) { for (int d = 0; d < a.delay; ++d) __builtin_amdgcn_s_sleep(127); }
  const int half = lane >> 5;
  if (V == 9) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = p0 + wc * 64 + j * 32 + (lane & 31);
      const bool live = p < a.P;
      const int pc = live ? p : a.P - 1;
      if (a.Z) {
        const int f = pc / a.zn;
        const float* zbase = a.Z + (size_t)f * a.zm * a.ldz;
        const float* zr[3];
        float zw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          zr[t] = zbase + (size_t)a.idx[(size_t)pc * 3 + t] * a.ldz + c0 + wr * 64 + 4 * half;
          zw[t] = a.wgt[(size_t)pc * 3 + t];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float4 z[4][3];
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int t = 0; t < 3; ++t) z[g][t] = *reinterpret_cast<const float4*>(zr[t] + i * 32 + 8 * g);
          __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            acc[i][j][4 * g + 0] += z[g][0].x * zw[0] + z[g][1].x * zw[1] + z[g][2].x * zw[2];
            acc[i][j][4 * g + 1] += z[g][0].y * zw[0] + z[g][1].y * zw[1] + z[g][2].y * zw[2];
            acc[i][j][4 * g + 2] += z[g][0].z * zw[0] + z[g][1].z * zw[1] + z[g][2].z * zw[2];
            acc[i][j][4 * g + 3] += z[g][0].w * zw[0] + z[g][1].w * zw[1] + z[g][2].w * zw[2];
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
      if (!live) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = c0 + wr * 64 + i * 32 + 8 * g + 4 * half;
          const float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          if (a.out_f) *reinterpret_cast<float4*>(a.out_f + (size_t)p * a.ld_out + ch) = make_float4(v[0], v[1], v[2], v[3]);
          if (a.out_s && ch < 16 * a.S_out) {
            uint2 h, m, l;
            sg_split4(v, h, m, l);
            char* o = a.out_s + ((size_t)p * a.S_out + (ch >> 4)) * 96 + (ch & 15) * 2;
            *reinterpret_cast<uint2*>(o) = h;
            *reinterpret_cast<uint2*>(o + 32) = m;
            *reinterpret_cast<uint2*>(o + 64) = l;
          }
        }
    }
    return;
  }
  // interleaved forms
  constexpr bool CT = (V == 8);            // V8: the launch configuration (Z, bias, relu, s16 output only) as compile-time facts
  uint2 ph = {0, 0}, pm_ = {0, 0}, pl = {0, 0};
  char* po = nullptr;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int p = p0 + wc * 64 + j * 32 + (lane & 31);
    const bool live = p < a.P;
    if (V == 10) { p = live ? p : a.P - 1; } else if (!live) continue;
    const float* zr[3] = {nullptr, nullptr, nullptr};
    float zw[3] = {0.f, 0.f, 0.f};
    if (CT || a.Z) {
      const int f = p / a.zn;
      const float* zbase = a.Z + (size_t)f * a.zm * a.ldz;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        zr[t] = zbase + (size_t)a.idx[(size_t)p * 3 + t] * a.ldz + c0 + wr * 64 + 4 * half;
        zw[t] = a.wgt[(size_t)p * 3 + t];
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = c0 + wr * 64 + i * 32 + 8 * g + 4 * half;
        float v[4] = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (CT || a.Z) {
          f32x4 z0, z1, z2;
          if (V == 2) __builtin_amdgcn_s_waitcnt(0x0f70);
          if (V == 3 || V == 4) {
            const float* q0 = zr[0] + i * 32 + 8 * g;
            const float* q1 = zr[1] + i * 32 + 8 * g;
            const float* q2 = zr[2] + i * 32 + 8 * g;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z0) : "v"(q0) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z1) : "v"(q1) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z2) : "v"(q2) : "memory");
            if (V == 3) asm volatile("" :: "v"(u32x2{ph.x, ph.y}), "v"(u32x2{pm_.x, pm_.y}), "v"(u32x2{pl.x, pl.y}), "v"(po));
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(z0) :: "memory");
            asm volatile("s_waitcnt vmcnt(1)" : "+v"(z1) :: "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(z2) :: "memory");
          } else if (V == 5) {               // rows requested in the order 2, 1, 0
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z2) : "v"(zr[2] + i * 32 + 8 * g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z1) : "v"(zr[1] + i * 32 + 8 * g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z0) : "v"(zr[0] + i * 32 + 8 * g) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(z0), "+v"(z1), "+v"(z2) :: "memory");
          } else if (V == 6) {               // a dummy request of row 0 ahead of the three real ones
            f32x4 zd;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(zd) : "v"(zr[0] + i * 32 + 8 * g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z0) : "v"(zr[0] + i * 32 + 8 * g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z1) : "v"(zr[1] + i * 32 + 8 * g) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(z2) : "v"(zr[2] + i * 32 + 8 * g) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(zd) :: "memory");
            if (a.dbg && (zd.x != z0.x || zd.y != z0.y || zd.z != z0.z || zd.w != z0.w)) {
              const unsigned slot = atomicAdd(a.dbg, 1u);
              if (slot < 4096) {
                unsigned* d = a.dbg + 16 + slot * 12;
                d[0] = p; d[1] = ch; d[2] = lane; d[3] = (unsigned)(i * 4 + g);
                d[4] = __float_as_uint(zd.x); d[5] = __float_as_uint(zd.y); d[6] = __float_as_uint(zd.z); d[7] = __float_as_uint(zd.w);
                d[8] = __float_as_uint(z0.x); d[9] = __float_as_uint(z0.y); d[10] = __float_as_uint(z0.z); d[11] = __float_as_uint(z0.w);
              }
            }
          } else if (V == 20) {
            // operand-position table: the output is computed with single-lane instructions (always right); beside it
            // nine packed forms are evaluated on the same live operands and checked in the kernel against single-lane
            // products.  dbg[form * 8 + quarter] counts wrong LOW halves, dbg[form * 8 + 4 + quarter] wrong HIGH halves.
            z0 = *reinterpret_cast<const f32x4*>(zr[0] + i * 32 + 8 * g);
            z1 = *reinterpret_cast<const f32x4*>(zr[1] + i * 32 + 8 * g);
            z2 = *reinterpret_cast<const f32x4*>(zr[2] + i * 32 + 8 * g);
            __builtin_amdgcn_s_waitcnt(0x0f70);
            const f32x2 zxy = {z0.x, z0.y}, wp = {zw[2], zw[0]}, cp = {zw[1], z1.x};
            const float ws_lo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zw[2])));
            const float ws_hi = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, zw[0])));
            f32x2 r[9];
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r[0]) : "v"(zxy), "v"(wp));                     // src1 hi -> low half
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r[1]) : "v"(wp), "v"(zxy));                     // src0 hi -> low half
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r[2]) : "v"(zxy), "v"(zxy), "v"(cp));     // src2 hi -> low half
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r[3]) : "v"(zxy), "v"(wp));                  // src1 lo -> high half
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r[4]) : "v"(wp), "v"(zxy));                  // src0 lo -> high half
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r[5]) : "v"(zxy), "v"(zxy), "v"(cp));  // src2 lo -> high half
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r[6]) : "v"(zxy), "v"(wp));                     // add, src1 hi -> low
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r[7]) : "v"(zxy), "v"(wp), "v"(cp));      // fma, src1 hi -> low
            {
              const f32x2 wsg = {ws_lo, ws_hi};
              asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r[8]) : "v"(zxy), "s"(wsg));                  // src1 = SGPR pair, hi -> low
            }
#define SG_MUL(A, B) ({ float r_; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r_) : "v"(A), "v"(B)); r_; })
#define SG_ADD(A, B) ({ float r_; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r_) : "v"(A), "v"(B)); r_; })
#define SG_FMA(A, B, C) ({ float r_; asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r_) : "v"(A), "v"(B), "v"(C)); r_; })
            const float e_lo[9] = {SG_MUL(zxy.x, wp.y), SG_MUL(wp.y, zxy.x), SG_FMA(zxy.x, zxy.x, cp.y), SG_MUL(zxy.x, wp.x), SG_MUL(wp.x, zxy.x),
                                   SG_FMA(zxy.x, zxy.x, cp.x), SG_ADD(zxy.x, wp.y), SG_FMA(zxy.x, wp.y, cp.x), SG_MUL(zxy.x, ws_hi)};
            const float e_hi[9] = {SG_MUL(zxy.y, wp.y), SG_MUL(wp.y, zxy.y), SG_FMA(zxy.y, zxy.y, cp.y), SG_MUL(zxy.y, wp.x), SG_MUL(wp.x, zxy.y),
                                   SG_FMA(zxy.y, zxy.y, cp.x), SG_ADD(zxy.y, wp.y), SG_FMA(zxy.y, wp.y, cp.y), SG_MUL(zxy.y, ws_hi)};
            if (a.dbg) {
#pragma unroll
              for (int fo = 0; fo < 9; ++fo) {
                if (__float_as_uint(r[fo].x) != __float_as_uint(e_lo[fo])) atomicAdd(a.dbg + 64 + fo * 8 + (lane >> 4), 1u);
                if (__float_as_uint(r[fo].y) != __float_as_uint(e_hi[fo])) atomicAdd(a.dbg + 64 + fo * 8 + 4 + (lane >> 4), 1u);
              }
            }
            v[0] = SG_ADD(v[0], SG_ADD(SG_ADD(SG_MUL(z0.x, zw[0]), SG_MUL(z1.x, zw[1])), SG_MUL(z2.x, zw[2])));
            v[1] = SG_ADD(v[1], SG_ADD(SG_ADD(SG_MUL(z0.y, zw[0]), SG_MUL(z1.y, zw[1])), SG_MUL(z2.y, zw[2])));
            v[2] = SG_ADD(v[2], SG_ADD(SG_ADD(SG_MUL(z0.z, zw[0]), SG_MUL(z1.z, zw[1])), SG_MUL(z2.z, zw[2])));
            v[3] = SG_ADD(v[3], SG_ADD(SG_ADD(SG_MUL(z0.w, zw[0]), SG_MUL(z1.w, zw[1])), SG_MUL(z2.w, zw[2])));
            z0 = f32x4{0.f, 0.f, 0.f, 0.f}; z1 = z0; z2 = z0;
          } else if (V >= 11 && V <= 16) {
            // the row-0 product of channels 0 / 1 as ONE hand-written packed instruction (the rest of the expression in
            // single-lane inline asm, so the compiler packs nothing of it)
            z0 = *reinterpret_cast<const f32x4*>(zr[0] + i * 32 + 8 * g);
            z1 = *reinterpret_cast<const f32x4*>(zr[1] + i * 32 + 8 * g);
            z2 = *reinterpret_cast<const f32x4*>(zr[2] + i * 32 + 8 * g);
            f32x2 zxy = {z0.x, z0.y}, r;
            const f32x2 w20 = {zw[2], zw[0]}, w00 = {zw[0], zw[0]}, w02 = {zw[0], zw[2]};
            if (V == 11) { asm volatile("s_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %0, %0, %3 op_sel:[0,1]" : "+v"(zxy), "+v"(z1), "+v"(z2) : "v"(w20)); r = zxy; }
            if (V == 12) { asm volatile("s_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %0, %1, %4 op_sel:[0,1]" : "=&v"(r), "+v"(zxy), "+v"(z1), "+v"(z2) : "v"(w20)); }
            if (V == 13) { asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\tv_pk_mul_f32 %0, %0, %3 op_sel:[0,1]" : "+v"(zxy), "+v"(z1), "+v"(z2) : "v"(w20)); r = zxy; }
            if (V == 14) { asm volatile("s_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %0, %0, %3" : "+v"(zxy), "+v"(z1), "+v"(z2) : "v"(w00)); r = zxy; }
            if (V == 15) { asm volatile("s_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %0, %0, %3 op_sel_hi:[1,0]" : "+v"(zxy), "+v"(z1), "+v"(z2) : "v"(w02)); r = zxy; }
            if (V == 16) { asm volatile("s_waitcnt vmcnt(0)\n\tv_pk_mul_f32 %0, %0, %3 op_sel:[0,1]\n\ts_nop 1" : "+v"(zxy), "+v"(z1), "+v"(z2) : "v"(w20)); r = zxy; }
            // everything else of the expression as single-lane instructions the compiler cannot pack
            v[0] = SG_ADD(v[0], SG_ADD(SG_ADD(r.x, SG_MUL(z1.x, zw[1])), SG_MUL(z2.x, zw[2])));
            v[1] = SG_ADD(v[1], SG_ADD(SG_ADD(r.y, SG_MUL(z1.y, zw[1])), SG_MUL(z2.y, zw[2])));
            v[2] = SG_ADD(v[2], SG_ADD(SG_ADD(SG_MUL(z0.z, zw[0]), SG_MUL(z1.z, zw[1])), SG_MUL(z2.z, zw[2])));
            v[3] = SG_ADD(v[3], SG_ADD(SG_ADD(SG_MUL(z0.w, zw[0]), SG_MUL(z1.w, zw[1])), SG_MUL(z2.w, zw[2])));
            z0 = f32x4{0.f, 0.f, 0.f, 0.f}; z1 = z0; z2 = z0;      // the common expression below adds exact zeros
          } else {
            z0 = *reinterpret_cast<const f32x4*>(zr[0] + i * 32 + 8 * g);
            z1 = *reinterpret_cast<const f32x4*>(zr[1] + i * 32 + 8 * g);
            z2 = *reinterpret_cast<const f32x4*>(zr[2] + i * 32 + 8 * g);
            if (V != 0 && V != 2) __builtin_amdgcn_s_waitcnt(0x0f70);
            if (V == 7 && a.dbg) {            // what the first row's registers hold right after the wait
              const float* q = zr[0] + i * 32 + 8 * g;
              const f32x4 again = *reinterpret_cast<const volatile f32x4*>(q);     // a second, later read of the same 16 bytes
              if (again.x != z0.x || again.y != z0.y || again.z != z0.z || again.w != z0.w) {
                const unsigned slot = atomicAdd(a.dbg, 1u);
                if (slot < 4096) {
                  unsigned* d = a.dbg + 16 + slot * 12;
                  d[0] = p; d[1] = ch; d[2] = lane; d[3] = (unsigned)(i * 4 + g);
                  d[4] = __float_as_uint(again.x); d[5] = __float_as_uint(again.y); d[6] = __float_as_uint(again.z); d[7] = __float_as_uint(again.w);
                  d[8] = __float_as_uint(z0.x); d[9] = __float_as_uint(z0.y); d[10] = __float_as_uint(z0.z); d[11] = __float_as_uint(z0.w);
                }
              }
            }
          }
          v[0] += z0.x * zw[0] + z1.x * zw[1] + z2.x * zw[2];
          v[1] += z0.y * zw[0] + z1.y * zw[1] + z2.y * zw[2];
          v[2] += z0.z * zw[0] + z1.z * zw[1] + z2.z * zw[2];
          v[3] += z0.w * zw[0] + z1.w * zw[1] + z2.w * zw[2];
        }
        if (CT || a.bias) {
          const float4 b = *reinterpret_cast<const float4*>(a.bias + ch);
          v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (CT || a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (V == 10 && !live) continue;
        if (!CT && a.out_f) *reinterpret_cast<float4*>(a.out_f + (size_t)p * a.ld_out + ch) = make_float4(v[0], v[1], v[2], v[3]);
        if (CT || (a.out_s && ch < 16 * a.S_out)) {
          uint2 h, m, l;
          sg_split4(v, h, m, l);
          char* o = a.out_s + ((size_t)p * a.S_out + (ch >> 4)) * 96 + (ch & 15) * 2;
          *reinterpret_cast<uint2*>(o) = h;
          *reinterpret_cast<uint2*>(o + 32) = m;
          *reinterpret_cast<uint2*>(o + 64) = l;
          ph = h; pm_ = m; pl = l; po = o;
        }
      }
  }
}

}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static uint32_t rng_state = 12345u;
static inline uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state; }
static inline float rndf() { return ((rnd() >> 8) & 0xffff) / 65536.0f - 0.5f; }
static inline uint16_t bf16_trunc(float x) { uint32_t u; memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
static inline float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }

// s16 rows with the exact three-piece split of random fp32 values
static void fill_s16(std::vector<uint16_t>& buf, size_t rows, int S, float scale) {
  buf.resize(rows * S * 48);
  for (size_t r = 0; r < rows; ++r)
    for (int s = 0; s < S; ++s)
      for (int k = 0; k < 16; ++k) {
        float x = rndf() * scale;
        uint16_t h = bf16_trunc(x); float r1 = x - bf16_f(h);
        uint16_t m = bf16_trunc(r1); float r2 = r1 - bf16_f(m);
        uint16_t l = bf16_trunc(r2);
        uint16_t* o = &buf[(r * S + s) * 48];
        o[k] = h; o[16 + k] = m; o[32 + k] = l;
      }
}

// Z as the output of kernels on the same stream (like the library: Z = Wa . known is the previous launch): a zero fill,
// then a float4 copy of the real rows.  A gather that sees the zero fill instead of the copy is a visibility problem
// between launches, not a wait-count problem inside the kernel.
__global__ void fill_zero_kernel(float4* p, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void copy_kernel(float4* dst, const float4* src, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) dst[i] = src[i];
}

static int g_dyn_lds = 0;      // context sweep: extra dynamic LDS per workgroup (limits the workgroups per CU)
template <int V>
static void launch(const SgArgs& a) {
  const dim3 grid(pvn3d_ceil_div(a.N, SG_T), pvn3d_ceil_div(a.P, SG_T));
  if (g_dyn_lds) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sg_repro_kernel<V>), hipFuncAttributeMaxDynamicSharedMemorySize, g_dyn_lds));
  hipLaunchKernelGGL(sg_repro_kernel<V>, grid, dim3(256), g_dyn_lds, 0, a);
}
static void launch_v(int v, const SgArgs& a) {
  switch (v) {
    case 0: launch<0>(a); break; case 1: launch<1>(a); break; case 2: launch<2>(a); break;
    case 3: launch<3>(a); break; case 4: launch<4>(a); break; case 5: launch<5>(a); break; case 6: launch<6>(a); break;
    case 7: launch<7>(a); break; case 8: launch<8>(a); break; case 10: launch<10>(a); break;
    case 11: launch<11>(a); break; case 12: launch<12>(a); break; case 13: launch<13>(a); break; case 14: launch<14>(a); break;
    case 15: launch<15>(a); break; case 16: launch<16>(a); break; case 20: launch<20>(a); break; default: launch<9>(a); break;
  }
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 64;
  const int reps = argc > 2 ? atoi(argv[2]) : 6;
  const bool quick = argc > 3 && atoi(argv[3]) != 0;     // only the "s16 only" configuration
  const int zn = 1024, zm = 512, N = 512, S = 16, S_out = 32, ldz = 512;
  const int P = frames * zn;
  printf("sg_fault_repro: P=%d N=%d K=%d, Z %d x %d rows, reps %d\n", P, N, S * 16, frames, zm, reps);
  std::vector<uint16_t> hX, hW;
  fill_s16(hX, P, S, 2.0f);
  fill_s16(hW, N, S, 0.25f);
  std::vector<float> hZ((size_t)frames * zm * ldz), hwgt((size_t)P * 3), hbias(N);
  std::vector<int> hidx((size_t)P * 3);
  for (auto& z : hZ) { z = rndf() * 4.0f; if (z == 0.f) z = 0.123f; }        // no exact zeros in Z
  for (auto& w : hwgt) w = 0.1f + (rndf() + 0.5f) * 0.8f;
  for (auto& b : hbias) b = rndf();
  for (auto& i : hidx) i = (int)(rnd() >> 8) % zm;
  char *dX, *dW, *dS, *dSref; float *dZ, *dwgt, *dbias, *dF, *dFref, *dAcc; int* didx;
  const size_t sbytes = (size_t)P * S_out * 96, fbytes = (size_t)P * N * 4;
  CK(hipMalloc(&dX, hX.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2));
  CK(hipMalloc(&dZ, hZ.size() * 4)); CK(hipMalloc(&dwgt, hwgt.size() * 4)); CK(hipMalloc(&dbias, N * 4));
  CK(hipMalloc(&didx, hidx.size() * 4));
  CK(hipMalloc(&dS, sbytes)); CK(hipMalloc(&dSref, sbytes)); CK(hipMalloc(&dF, fbytes)); CK(hipMalloc(&dFref, fbytes));
  CK(hipMalloc(&dAcc, fbytes));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dZ, hZ.data(), hZ.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwgt, hwgt.data(), hwgt.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
  SgArgs base = {};
  base.P = P; base.N = N; base.S = S; base.X = dX; base.W = dW; base.bias = dbias; base.relu = 1;
  base.Z = dZ; base.ldz = ldz; base.zn = zn; base.zm = zm; base.idx = didx; base.wgt = dwgt; base.ld_out = N; base.S_out = S_out;
  // the bare accumulators (no Z, bias, relu) and the reference result, both through the two-pass epilogue
  SgArgs aacc = base; aacc.Z = nullptr; aacc.bias = nullptr; aacc.relu = 0; aacc.out_f = dAcc;
  launch<9>(aacc);
  SgArgs aref = base; aref.out_f = dFref; aref.out_s = dSref;
  launch<9>(aref);
  CK(hipDeviceSynchronize());
  std::vector<float> hAcc((size_t)P * N), hFref((size_t)P * N), hF((size_t)P * N);
  std::vector<uint16_t> hSref(sbytes / 2), hS(sbytes / 2);
  CK(hipMemcpy(hAcc.data(), dAcc, fbytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hFref.data(), dFref, fbytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hSref.data(), dSref, sbytes, hipMemcpyDeviceToHost));
  // host check of the reference expression on a sample (fp32, one rounding per operation)
  {
    long bad = 0;
    for (int s = 0; s < 200000; ++s) {
      const size_t p = rnd() % P; const int ch = rnd() % N; const int f = (int)(p / zn);
      float zt[3];
      for (int t = 0; t < 3; ++t) zt[t] = hZ[((size_t)f * zm + hidx[p * 3 + t]) * ldz + ch] * hwgt[p * 3 + t];
      volatile float s01 = zt[0] + zt[1]; volatile float s012 = s01 + zt[2];
      volatile float v = hAcc[p * N + ch] + s012; v = v + hbias[ch]; float r = v > 0.f ? v : 0.f;
      if (r != hFref[p * N + ch]) ++bad;
    }
    printf("reference epilogue vs host expression on 200000 samples: %ld mismatches\n", bad);
  }
  unsigned* dDbg; CK(hipMalloc(&dDbg, (16 + 4096 * 12) * 4));
  std::vector<unsigned> hDbg(16 + 4096 * 12);
  if (argc > 3 && atoi(argv[3]) == 2) {
    // context sweep on V11 (hand-written v_pk_mul_f32 op_sel:[0,1] behind a full wait): what has to be around it?
    struct Ctx { const char* name; int dyn_lds, skip_loop, delay, pts; } ctx[] = {
      {"as in the library (2 workgroups per CU)", 0, 0, 0, P},
      {"1 workgroup per CU (60 KB extra LDS)", 60 * 1024, 0, 0, P},
      {"K loop skipped (epilogue only)", 0, 1, 0, P},
      {"K loop skipped, 1 workgroup per CU", 60 * 1024, 1, 0, P},
      {"~80k idle cycles between K loop and epilogue", 0, 0, 10, P},
      {"first 16384 points only", 0, 0, 0, 16384},
      {"first 4096 points only", 0, 0, 0, 4096},
    };
    SgArgs r = base; r.out_s = dSref; r.out_f = nullptr;
    for (auto& c : ctx) {
      long tot = 0, zeros_q[4] = {0, 0, 0, 0};
      for (int v : {9, 11}) {
        for (int rep = 0; rep < reps; ++rep) {
          SgArgs a = base; a.P = c.pts; a.skip_loop = c.skip_loop; a.delay = c.delay; a.out_s = v == 9 ? dSref : dS; a.out_f = nullptr;
          g_dyn_lds = c.dyn_lds;
          CK(hipMemset(a.out_s, 0xff, sbytes));
          launch_v(v, a);
          CK(hipDeviceSynchronize());
          if (v == 9) { CK(hipMemcpy(hSref.data(), dSref, sbytes, hipMemcpyDeviceToHost)); break; }
          CK(hipMemcpy(hS.data(), dS, sbytes, hipMemcpyDeviceToHost));
          const size_t words = (size_t)c.pts * S_out * 48;
          for (size_t pnt = 0; pnt < (size_t)c.pts; ++pnt)
            for (int ch = 0; ch < N; ++ch) {
              const size_t o = (pnt * S_out + (ch >> 4)) * 48 + (ch & 15);
              if (hS[o] != hSref[o] || hS[o + 16] != hSref[o + 16] || hS[o + 32] != hSref[o + 32]) {
                ++tot;
                const int cc = ch % 128, pp = (int)(pnt % 128);
                zeros_q[((pp % 32) + 32 * ((cc % 8) / 4)) >> 4]++;
              }
            }
          (void)words;
        }
      }
      g_dyn_lds = 0;
      printf("V11, %-48s: %ld wrong values in %d runs, by lane/16 [%ld %ld %ld %ld]\n", c.name, tot, reps, zeros_q[0], zeros_q[1], zeros_q[2], zeros_q[3]);
      fflush(stdout);
    }
    return 0;
  }
  if (argc > 3 && atoi(argv[3]) == 3) {
    // operand-position table (V20): which packed forms go wrong in the triggering context (2 workgroups per CU)?
    static const char* names[9] = {"v_pk_mul_f32  src1 hi->lo  op_sel:[0,1]", "v_pk_mul_f32  src0 hi->lo  op_sel:[1,0]",
                                   "v_pk_fma_f32  src2 hi->lo  op_sel:[0,0,1]", "v_pk_mul_f32  src1 lo->hi  op_sel_hi:[1,0]",
                                   "v_pk_mul_f32  src0 lo->hi  op_sel_hi:[0,1]", "v_pk_fma_f32  src2 lo->hi  op_sel_hi:[1,1,0]",
                                   "v_pk_add_f32  src1 hi->lo  op_sel:[0,1]", "v_pk_fma_f32  src1 hi->lo  op_sel:[0,1,0]",
                                   "v_pk_mul_f32  src1 = SGPR pair, hi->lo"};
    unsigned long long tot[9][8] = {};
    long out_bad = 0;
    for (int rep = 0; rep < reps; ++rep) {
      SgArgs a = base; a.out_s = dS; a.out_f = nullptr; a.dbg = dDbg;
      CK(hipMemset(dDbg, 0, (64 + 9 * 8) * 4));
      CK(hipMemset(dS, 0xff, sbytes));
      launch_v(20, a);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(hDbg.data(), dDbg, (64 + 9 * 8) * 4, hipMemcpyDeviceToHost));
      for (int f = 0; f < 9; ++f) for (int k = 0; k < 8; ++k) tot[f][k] += hDbg[64 + f * 8 + k];
      CK(hipMemcpy(hS.data(), dS, sbytes, hipMemcpyDeviceToHost));
      out_bad += memcmp(hS.data(), hSref.data(), sbytes) != 0;
    }
    printf("V20: %d launches, %.3g executions of each form; output (single-lane arithmetic) differs from the reference in %ld launches\n",
           reps, (double)reps * P * 128.0 * 64 / 64, out_bad);
    for (int f = 0; f < 9; ++f)
      printf("  %-46s wrong low halves by lane/16 [%llu %llu %llu %llu]   wrong high halves [%llu %llu %llu %llu]\n", names[f],
             tot[f][0], tot[f][1], tot[f][2], tot[f][3], tot[f][4], tot[f][5], tot[f][6], tot[f][7]);
    return 0;
  }
  const int variants[] = {9, 0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 14, 15, 16};
  const int n_variants = 17;
  const char* modes[] = {"s16 only", "fp32 only", "both", "s16 only, Z written by the preceding launches"};
  float* dZsrc;
  CK(hipMalloc(&dZsrc, hZ.size() * 4));
  CK(hipMemcpy(dZsrc, dZ, hZ.size() * 4, hipMemcpyDeviceToDevice));
  for (int cfg = 0; cfg < 4; ++cfg)
    for (int vi = 0; vi < n_variants; ++vi) {
      const int mode = cfg == 3 ? 0 : cfg;
      if (variants[vi] == 8 && mode != 0) continue;
      if (quick && cfg != 0) continue;
      const int v = variants[vi];
      long tot_bad = 0, runs_bad = 0;
      long by_lane16[4] = {0}, by_word[4] = {0}, by_ig[8] = {0}, by_j[2] = {0}, by_piece[3] = {0};
      long expl[8] = {0};   // 0..2: row t read as 0; 3: all three rows 0; 4: equals acc path w/o bias...; 7: unexplained
      for (int rep = 0; rep < reps; ++rep) {
        SgArgs a = base;
        a.out_s = (mode == 0 || mode == 2) ? dS : nullptr;
        a.out_f = (mode == 1 || mode == 2) ? dF : nullptr;
        CK(hipMemset(dS, 0xff, sbytes)); CK(hipMemset(dF, 0xff, fbytes));
        CK(hipMemset(dDbg, 0, 64)); a.dbg = dDbg;
        if (cfg == 3) {
          CK(hipDeviceSynchronize());
          hipLaunchKernelGGL(fill_zero_kernel, dim3(2048), dim3(256), 0, 0, (float4*)dZ, hZ.size() / 4);
          hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, 0, (float4*)dZ, (const float4*)dZsrc, hZ.size() / 4);
        }
        launch_v(v, a);
        CK(hipDeviceSynchronize());
        long bad = 0;
        if (a.out_f) {
          CK(hipMemcpy(hF.data(), dF, fbytes, hipMemcpyDeviceToHost));
          if (memcmp(hF.data(), hFref.data(), fbytes) != 0)
            for (size_t p = 0; p < (size_t)P; ++p)
              for (int ch = 0; ch < N; ++ch) {
                const float got = hF[p * N + ch];
                if (memcmp(&got, &hFref[p * N + ch], 4) == 0) continue;
                ++bad;
                // lane of the value: point block (p % 32), half = (ch % 8) / 4
                const int c = ch % 128, pp = (int)(p % 128);
                const int hf = (c % 8) / 4, lane = (pp % 32) + 32 * hf, e = c % 4, g = (c % 32) / 8, i = (c % 64) / 32, j = (pp % 64) / 32;
                by_lane16[lane >> 4]++; by_word[e]++; by_ig[i * 4 + g]++; by_j[j]++;
                const int f = (int)(p / zn);
                float zt[3];
                for (int t = 0; t < 3; ++t) zt[t] = hZ[((size_t)f * zm + hidx[p * 3 + t]) * ldz + ch] * hwgt[p * 3 + t];
                bool done = false;
                for (int drop = 0; drop < 4 && !done; ++drop) {
                  float q[3] = {zt[0], zt[1], zt[2]};
                  if (drop < 3) q[drop] = 0.f * hwgt[p * 3 + drop]; else q[0] = q[1] = q[2] = 0.f;
                  volatile float s01 = q[0] + q[1]; volatile float s012 = s01 + q[2];
                  volatile float vv = hAcc[p * N + ch] + s012; vv = vv + hbias[ch]; const float r = vv > 0.f ? vv : 0.f;
                  if (r == got) { expl[drop]++; done = true; }
                }
                if (!done) {
                  expl[7]++;
                  if (expl[7] <= 5) printf("    unexplained: p=%zu ch=%d got=%g want=%g acc=%g z*w=(%g,%g,%g) bias=%g\n", p, ch, got,
                                           hFref[p * N + ch], hAcc[p * N + ch], zt[0], zt[1], zt[2], hbias[ch]);
                }
              }
        }
        if (a.out_s) {
          CK(hipMemcpy(hS.data(), dS, sbytes, hipMemcpyDeviceToHost));
          if (memcmp(hS.data(), hSref.data(), sbytes) != 0)
            for (size_t p = 0; p < (size_t)P; ++p)
              for (int ch = 0; ch < N; ++ch) {
                const size_t o = (p * S_out + (ch >> 4)) * 48 + (ch & 15);
                int pieces_bad = 0;
                for (int pc = 0; pc < 3; ++pc) if (hS[o + 16 * pc] != hSref[o + 16 * pc]) { ++pieces_bad; by_piece[pc]++; }
                if (!pieces_bad) continue;
                if (a.out_f) continue;         // the fp32 pass above already classified this value
                ++bad;
                const float got = bf16_f(hS[o]) + bf16_f(hS[o + 16]) + bf16_f(hS[o + 32]);
                const int c = ch % 128, pp = (int)(p % 128);
                const int hf = (c % 8) / 4, lane = (pp % 32) + 32 * hf, e = c % 4, g = (c % 32) / 8, i = (c % 64) / 32, j = (pp % 64) / 32;
                by_lane16[lane >> 4]++; by_word[e]++; by_ig[i * 4 + g]++; by_j[j]++;
                const int f = (int)(p / zn);
                float zt[3];
                for (int t = 0; t < 3; ++t) zt[t] = hZ[((size_t)f * zm + hidx[p * 3 + t]) * ldz + ch] * hwgt[p * 3 + t];
                bool done = false;
                for (int drop = 0; drop < 4 && !done; ++drop) {
                  float q[3] = {zt[0], zt[1], zt[2]};
                  if (drop < 3) q[drop] = 0.f; else q[0] = q[1] = q[2] = 0.f;
                  volatile float s01 = q[0] + q[1]; volatile float s012 = s01 + q[2];
                  volatile float vv = hAcc[p * N + ch] + s012; vv = vv + hbias[ch]; const float r = vv > 0.f ? vv : 0.f;
                  if (r == got) { expl[drop]++; done = true; }
                }
                if (!done) {
                  expl[7]++;
                  if (expl[7] <= 5) printf("    unexplained: p=%zu ch=%d got=%g want=%g acc=%g z*w=(%g,%g,%g) bias=%g pieces got %04x %04x %04x want %04x %04x %04x\n",
                                           p, ch, got, hFref[p * N + ch], hAcc[p * N + ch], zt[0], zt[1], zt[2], hbias[ch],
                                           hS[o], hS[o + 16], hS[o + 32], hSref[o], hSref[o + 16], hSref[o + 32]);
                }
              }
        }
        tot_bad += bad; runs_bad += bad != 0;
        if (v == 6 || v == 7) {
          CK(hipMemcpy(hDbg.data(), dDbg, hDbg.size() * 4, hipMemcpyDeviceToHost));
          const unsigned n = hDbg[0];
          if (n) printf("    V%d run %d: %u in-kernel disagreements between two reads of row 0 (%s); first records:\n", v, rep, n,
                        v == 6 ? "dummy request first, then the real one" : "a second read after the wait");
          for (unsigned r = 0; r < n && r < 6; ++r) {
            const unsigned* d = &hDbg[16 + r * 12];
            const size_t pp = d[0]; const int cc = (int)d[1]; const int ff = (int)(pp / zn);
            const float* zt = &hZ[((size_t)ff * zm + hidx[pp * 3 + 0]) * ldz + cc];
            unsigned tb[4]; memcpy(tb, zt, 16);
            printf("      p=%zu ch=%d lane=%u (i,g)=%u  %s=%08x %08x %08x %08x  first(z0)=%08x %08x %08x %08x  memory=%08x %08x %08x %08x\n", pp, cc, d[2], d[3],
                   v == 6 ? "dummy" : "again", d[4], d[5], d[6], d[7], d[8], d[9], d[10], d[11], tb[0], tb[1], tb[2], tb[3]);
          }
        }
      }
      printf("V%d [%s]: %ld wrong values in %d runs (%ld runs affected)", v, modes[cfg], tot_bad, reps, runs_bad);
      if (tot_bad)
        printf("\n    lane/16 [%ld %ld %ld %ld]  word [%ld %ld %ld %ld]  j [%ld %ld]  (i,g) [%ld %ld %ld %ld | %ld %ld %ld %ld]\n"
               "    explained by: row0 read as 0: %ld, row1: %ld, row2: %ld, all rows: %ld, unexplained: %ld; s16 pieces differing h/m/l [%ld %ld %ld]",
               by_lane16[0], by_lane16[1], by_lane16[2], by_lane16[3], by_word[0], by_word[1], by_word[2], by_word[3], by_j[0], by_j[1],
               by_ig[0], by_ig[1], by_ig[2], by_ig[3], by_ig[4], by_ig[5], by_ig[6], by_ig[7], expl[0], expl[1], expl[2], expl[3], expl[7],
               by_piece[0], by_piece[1], by_piece[2]);
      printf("\n");
      fflush(stdout);
    }
  return 0;
}
