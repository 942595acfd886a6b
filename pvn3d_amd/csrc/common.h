// common.h -- shared helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/pvn3d_hip.h"

#define PVN3D_WAVE 64

#define PVN3D_RETURN_IF_ERR(expr)          \
  do {                                     \
    hipError_t e__ = (expr);               \
    if (e__ != hipSuccess) return (int)e__; \
  } while (0)

#define PVN3D_LAUNCH_CHECK()                 \
  do {                                       \
    hipError_t e__ = hipGetLastError();      \
    if (e__ != hipSuccess) return (int)e__;  \
  } while (0)

static inline int pvn3d_ceil_div(int a, int b) { return (a + b - 1) / b; }

// One-time opt-in of a kernel to > 48 KiB of dynamic LDS (a property of the loaded function on a
// device, not of a launch).  The only process-wide state of the library is this idempotent
// "already done on device d" bit PER KERNEL ADDRESS: the cache is keyed on the function pointer's
// value (two instantiations of one template share a pointer TYPE, so a per-type static would let
// the first instantiation's opt-in mask the others').  Lock-free open-addressed table; a slot is
// claimed once with a CAS on its key and never released, a full table only costs the (idempotent)
// runtime calls again.
struct pvn3d_big_lds_slot {
  std::atomic<const void*> key;
  std::atomic<unsigned long long> done;
};
inline pvn3d_big_lds_slot* pvn3d_big_lds_table() {
  static pvn3d_big_lds_slot table[256];  // zero-initialised; shared by every TU of the library
  return table;
}
inline std::atomic<unsigned long long>* pvn3d_big_lds_find(const void* fn) {
  pvn3d_big_lds_slot* t = pvn3d_big_lds_table();
  size_t h = ((uintptr_t)fn >> 4) * 0x9E3779B97F4A7C15ull >> 56;
  for (int probe = 0; probe < 256; ++probe) {
    pvn3d_big_lds_slot& s = t[(h + probe) & 255];
    const void* k = s.key.load(std::memory_order_acquire);
    if (k == fn) return &s.done;
    if (k == nullptr) {
      const void* expect = nullptr;
      if (s.key.compare_exchange_strong(expect, fn, std::memory_order_acq_rel)) return &s.done;
      if (expect == fn) return &s.done;
    }
  }
  return nullptr;
}
template <typename K>
static inline int pvn3d_allow_big_lds(K kern) {
  const void* fn = reinterpret_cast<const void*>(kern);
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  std::atomic<unsigned long long>* done = pvn3d_big_lds_find(fn);
  if (done && dev < 64 && ((done->load(std::memory_order_acquire) >> dev) & 1ull)) return 0;
  hipFuncAttributes fa;
  e = hipFuncGetAttributes(&fa, fn);
  if (e != hipSuccess) return (int)e;
  // the CU has 160 KiB; the kernel's static __shared__ arrays come out of the same budget
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024 - (int)fa.sharedSizeBytes);
  if (e != hipSuccess) return (int)e;
  if (done && dev < 64) done->fetch_or(1ull << dev, std::memory_order_release);
  return 0;
}

// Two fp16 pieces of two (already scaled) fp32 values: h = fp16(x) (round to nearest), l = fp16(x - h) -- the residual is
// exact in fp32, |x - h - l| <= 2^-22 |x|.  The straightforward form costs five instructions per pair (v_cvt_pk_f16_f32,
// two v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32); v_fma_mixlo/hi_f16 read the fp16 halves of h in place, subtract
// in fp32 and round the result to fp16 in one instruction each: three per pair, the same bits (tools/split_ab.py: equal
// digests of pvn3d_split_rows2 over adversarial values and of a 16-frame forward against a -DPVN3D_SPLIT_PLAIN build,
// tools/split_ab.sh).  Measured effect on the chain kernels: none (sa_mlp 2.61-2.65 ms either way) -- which retires the
// round-5 reading that those kernels are bound by the COUNT of their vector instructions (the split was 40 % of them in
// the narrow-chain kernels); kept for the instructions it saves.
__device__ __forceinline__ void pvn3d_split2_f16(float x0, float x1, unsigned& h, unsigned& l) {
  typedef __attribute__((ext_vector_type(2))) _Float16 h2_t;
  typedef __attribute__((ext_vector_type(2))) float f2_t;
  const f2_t x = {x0, x1};
  const h2_t hh = __builtin_convertvector(x, h2_t);
  h = __builtin_bit_cast(unsigned, hh);
#ifdef PVN3D_SPLIT_PLAIN
  l = __builtin_bit_cast(unsigned, __builtin_convertvector(x - __builtin_convertvector(hh, f2_t), h2_t));
#else
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l)
      : "v"(h), "v"(x0), "v"(x1));
#endif
}

// exclusive rank of this lane among the set bits of a wave64 ballot mask
__device__ __forceinline__ int pvn3d_mbcnt(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ int pvn3d_lane() { return threadIdx.x & (PVN3D_WAVE - 1); }

// XCD-aware (frame, tile) of a workgroup in a (tiles, frames) grid.  Workgroups are dealt round-robin to the 8 XCDs in
// linear-id order, each XCD with its own 4 MB L2: with the plain (x = tile, y = frame) reading every XCD touches every
// frame's tables (and fetches its own copy of them from HBM); here XCD x works through frames x, x + 8, x + 16, ...
__device__ __forceinline__ void pvn3d_xcd_frame_map(int& frame, int& tile) {
  const int nb = gridDim.x, nf = gridDim.y;
  tile = blockIdx.x;
  frame = blockIdx.y;
  if ((nf & 7) == 0) {
    const unsigned lin = blockIdx.x + (unsigned)nb * blockIdx.y;
    const unsigned q = lin >> 3;
    frame = (int)(lin & 7) + 8 * (int)(q / nb);
    tile = (int)(q % nb);
  }
}

// Word fill as a kernel of this library.  Used instead of hipMemsetAsync wherever the call sequence may be captured into
// a HIP graph (Pointnet2MSG.graphed, GraphedFramePoses): the runtime's memset node was observed to land out of order
// with the kernels around it on replay; a kernel node is an ordinary link of the captured chain.
static __global__ __launch_bounds__(256) void pvn3d_fill_u32_kernel(unsigned* __restrict__ p, unsigned v, size_t n) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) p[i] = v;
}
static inline void pvn3d_fill_u32(void* p, unsigned v, size_t n_words, hipStream_t st) {
  if (n_words == 0) return;
  const size_t blocks = (n_words + 255) / 256;
  hipLaunchKernelGGL(pvn3d_fill_u32_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, st,
                     (unsigned*)p, v, n_words);
}
