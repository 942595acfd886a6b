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
// "already done on device d" bit per kernel.
template <typename K>
static inline int pvn3d_allow_big_lds(K kern) {
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1ull)) return 0;
  hipFuncAttributes fa;
  e = hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern));
  if (e != hipSuccess) return (int)e;
  // the CU has 160 KiB; the kernel's static __shared__ arrays come out of the same budget
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                          160 * 1024 - (int)fa.sharedSizeBytes);
  if (e != hipSuccess) return (int)e;
  if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
  return 0;
}

// exclusive rank of this lane among the set bits of a wave64 ballot mask
__device__ __forceinline__ int pvn3d_mbcnt(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ int pvn3d_lane() { return threadIdx.x & (PVN3D_WAVE - 1); }
