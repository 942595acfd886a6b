// common.h -- shared helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// exclusive rank of this lane among the set bits of a wave64 ballot mask
__device__ __forceinline__ int pvn3d_mbcnt(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                   __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

__device__ __forceinline__ int pvn3d_lane() { return threadIdx.x & (PVN3D_WAVE - 1); }
