// sampling.hip -- furthest point sampling + gather_points for gfx950.
//
// Replaces pvn3d/_ext-src/src/sampling_gpu.cu (reference): furthest_point_sampling_kernel
// (:69-173), gather_points_kernel (:8-20), gather_points_grad_kernel (:34-47).
//
// FPS design (one workgroup per cloud, the m-1 rounds are inherently serial):
//   * every point and its running min-distance live in VGPRs for the whole kernel
//     (PPT points per thread); global memory is read once.  The reference re-reads xyz and
//     temp from global memory every round.
//   * per round: VALU scan -> wave64 butterfly arg-max on a packed 64-bit key -> one LDS slot
//     per wave -> ONE barrier (double-buffered slots) -> winner coordinates from an LDS copy
//     of the cloud.  The reference uses 9 __syncthreads per round.
//   * the packed key reproduces the reference's tie-break exactly without emulating its
//     block: the reference's 512-thread strided scan + halving tree picks, among equal
//     distances, the smallest bit-reversed (k mod bs), then the smallest k
//     (sampling_gpu.cu:59-65,96-168, bs = opt_n_threads(n)).  key = (bits(d2) << 32) | ~prio
//     with prio(k) = bitrev(k mod bs) * ceil(n/bs) + k / bs, so ANY reduction order gives the
//     reference's winner.
// Arithmetic: this TU is compiled with -ffp-contract=off; d is evaluated as
// ((dx*dx + dy*dy) + dz*dz) with one rounding per operation (oracle/pvn3d_oracle.c).
#include "common.h"

namespace {

__device__ __forceinline__ long long shfl_xor_i64(long long v, int mask) {
  int lo = __shfl_xor((int)(v & 0xffffffffLL), mask, 64);
  int hi = __shfl_xor((int)(v >> 32), mask, 64);
  return ((long long)hi << 32) | (unsigned int)lo;
}

__device__ __forceinline__ long long wave_max_i64(long long v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    long long o = shfl_xor_i64(v, s);
    v = o > v ? o : v;
  }
  return v;
}

__device__ __forceinline__ unsigned fps_prio(int k, int L, int Q) {
  unsigned r = L ? (__brev((unsigned)k & ((1u << L) - 1u)) >> (32 - L)) : 0u;
  return r * (unsigned)Q + ((unsigned)k >> L);
}

__device__ __forceinline__ int fps_prio_to_k(unsigned p, int L, int Q) {
  unsigned r = p / (unsigned)Q, q = p % (unsigned)Q;
  unsigned t = L ? (__brev(r) >> (32 - L)) : 0u;
  return (int)((q << L) | t);
}

// mag <= 1e-3 with mag fp32 and the literal double (sampling_gpu.cu:100-101)
__device__ __forceinline__ bool fps_skipped(float x, float y, float z) {
  float mag = (x * x) + (y * y) + (z * z);
  return (double)mag <= 1e-3;
}

// Register-resident FPS.  THREADS in {64,256,1024}; PPT = points per thread.
// lds_xyz != 0: dynamic LDS holds an SoA copy of the cloud (3*n floats) for the winner fetch.
template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(int n, int m, int L, int Q,
                                                          int lds_xyz,
                                                          const float* __restrict__ dataset,
                                                          int* __restrict__ idxs) {
  constexpr int NW = THREADS / 64;
  extern __shared__ float s_dyn[];
  __shared__ long long s_slot[2][NW > 1 ? NW : 1];
  if (m <= 0) return;
  const int tid = threadIdx.x;
  dataset += (size_t)blockIdx.x * n * 3;
  idxs += (size_t)blockIdx.x * m;

  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
  unsigned nprio[PPT];  // ~prio
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = tid + i * THREADS;
    if (k < n) {
      px[i] = dataset[k * 3 + 0];
      py[i] = dataset[k * 3 + 1];
      pz[i] = dataset[k * 3 + 2];
      // skipped points never update temp and never win: temp = -inf makes d2 = -inf, whose
      // key is below the "no candidate" key -1.
      tmp[i] = fps_skipped(px[i], py[i], pz[i]) ? -__builtin_inff() : 1e10f;
      nprio[i] = ~fps_prio(k, L, Q);
      if (lds_xyz) {
        s_dyn[k] = px[i];
        s_dyn[n + k] = py[i];
        s_dyn[2 * n + k] = pz[i];
      }
    } else {
      px[i] = py[i] = pz[i] = 0.f;
      tmp[i] = -__builtin_inff();
      nprio[i] = 0u;
    }
  }
  int old = 0;
  if (tid == 0) idxs[0] = 0;
  float x1 = dataset[0], y1 = dataset[1], z1 = dataset[2];
  if (NW > 1 || lds_xyz) __syncthreads();

  for (int j = 1; j < m; ++j) {
    long long best = -1LL;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float d2 = fminf(d, tmp[i]);
      tmp[i] = d2;
      const long long key = ((long long)__float_as_int(d2) << 32) | (long long)nprio[i];
      best = key > best ? key : best;
    }
    best = wave_max_i64(best);
    if (NW > 1) {
      const int buf = j & 1;
      if ((tid & 63) == 0) s_slot[buf][tid >> 6] = best;
      __syncthreads();
      long long b2 = s_slot[buf][0];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        long long o = s_slot[buf][w];
        b2 = o > b2 ? o : b2;
      }
      best = b2;
    }
    // "no candidate": every thread kept (best=-1, besti=0) -> the tree returns index 0
    old = (best == -1LL) ? 0 : fps_prio_to_k(~(unsigned)(best & 0xffffffffLL), L, Q);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idxs[j] = old;
    if (lds_xyz) {
      x1 = s_dyn[old];
      y1 = s_dyn[n + old];
      z1 = s_dyn[2 * n + old];
    } else {
      x1 = dataset[old * 3 + 0];
      y1 = dataset[old * 3 + 1];
      z1 = dataset[old * 3 + 2];
    }
  }
}

// Any-n fallback: distances in the caller's `temp` scratch (global memory), 1024 threads.
__global__ __launch_bounds__(1024) void fps_global_kernel(int n, int m, int L, int Q,
                                                          const float* __restrict__ dataset,
                                                          float* __restrict__ temp,
                                                          int* __restrict__ idxs) {
  __shared__ long long s_slot[2][16];
  if (m <= 0) return;
  const int tid = threadIdx.x;
  dataset += (size_t)blockIdx.x * n * 3;
  temp += (size_t)blockIdx.x * n;
  idxs += (size_t)blockIdx.x * m;
  for (int k = tid; k < n; k += 1024)
    temp[k] = fps_skipped(dataset[k * 3], dataset[k * 3 + 1], dataset[k * 3 + 2])
                  ? -__builtin_inff() : 1e10f;
  int old = 0;
  if (tid == 0) idxs[0] = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = dataset[old * 3 + 0], y1 = dataset[old * 3 + 1], z1 = dataset[old * 3 + 2];
    long long best = -1LL;
    for (int k = tid; k < n; k += 1024) {
      const float dx = dataset[k * 3 + 0] - x1, dy = dataset[k * 3 + 1] - y1,
                  dz = dataset[k * 3 + 2] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      const long long key =
          ((long long)__float_as_int(d2) << 32) | (long long)(~fps_prio(k, L, Q));
      best = key > best ? key : best;
    }
    best = wave_max_i64(best);
    const int buf = j & 1;
    if ((tid & 63) == 0) s_slot[buf][tid >> 6] = best;
    __syncthreads();
    long long b2 = s_slot[buf][0];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
      long long o = s_slot[buf][w];
      b2 = o > b2 ? o : b2;
    }
    old = (b2 == -1LL) ? 0 : fps_prio_to_k(~(unsigned)(b2 & 0xffffffffLL), L, Q);
    old = __builtin_amdgcn_readfirstlane(old);
    if (tid == 0) idxs[j] = old;
  }
}

template <int THREADS, int PPT>
int launch_fps_reg(int b, int n, int m, int L, int Q, const float* dataset, int* idxs,
                   hipStream_t st) {
  size_t lds = (size_t)n * 3 * sizeof(float);
  int use_lds = lds + 1024 <= 160 * 1024;
  if (!use_lds) lds = 0;
  auto kern = fps_reg_kernel<THREADS, PPT>;
  if (lds > 48 * 1024)
    PVN3D_RETURN_IF_ERR(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                            hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds));
  hipLaunchKernelGGL(kern, dim3(b), dim3(THREADS), lds, st, n, m, L, Q, use_lds, dataset, idxs);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

__global__ void gather_points_kernel(int c, int n, int m, const float* __restrict__ points,
                                     const int* __restrict__ idx, float* __restrict__ out) {
  // grid: (ceil(m/256), c, b)
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y, i = blockIdx.z;
  if (j < m) {
    const int a = idx[(size_t)i * m + j];
    out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
  }
}

__global__ void gather_points_grad_kernel(int c, int n, int m,
                                          const float* __restrict__ grad_out,
                                          const int* __restrict__ idx,
                                          float* __restrict__ grad_points) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const int l = blockIdx.y, i = blockIdx.z;
  if (j < m) {
    const int a = idx[(size_t)i * m + j];
    atomicAdd(grad_points + ((size_t)i * c + l) * n + a, grad_out[((size_t)i * c + l) * m + j]);
  }
}

}  // namespace

extern "C" int pvn3d_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

extern "C" int pvn3d_abi_version(void) { return PVN3D_ABI_VERSION; }

extern "C" int pvn3d_furthest_point_sampling(int b, int n, int m, const float* dataset,
                                             float* temp, int* idxs, void* stream) {
  if (b <= 0 || m <= 0) return 0;
  if (n <= 0 || !dataset || !idxs) return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  const int bs = pvn3d_opt_n_threads(n);
  int L = 0;
  while ((1 << L) < bs) ++L;
  const int Q = (n + bs - 1) / bs;
  if (n <= 64) return launch_fps_reg<64, 1>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 128) return launch_fps_reg<64, 2>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 256) return launch_fps_reg<64, 4>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 512) return launch_fps_reg<64, 8>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 1024) return launch_fps_reg<256, 4>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 2048) return launch_fps_reg<256, 8>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 4096) return launch_fps_reg<1024, 4>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 8192) return launch_fps_reg<1024, 8>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 12288) return launch_fps_reg<1024, 12>(b, n, m, L, Q, dataset, idxs, st);
  if (n <= 16384) return launch_fps_reg<1024, 16>(b, n, m, L, Q, dataset, idxs, st);
  if (!temp) return (int)hipErrorInvalidValue;  // large clouds need the caller's scratch
  hipLaunchKernelGGL(fps_global_kernel, dim3(b), dim3(1024), 0, st, n, m, L, Q, dataset, temp,
                     idxs);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_gather_points(int b, int c, int n, int npoints, const float* points,
                                   const int* idx, float* out, void* stream) {
  if (b <= 0 || c <= 0 || npoints <= 0) return 0;
  hipLaunchKernelGGL(gather_points_kernel, dim3(pvn3d_ceil_div(npoints, 256), c, b), dim3(256),
                     0, (hipStream_t)stream, c, n, npoints, points, idx, out);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_gather_points_grad(int b, int c, int n, int npoints,
                                        const float* grad_out, const int* idx,
                                        float* grad_points, void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, st));
  if (npoints <= 0) return 0;
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3(pvn3d_ceil_div(npoints, 256), c, b),
                     dim3(256), 0, st, c, n, npoints, grad_out, idx, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
