// CUDA-on-CPU execution shim used ONLY to compile the reference's own kernel sources
// (/root/reference/pvn3d/_ext-src/src/*_gpu.cu, where they lie) with g++ into oracle/_ref/.
// TEST INFRASTRUCTURE: nothing under pvn3d_amd/ includes, links or loads this.
//
// Model: a launch runs the blocks of the grid one after another; the threads of a block are
// cooperative fibers (ucontext) on one OS thread, resumed in descending linear-thread-id order
// (why: cuda_cpu_shim.cpp); __syncthreads() yields to the scheduler, which resumes
// every other live fiber before coming back, i.e. a true block barrier.  `__shared__` becomes
// function-local `static` storage, which is per-block because blocks do not overlap in time.
// atomicAdd is a plain read-modify-write (threads never run concurrently).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <functional>

struct uint3_shim { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

extern uint3_shim threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

#define __global__
#define __device__
#define __host__
#define __shared__ static

typedef int cudaError_t;
typedef void *cudaStream_t;
enum { cudaSuccess = 0 };
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t) { return "no error (CPU shim)"; }

// CUDA's overloaded min/max in the global namespace (float versions have fminf/fmaxf semantics).
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }

inline float atomicAdd(float *p, float v) { float o = *p; *p = o + v; return o; }

void __syncthreads();

namespace at { namespace cuda { inline cudaStream_t getCurrentCUDAStream() { return nullptr; } } }

namespace shim {
struct LaunchCfg {
  dim3 grid, block;
  LaunchCfg(dim3 g, dim3 b, size_t = 0, cudaStream_t = nullptr) : grid(g), block(b) {}
};
void run_grid(const LaunchCfg &cfg, const std::function<void()> &thread_body);

template <typename... P, typename... A>
void launch(const LaunchCfg &cfg, void (*kernel)(P...), A... args) {
  run_grid(cfg, [=]() { kernel(args...); });
}
}  // namespace shim
