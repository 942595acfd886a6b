// mfma_loop_bench.hip -- isolates the inner pair loop of the fused-MLP kernels (csrc/sa_mlp.hip, span8):
// per pair of K steps a wave issues 4 LDS fragment reads, NT*4 v_mfma_f32_32x32x2_f32 and NT 8-byte weight
// loads into an 8-deep register ring.  Variants switch the operand streams off to see which one costs the
// matrix pipe its idle time.  Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_bench.hip -o tools/mfma_loop_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NT, bool LOADW, bool LDSB, bool FENCE>
__global__ __launch_bounds__(256, 2) void k(const float2* __restrict__ W, float* __restrict__ out, int pairs, int reps,
                                            int lds_floats) {
  extern __shared__ float H[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < lds_floats; i += 256) H[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[NT][2];
  for (int t = 0; t < NT; ++t) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
  const int half = lane >> 5, col = lane & 31;
  const float2* wp = W + (size_t)wave * 64 + lane;
  const size_t pstride = 8 * 64, tstride = 4 * 64;
  for (int rep = 0; rep < reps; ++rep) {
    float2 ring[8][NT];
    for (int u = 0; u < 8; ++u) for (int t = 0; t < NT; ++t) ring[u][t] = LOADW ? wp[(size_t)u * pstride + t * tstride] : float2{1.f + u, 2.f + t};
    const float* rows = H + half * 64;
    for (int p0 = 0; p0 + 8 <= pairs; p0 += 8) {
      float b[2][2][2];
      const float* r0 = rows + (size_t)p0 * 4 * 64;
      if (LDSB) { b[0][0][0] = r0[col]; b[0][1][0] = r0[128 + col]; b[0][0][1] = r0[col + 32]; b[0][1][1] = r0[128 + col + 32]; }
      else { b[0][0][0] = b[0][1][0] = b[0][0][1] = b[0][1][1] = 0.5f; }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (u + 1 < 8) {
          const float* r1 = r0 + (u + 1) * 4 * 64;
          if (LDSB) { b[(u + 1) & 1][0][0] = r1[col]; b[(u + 1) & 1][1][0] = r1[128 + col]; b[(u + 1) & 1][0][1] = r1[col + 32]; b[(u + 1) & 1][1][1] = r1[128 + col + 32]; }
          else { b[(u + 1) & 1][0][0] = b[(u + 1) & 1][1][0] = b[(u + 1) & 1][0][1] = b[(u + 1) & 1][1][1] = 0.25f; }
        }
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].x, b[u & 1][0][0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].x, b[u & 1][0][1], acc[t][1], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].y, b[u & 1][1][0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].y, b[u & 1][1][1], acc[t][1], 0, 0, 0);
        }
        if (LOADW) {
          const int pn = min(p0 + 8 + u, pairs - 1);
#pragma unroll
          for (int t = 0; t < NT; ++t) ring[u][t] = wp[(size_t)pn * pstride + t * tstride];
        }
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0;
  for (int t = 0; t < NT; ++t) for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) s += acc[t][c][r];
  out[blockIdx.x * 256 + tid] = s;
}

// ---- chain emulation: L "layers" of `pairs` K-pairs each; a barrier after every 8 pairs (layer-0 style chunk
// hand-over) when CHUNK_BARRIERS, and between layers: barrier, relu(acc) -> H (LDS), barrier.  CT column tiles
// of 32 columns per workgroup (CT = 2: the shipped 64-column block, CT = 1: a 32-column block).
template <int NT, int CT, bool CHUNK_BARRIERS>
__global__ __launch_bounds__(256) void chain_k(const float2* __restrict__ W, float* __restrict__ out, int pairs, int layers,
                                               int reps, int lds_floats) {
  extern __shared__ float H[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < lds_floats; i += 256) H[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[NT][CT];
  const int half = lane >> 5, col = lane & 31;
  const float2* wp = W + (size_t)wave * 64 + lane;
  const size_t pstride = 8 * 64, tstride = 4 * 64;
  constexpr int HC = 32 * CT;
  float s = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int l = 0; l < layers; ++l) {
      for (int t = 0; t < NT; ++t) for (int c = 0; c < CT; ++c) for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;
      float2 ring[8][NT];
      for (int u = 0; u < 8; ++u) for (int t = 0; t < NT; ++t) ring[u][t] = wp[(size_t)u * pstride + t * tstride];
      const float* rows = H + half * HC;
      for (int p0 = 0; p0 + 8 <= pairs; p0 += 8) {
        float b[2][2][CT];
        const float* r0 = rows + (size_t)p0 * 4 * HC;
#pragma unroll
        for (int c = 0; c < CT; ++c) { b[0][0][c] = r0[col + 32 * c]; b[0][1][c] = r0[2 * HC + col + 32 * c]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u + 1 < 8) {
            const float* r1 = r0 + (u + 1) * 4 * HC;
#pragma unroll
            for (int c = 0; c < CT; ++c) { b[(u + 1) & 1][0][c] = r1[col + 32 * c]; b[(u + 1) & 1][1][c] = r1[2 * HC + col + 32 * c]; }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].x, b[u & 1][0][c], acc[t][c], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][t].y, b[u & 1][1][c], acc[t][c], 0, 0, 0);
          const int pn = min(p0 + 8 + u, pairs - 1);
#pragma unroll
          for (int t = 0; t < NT; ++t) ring[u][t] = wp[(size_t)pn * pstride + t * tstride];
          __builtin_amdgcn_sched_barrier(0);
        }
        if (CHUNK_BARRIERS && l == 0) __syncthreads();
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (wave + 4 * t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            H[row * HC + 32 * c + col] = fmaxf(acc[t][c][r], 0.f) * 1e-6f + 1e-3f;
          }
      __syncthreads();
    }
  }
  for (int t = 0; t < NT; ++t) for (int c = 0; c < CT; ++c) for (int r = 0; r < 16; ++r) s += acc[t][c][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <int NT, int CT, bool CB>
void run_chain(const char* name, const float2* W, float* out, int wgs_per_cu) {
  const int pairs = 64, layers = 3, reps = 40, hrows = 256;
  const size_t lds_need = (size_t)hrows * 32 * CT * 4;
  size_t lds = 160 * 1024 / wgs_per_cu - 2048;
  if (lds < lds_need) { printf("%-44s wg/cu %d: does not fit\n", name, wgs_per_cu); return; }
  auto kern = chain_k<NT, CT, CB>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * wgs_per_cu * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256, lds>>>(W, out, pairs, layers, 2, hrows * 32 * CT);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, 256, lds>>>(W, out, pairs, layers, reps, hrows * 32 * CT);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)grid * 4 * reps * layers * pairs * NT * CT * 2;
  const double tf = mfma * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-44s wg/cu %d  %8.3f ms  %7.1f TFLOP/s  (%4.1f %% of 157.3)\n", name, wgs_per_cu, ms, tf, 100 * tf / 157.3);
}

template <int NT, bool LOADW, bool LDSB, bool FENCE>
void run(const char* name, const float2* W, float* out, int wgs_per_cu) {
  const int pairs = 64, reps = 200, hrows = pairs * 4;
  const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : 64 * 1024;          // forces 1 or 2 workgroups per CU
  auto kern = k<NT, LOADW, LDSB, FENCE>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int grid = 256 * wgs_per_cu * 4;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, 256, lds>>>(W, out, pairs, 4, hrows * 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  kern<<<grid, 256, lds>>>(W, out, pairs, reps, hrows * 64);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)grid * 4 * reps * pairs * NT * 4;       // per-wave MFMA instructions, all waves
  const double tf = mfma * 4096.0 / (ms * 1e-3) / 1e12;
  printf("%-34s wg/cu %d  %8.3f ms  %7.1f TFLOP/s  (%4.1f %% of 157.3)\n", name, wgs_per_cu, ms, tf, 100 * tf / 157.3);
}

int main() {
  float2* W; float* out;
  hipMalloc(&W, sizeof(float2) * 64 * 8 * 64 * 2);
  hipMemset(W, 0, sizeof(float2) * 64 * 8 * 64 * 2);
  hipMalloc(&out, 4 * 256 * 8192);
  for (int occ = 1; occ <= 2; ++occ) {
    run<2, false, false, true>("NT2 no-weights no-lds", W, out, occ);
    run<2, true, false, true>("NT2 weights", W, out, occ);
    run<2, false, true, true>("NT2 lds", W, out, occ);
    run<2, true, true, true>("NT2 weights+lds (as shipped)", W, out, occ);
    run<2, true, true, false>("NT2 weights+lds no fences", W, out, occ);
    run<1, true, true, true>("NT1 weights+lds (as shipped)", W, out, occ);
    run<1, false, false, true>("NT1 no-weights no-lds", W, out, occ);
  }
  printf("chains: 3 layers x 64 pairs, 4 waves, relu->H hand-over between layers (+ a barrier per 8 pairs in layer 0)\n");
  for (int occ = 1; occ <= 4; ++occ) {
    run_chain<2, 2, true>("64-col block, 2 row tiles/wave, chunk barriers", W, out, occ);
    run_chain<2, 2, false>("64-col block, 2 row tiles/wave", W, out, occ);
    run_chain<2, 1, true>("32-col block, 2 row tiles/wave, chunk barriers", W, out, occ);
    run_chain<2, 1, false>("32-col block, 2 row tiles/wave", W, out, occ);
  }
  return 0;
}
