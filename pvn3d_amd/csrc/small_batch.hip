// small_batch.hip -- layer-by-layer SharedMLP for launches too small for the fused chains of sa_mlp.hip.
//
// The fused inference kernels give one workgroup 64 columns and the WHOLE layer chain.  At one frame per call (the
// reference evaluates at test_mini_batch_size = 1, pvn3d/common.py:41) the deep levels have 512 - 4096 columns:
// 8 - 64 workgroups on 256 CUs, each grinding through K = 1536 x M = 512 alone (FP level 3: 284 us at 3 % of the
// chip).  Here the same arithmetic -- fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32, eval BatchNorm folded
// into W', b' -- runs one layer per launch with one wave per 32 x 32 output tile (64 x 64 per workgroup), activations
// as point-major fp32 matrices [columns][channels] in L2-resident global memory between the launches.
// Same layers as pointnet2_modules.py:58-71, 188-206 (SharedMLP, pytorch_utils.py:25-50).
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int SB_KC = 32;          // K chunk
constexpr int SB_LS = SB_KC + 1;   // LDS row stride (floats): the 32 rows of a fragment read hit 32 banks

// C[m][n] = act(sum_k A[m][k] W[n][k] + bias[n]); A [M][lda], W [N][ldw], C [M][ldc] fp32.  Workgroup = 64 x 64
// output tile, four waves of one 32 x 32 MFMA tile each; K in 32-chunks through double-buffered LDS (a single
// wave staging its own operands spent 8x the MFMA time on loads and LDS writes).
// Split-K (gridDim.z > 1): slice z owns the K range [z * klen, (z + 1) * klen) and writes its raw partial sums to
// part[z][M][N]; sb_splitk_reduce_kernel adds the slices in ascending z (deterministic), then bias and ReLU.  One frame
// gives the deep levels 64 - 256 output tiles of K = 512 - 1536: without the split a tile is 16 - 48 dependent
// load -> LDS -> MFMA round trips on one CU while the other CUs idle.
__global__ __launch_bounds__(256) void sb_linear_kernel(int M, int N, int K, const float* __restrict__ A, int lda,
                                                        const float* __restrict__ W, int ldw,
                                                        const float* __restrict__ bias, int relu,
                                                        float* __restrict__ C, int ldc, int klen,
                                                        float* __restrict__ part) {
  __shared__ float sA[2][64 * SB_LS];
  __shared__ float sW[2][64 * SB_LS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int kb = blockIdx.z * klen;
  K = min(K, kb + klen);                                 // this slice's end
  const int nchunks = (K - kb + SB_KC - 1) / SB_KC;
  const bool vec = ((lda | ldw) & 3) == 0;
  float4 ra[2], rw[2];
  auto gload = [&](int c) {
    const int k0 = kb + c * SB_KC;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int t = tid + p * 256;
      const int row = t >> 3, k = k0 + (t & 7) * 4;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vw = va;
      if (m0 + row < M) {
        const float* pa = A + (size_t)(m0 + row) * lda + k;
        if (vec && k + 3 < K) va = *reinterpret_cast<const float4*>(pa);
        else { if (k < K) va.x = pa[0]; if (k + 1 < K) va.y = pa[1]; if (k + 2 < K) va.z = pa[2]; if (k + 3 < K) va.w = pa[3]; }
      }
      if (n0 + row < N) {
        const float* pw = W + (size_t)(n0 + row) * ldw + k;
        if (vec && k + 3 < K) vw = *reinterpret_cast<const float4*>(pw);
        else { if (k < K) vw.x = pw[0]; if (k + 1 < K) vw.y = pw[1]; if (k + 2 < K) vw.z = pw[2]; if (k + 3 < K) vw.w = pw[3]; }
      }
      ra[p] = va;
      rw[p] = vw;
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int t = tid + p * 256;
      const int row = t >> 3, kq = (t & 7) * 4;
      float* da = &sA[buf][row * SB_LS + kq];
      float* dw = &sW[buf][row * SB_LS + kq];
      da[0] = ra[p].x; da[1] = ra[p].y; da[2] = ra[p].z; da[3] = ra[p].w;
      dw[0] = rw[p].x; dw[1] = rw[p].y; dw[2] = rw[p].z; dw[3] = rw[p].w;
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();
  const int mr = (wave & 1) * 32, nr = (wave >> 1) * 32;
  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < nchunks) gload(c + 1);
    // v_mfma_f32_32x32x2_f32: A operand = row (lane & 31), k = lane >> 5; B operand = column (lane & 31), same k
    const float* pa = &sA[buf][(mr + (lane & 31)) * SB_LS + (lane >> 5)];
    const float* pw = &sW[buf][(nr + (lane & 31)) * SB_LS + (lane >> 5)];
#pragma unroll
    for (int ks = 0; ks < SB_KC; ks += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[ks], pw[ks], acc, 0, 0, 0);
    if (c + 1 < nchunks) lstore(buf ^ 1);
    __syncthreads();
  }
  // C/D layout: column (n) = lane & 31, row (m) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int n = n0 + nr + (lane & 31);
  if (part) {
    if (n < N) {
      float* P = part + (size_t)blockIdx.z * M * N;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + mr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < M) P[(size_t)m * N + n] = acc[r];
      }
    }
    return;
  }
  if (n < N) {
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + mr + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m < M) {
        float v = acc[r] + b;
        if (relu) v = fmaxf(v, 0.f);
        C[(size_t)m * ldc + n] = v;
      }
    }
  }
}

__global__ void sb_splitk_reduce_kernel(int M, int N, int S, const float* __restrict__ part,
                                        const float* __restrict__ bias, int relu, float* __restrict__ C, int ldc) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= M * N) return;
  const int m = t / N, n = t - m * N;
  float v = part[t];
  for (int z = 1; z < S; ++z) v += part[(size_t)z * M * N + t];
  if (bias) v += bias[n];
  if (relu) v = fmaxf(v, 0.f);
  C[(size_t)m * ldc + n] = v;
}

// SA input rows (fp32): X0[(b*m + j)*ns + s][c] = relative xyz (c < 3 when use_xyz) ++ feat[b, c, idx] (strided source)
__global__ void sb_gather_sa_kernel(int b, int n, int m, int ns, int C, int use_xyz, const float* __restrict__ xyz,
                                    const float* __restrict__ new_xyz, const float* __restrict__ feat, long long fsb,
                                    long long fsc, long long fsn, const int* __restrict__ idx, float* __restrict__ X0,
                                    int ld) {
  const int cpr = (ld + 3) >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long rows = (long long)b * m * ns;
  if (t >= rows * cpr) return;
  const long long row = t / cpr;
  const int c0 = (int)(t % cpr) * 4;
  const int bi = (int)(row / ((long long)m * ns));
  const int j = (int)((row / ns) % m);
  const int k = idx[row];
  const int nx = use_xyz ? 3 : 0;
  float v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + i;
    float x = 0.f;
    if (c < nx) x = xyz[((size_t)bi * n + k) * 3 + c] - new_xyz[((size_t)bi * m + j) * 3 + c];
    else if (c < nx + C) x = feat[bi * fsb + (c - nx) * fsc + k * fsn];
    v[i] = x;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (c0 + i < ld) X0[row * ld + c0 + i] = v[i];
}

// FP input rows (fp32): three_interpolate(known)[c < C2] ++ unknown  (same expression order as
// three_interpolate_kernel, interpolate_gpu.cu:72-101: p1*w1 + p2*w2 + p3*w3)
__global__ void sb_gather_fp_kernel(int b, int n, int mk, int C2, int C1, const float* __restrict__ known,
                                    long long ksb, long long ksc, long long ksn, const float* __restrict__ unknown,
                                    long long usb, long long usc, long long usn, const int* __restrict__ idx,
                                    const float* __restrict__ w, float* __restrict__ X0, int ld) {
  const int cpr = (ld + 3) >> 2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long rows = (long long)b * n;
  if (t >= rows * cpr) return;
  const long long row = t / cpr;
  const int c0 = (int)(t % cpr) * 4;
  const int bi = (int)(row / n), i0 = (int)(row % n);
  const int k0 = idx[row * 3], k1 = idx[row * 3 + 1], k2 = idx[row * 3 + 2];
  const float w0 = w[row * 3], w1 = w[row * 3 + 1], w2 = w[row * 3 + 2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + i;
    if (c >= ld) break;
    float x = 0.f;
    if (c < C2) {
      const float* p = known + bi * ksb + c * ksc;
      x = p[k0 * ksn] * w0 + p[k1 * ksn] * w1 + p[k2 * ksn] * w2;
    } else if (c < C2 + C1) {
      x = unknown[bi * usb + (c - C2) * usc + i0 * usn];
    }
    X0[row * ld + c] = x;
  }
}

// max over the ns rows of every group: H [G*ns][ld] -> out[g*out_ld + c], c < C
__global__ void sb_pool_kernel(long long G, int ns, int ld, int C, const float* __restrict__ H, float* __restrict__ out,
                               long long out_ld) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= G * C) return;
  const long long g = t / C;
  const int c = (int)(t % C);
  float best = -__builtin_inff();
  for (int s = 0; s < ns; ++s) best = fmaxf(best, H[(g * ns + s) * ld + c]);
  out[g * out_ld + c] = best;
}

inline unsigned sb_grid1(long long work, int block) { return (unsigned)((work + block - 1) / block); }

}  // namespace

#define SB_ST ((hipStream_t)stream)

extern "C" int pvn3d_sb_linear_splits(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  const long long tiles = (long long)pvn3d_ceil_div(M, 64) * pvn3d_ceil_div(N, 64);
  int s = 1;
  while (s < 8 && tiles * s < 512 && K / (s * 2) >= 128) s *= 2;      // >= 128 of K per slice, <= 8 slices
  return s;
}

extern "C" int pvn3d_sb_linear(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias,
                               int relu, float* C, int ldc, float* part, int splits, void* stream) {
  if (M <= 0 || N <= 0) return 0;
  if (K <= 0 || !A || !W || !C || splits < 1 || (splits > 1 && !part)) return (int)hipErrorInvalidValue;
  if (splits == 1) {
    hipLaunchKernelGGL(sb_linear_kernel, dim3(pvn3d_ceil_div(M, 64), pvn3d_ceil_div(N, 64), 1), dim3(256), 0, SB_ST, M, N,
                       K, A, lda, W, ldw, bias, relu, C, ldc, K, (float*)nullptr);
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
  const int klen = pvn3d_ceil_div(pvn3d_ceil_div(K, splits), SB_KC) * SB_KC;
  const int gz = pvn3d_ceil_div(K, klen);
  hipLaunchKernelGGL(sb_linear_kernel, dim3(pvn3d_ceil_div(M, 64), pvn3d_ceil_div(N, 64), gz), dim3(256), 0, SB_ST, M, N,
                     K, A, lda, W, ldw, bias, relu, C, ldc, klen, part);
  PVN3D_LAUNCH_CHECK();
  hipLaunchKernelGGL(sb_splitk_reduce_kernel, dim3(sb_grid1((long long)M * N, 256)), dim3(256), 0, SB_ST, M, N, gz, part,
                     bias, relu, C, ldc);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_sb_gather_sa(int b, int n, int m, int ns, int C, int use_xyz, const float* xyz,
                                  const float* new_xyz, const float* feat, long long fsb, long long fsc, long long fsn,
                                  const int* idx, float* X0, int ld, void* stream) {
  const long long rows = (long long)b * m * ns;
  if (rows <= 0) return 0;
  if (ld < (use_xyz ? 3 : 0) + C) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sb_gather_sa_kernel, dim3(sb_grid1(rows * ((ld + 3) >> 2), 256)), dim3(256), 0, SB_ST, b, n, m, ns,
                     C, use_xyz, xyz, new_xyz, feat, fsb, fsc, fsn, idx, X0, ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_sb_gather_fp(int b, int n, int mk, int C2, int C1, const float* known, long long ksb, long long ksc,
                                  long long ksn, const float* unknown, long long usb, long long usc, long long usn,
                                  const int* idx, const float* w, float* X0, int ld, void* stream) {
  const long long rows = (long long)b * n;
  if (rows <= 0) return 0;
  if (ld < C2 + C1) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(sb_gather_fp_kernel, dim3(sb_grid1(rows * ((ld + 3) >> 2), 256)), dim3(256), 0, SB_ST, b, n, mk,
                     C2, C1, known, ksb, ksc, ksn, unknown, usb, usc, usn, idx, w, X0, ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_sb_pool_max(long long G, int ns, int ld, int C, const float* H, float* out, long long out_ld,
                                 void* stream) {
  if (G <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(sb_pool_kernel, dim3(sb_grid1(G * C, 256)), dim3(256), 0, SB_ST, G, ns, ld, C, H, out, out_ld);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
