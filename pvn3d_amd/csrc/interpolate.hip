// interpolate.hip -- three_nn / three_interpolate (feature propagation) for gfx950.
//
// Replaces pvn3d/_ext-src/src/interpolate_gpu.cu (reference): three_nn_kernel (:9-59),
// three_interpolate_kernel (:72-101), three_interpolate_grad_kernel (:116-143).
//
// three_nn: one lane per unknown point, the known cloud is staged through LDS in 1024-point
// chunks and every lane reads the same LDS address (broadcast, conflict-free); insertion with
// strict '<' so the earlier k wins ties exactly as the reference.  The reference keeps its
// running bests in double (init 1e40); fp32 with +inf init is equivalent because every
// candidate is an fp32 value (a d of +inf or NaN is never inserted in either form, and the
// unfilled slots convert to +inf in both).
// three_interpolate: same shape as group_points -- a thread owns 4 consecutive unknown
// points, keeps their 12 indices/weights in registers and loops over channels with one
// 16-byte store per channel.
// Arithmetic: -ffp-contract=off; d = ((dx*dx + dy*dy) + dz*dz), out = ((p1*w1 + p2*w2) + p3*w3).

#include "common.h"

namespace {

constexpr int NN_CHUNK = 1024;
constexpr int TI_LDS_MAX_M = 4096;  // known-point rows up to 16 KiB are staged through LDS (row-owner kernel)

// grid: (ceil(n/256), b)
__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m,
                                                       const float* __restrict__ unknown,
                                                       const float* __restrict__ known,
                                                       float* __restrict__ dist2,
                                                       int* __restrict__ idx) {
  __shared__ float4 s_k[NN_CHUNK];
  const int bi = blockIdx.y;
  const int j = blockIdx.x * 256 + threadIdx.x;
  unknown += (size_t)bi * n * 3;
  known += (size_t)bi * m * 3;
  float ux = 0.f, uy = 0.f, uz = 0.f;
  if (j < n) {
    ux = unknown[j * 3 + 0];
    uy = unknown[j * 3 + 1];
    uz = unknown[j * 3 + 2];
  }
  float b1 = __builtin_inff(), b2 = __builtin_inff(), b3 = __builtin_inff();
  int i1 = 0, i2 = 0, i3 = 0;
  for (int k0 = 0; k0 < m; k0 += NN_CHUNK) {
    const int cnt = min(NN_CHUNK, m - k0);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += 256) {
      const float* q = known + (size_t)(k0 + t) * 3;
      s_k[t] = make_float4(q[0], q[1], q[2], 0.f);
    }
    __syncthreads();
    for (int kk = 0; kk < cnt; ++kk) {
      const float4 q = s_k[kk];   // one broadcast ds_read per candidate
      const float dx = ux - q.x, dy = uy - q.y, dz = uz - q.z;
      const float d = dx * dx + dy * dy + dz * dz;
      const bool lt3 = d < b3;
      // After the first few candidates most of them beat nobody's third best: skip the
      // insertion network unless some lane of the wave needs it (wave-uniform branch;
      // measured 853 -> 592 us at n=12288, m=2048, 64 clouds; grouping 4 candidates per test
      // was slower, 725 us).
      if (!__any(lt3)) continue;
      const int k = k0 + kk;
      // branch-free form of the if / else-if / else-if chain (interpolate_gpu.cu:38-56)
      const bool lt1 = d < b1, lt2 = d < b2;
      const float nb3 = lt2 ? b2 : (lt3 ? d : b3);
      const int ni3 = lt2 ? i2 : (lt3 ? k : i3);
      const float nb2 = lt1 ? b1 : (lt2 ? d : b2);
      const int ni2 = lt1 ? i1 : (lt2 ? k : i2);
      b1 = lt1 ? d : b1;
      i1 = lt1 ? k : i1;
      b2 = nb2; i2 = ni2;
      b3 = nb3; i3 = ni3;
    }
  }
  if (j < n) {
    float* od = dist2 + ((size_t)bi * n + j) * 3;
    int* oi = idx + ((size_t)bi * n + j) * 3;
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
  }
}

// grid: (ceil(n/1024), n_chunks, b), n % 4 == 0
__global__ __launch_bounds__(256) void three_interpolate_vec4_kernel(
    int c, int m, int n, int cch, const float* __restrict__ points,
    const int* __restrict__ idx, const float* __restrict__ weight, float* __restrict__ out) {
  const int j0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (j0 >= n) return;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * cch;
  const int c1 = min(c0 + cch, c);
  int id[12];
  float w[12];
  {
    const int4* ip = reinterpret_cast<const int4*>(idx + ((size_t)bi * n + j0) * 3);
    const float4* wp = reinterpret_cast<const float4*>(weight + ((size_t)bi * n + j0) * 3);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int4 a = ip[u];
      const float4 f = wp[u];
      id[u * 4 + 0] = a.x; id[u * 4 + 1] = a.y; id[u * 4 + 2] = a.z; id[u * 4 + 3] = a.w;
      w[u * 4 + 0] = f.x; w[u * 4 + 1] = f.y; w[u * 4 + 2] = f.z; w[u * 4 + 3] = f.w;
    }
  }
  const float* row = points + ((size_t)bi * c + c0) * m;
  float* o = out + ((size_t)bi * c + c0) * n + j0;
  for (int l = c0; l < c1; ++l) {
    float r[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      r[u] = row[id[u * 3 + 0]] * w[u * 3 + 0] + row[id[u * 3 + 1]] * w[u * 3 + 1] +
             row[id[u * 3 + 2]] * w[u * 3 + 2];
    *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1], r[2], r[3]);
    row += m;
    o += n;
  }
}

// Row-owner variant: a workgroup stages CPB channel rows of `points` (m floats each) in LDS
// once and streams over a span of unknown points; idx/weight are read once per CPB channels and
// each channel is one long contiguous write stream.
// grid: (n_jchunks, ceil(c/CPB), b); dynamic LDS = CPB*m floats; n % 4 == 0, m % 4 == 0.
template <int CPB>
__global__ __launch_bounds__(256) void three_interpolate_rows_kernel(
    int c, int m, int n, int jchunk, const float* __restrict__ points,
    const int* __restrict__ idx, const float* __restrict__ weight, float* __restrict__ out) {
  extern __shared__ float s_row[];  // [CPB][m]
  const int tid = threadIdx.x;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * CPB;
  const int nc = min(CPB, c - c0);
  const int m4 = m >> 2;
  const float4* row = reinterpret_cast<const float4*>(points + ((size_t)bi * c + c0) * m);
  float4* s4 = reinterpret_cast<float4*>(s_row);
  for (int q = tid; q < nc * m4; q += 256) s4[q] = row[q];
  __syncthreads();
  const int j_begin = blockIdx.x * jchunk;
  const int j_end = min(j_begin + jchunk, n);
  float* o = out + ((size_t)bi * c + c0) * n;
  typedef float v4f __attribute__((ext_vector_type(4)));
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int* const ibase = idx + (size_t)bi * n * 3;
  const float* const wbase = weight + (size_t)bi * n * 3;
  int j0 = j_begin + tid * 4;
  if (nc == CPB && j_end - j_begin > 2048) {      // (a span of one or two trips has nothing to overlap)
    // Full row group: the next four points' idx / weight vectors (6 x 16 B) are requested BEFORE this trip's
    // stores and awaited after them with s_waitcnt vmcnt(CPB) -- loads and stores share one in-order counter
    // on gfx950 (see group_points.hip); left to the compiler every trip waited for the previous trip's write
    // acknowledgements.
    v4i ic[3], in_[3];
    v4f wc[3], wn[3];
    {
      const int jc = min(j0, n - 4);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        ic[u] = reinterpret_cast<const v4i*>(ibase + (size_t)jc * 3)[u];
        wc[u] = reinterpret_cast<const v4f*>(wbase + (size_t)jc * 3)[u];
      }
      // (the compiler's own wait for these six loads stays outside the loop)
      asm volatile("" : "+v"(ic[0]), "+v"(ic[1]), "+v"(ic[2]), "+v"(wc[0]), "+v"(wc[1]), "+v"(wc[2]));
    }
    while (j0 < j_end) {
      const int jn = min(j0 + 1024, n - 4);
      const v4i* ipn = reinterpret_cast<const v4i*>(ibase + (size_t)jn * 3);
      const v4f* wpn = reinterpret_cast<const v4f*>(wbase + (size_t)jn * 3);
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(in_[u]) : "v"(ipn + u) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(wn[u]) : "v"(wpn + u) : "memory");
      }
      int id[12];
      float w[12];
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        id[u * 4 + 0] = ic[u].x; id[u * 4 + 1] = ic[u].y; id[u * 4 + 2] = ic[u].z; id[u * 4 + 3] = ic[u].w;
        w[u * 4 + 0] = wc[u].x; w[u * 4 + 1] = wc[u].y; w[u * 4 + 2] = wc[u].z; w[u * 4 + 3] = wc[u].w;
      }
#pragma unroll
      for (int ch = 0; ch < CPB; ++ch) {
        const float* sr = s_row + ch * m;
        float r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
          r[u] = sr[id[u * 3 + 0]] * w[u * 3 + 0] + sr[id[u * 3 + 1]] * w[u * 3 + 1] +
                 sr[id[u * 3 + 2]] * w[u * 3 + 2];
        // streaming output (written once, not re-read here): non-temporal 16-byte store
        v4f val = {r[0], r[1], r[2], r[3]};
        __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + (size_t)ch * n + j0));
      }
      asm volatile("s_waitcnt vmcnt(%6)"
                   : "+v"(in_[0]), "+v"(in_[1]), "+v"(in_[2]), "+v"(wn[0]), "+v"(wn[1]), "+v"(wn[2])
                   : "n"(CPB)
                   : "memory");
#pragma unroll
      for (int u = 0; u < 3; ++u) { ic[u] = in_[u]; wc[u] = wn[u]; }
      j0 += 1024;
    }
    return;
  }
  for (; j0 < j_end; j0 += 1024) {
    int id[12];
    float w[12];
    const int4* ip = reinterpret_cast<const int4*>(ibase + (size_t)j0 * 3);
    const float4* wp = reinterpret_cast<const float4*>(wbase + (size_t)j0 * 3);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int4 a = ip[u];
      const float4 f = wp[u];
      id[u * 4 + 0] = a.x; id[u * 4 + 1] = a.y; id[u * 4 + 2] = a.z; id[u * 4 + 3] = a.w;
      w[u * 4 + 0] = f.x; w[u * 4 + 1] = f.y; w[u * 4 + 2] = f.z; w[u * 4 + 3] = f.w;
    }
    for (int ch = 0; ch < nc; ++ch) {
      const float* sr = s_row + ch * m;
      float r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        r[u] = sr[id[u * 3 + 0]] * w[u * 3 + 0] + sr[id[u * 3 + 1]] * w[u * 3 + 1] +
               sr[id[u * 3 + 2]] * w[u * 3 + 2];
      v4f val = {r[0], r[1], r[2], r[3]};
      __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + (size_t)ch * n + j0));
    }
  }
}

// any-n fallback / reference-bug-compat path.  out[(bi*c+l)*n_out + j] for j < n_out, with
// idx/weight batch stride given explicitly.  grid: (ceil(n_out/256), c, b)
__global__ __launch_bounds__(256) void three_interpolate_scalar_kernel(
    int c, int m, int n_out, size_t iw_batch_stride, const float* __restrict__ points,
    const int* __restrict__ idx, const float* __restrict__ weight, float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_out) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  const int* id = idx + (size_t)bi * iw_batch_stride + (size_t)j * 3;
  const float* w = weight + (size_t)bi * iw_batch_stride + (size_t)j * 3;
  const float* row = points + ((size_t)bi * c + l) * m;
  out[((size_t)bi * c + l) * n_out + j] = row[id[0]] * w[0] + row[id[1]] * w[1] + row[id[2]] * w[2];
}

// grid: (ceil(n/256), c, b)
__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float* __restrict__ grad_out, const int* __restrict__ idx,
    const float* __restrict__ weight, float* __restrict__ grad_points) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  const int* id = idx + ((size_t)bi * n + j) * 3;
  const float* w = weight + ((size_t)bi * n + j) * 3;
  const float g = grad_out[((size_t)bi * c + l) * n + j];
  float* gp = grad_points + ((size_t)bi * c + l) * m;
  atomicAdd(gp + id[0], g * w[0]);
  atomicAdd(gp + id[1], g * w[1]);
  atomicAdd(gp + id[2], g * w[2]);
}

// Row-owner scatter (the forward row kernel run backwards): a workgroup owns CPB channel rows of one
// cloud's grad_points as LDS accumulators, streams a span of unknown points -- idx / weight / grad_out
// read with coalesced loads -- and adds the three weighted contributions with LDS float atomics; the
// rows leave with coalesced stores (the span covers all points) or one global atomic per touched
// element (spans split over several workgroups).  The element-per-thread kernel above issues
// 3 * c * n global atomics per cloud: 8.6 ms of a 65 ms training step at c = 256, n = 12288, 24 frames.
// grid: (n_jchunks, ceil(c/CPB), b); dynamic LDS = CPB*m floats.
template <int CPB>
__global__ __launch_bounds__(256) void three_interpolate_grad_rows_kernel(
    int c, int n, int m, int jchunk, const float* __restrict__ grad_out, const int* __restrict__ idx,
    const float* __restrict__ weight, float* __restrict__ grad_points) {
  extern __shared__ float s_acc[];  // [CPB][m]
  const int tid = threadIdx.x;
  const int bi = blockIdx.z, c0 = blockIdx.y * CPB;
  const int nc = min(CPB, c - c0);
  for (int q = tid; q < nc * m; q += 256) s_acc[q] = 0.f;
  __syncthreads();
  const int j_begin = blockIdx.x * jchunk, j_end = min(j_begin + jchunk, n);
  const int* ip = idx + (size_t)bi * n * 3;
  const float* wp = weight + (size_t)bi * n * 3;
  const float* g = grad_out + ((size_t)bi * c + c0) * n;
  for (int j = j_begin + tid; j < j_end; j += 256) {
    const int i0 = ip[j * 3], i1 = ip[j * 3 + 1], i2 = ip[j * 3 + 2];
    const float w0 = wp[j * 3], w1 = wp[j * 3 + 1], w2 = wp[j * 3 + 2];
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      if (u < nc) {
        const float gv = g[(size_t)u * n + j];
        atomicAdd(&s_acc[u * m + i0], gv * w0);
        atomicAdd(&s_acc[u * m + i1], gv * w1);
        atomicAdd(&s_acc[u * m + i2], gv * w2);
      }
    }
  }
  __syncthreads();
  float* o = grad_points + ((size_t)bi * c + c0) * m;
  if (gridDim.x == 1) {
    for (int q = tid; q < nc * m; q += 256) o[q] = s_acc[q];
  } else {
    for (int q = tid; q < nc * m; q += 256) {
      const float v = s_acc[q];
      if (v != 0.f) atomicAdd(o + q, v);
    }
  }
}

}  // namespace

extern "C" int pvn3d_three_nn(int b, int n, int m, const float* unknown, const float* known,
                              float* dist2, int* idx, void* stream) {
  if (b <= 0 || n <= 0) return 0;
  if (m < 0 || !unknown || !known || !dist2 || !idx) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(three_nn_kernel, dim3(pvn3d_ceil_div(n, 256), b), dim3(256), 0,
                     (hipStream_t)stream, n, m, unknown, known, dist2, idx);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_three_interpolate(int b, int c, int m, int n, const float* points,
                                       const int* idx, const float* weight, float* out,
                                       void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const bool aligned = (n % 4 == 0) && (((uintptr_t)out & 15) == 0) &&
                       (((uintptr_t)idx & 15) == 0) && (((uintptr_t)weight & 15) == 0);
  const bool rows_ok = aligned && (m % 4 == 0) && m <= TI_LDS_MAX_M && m > 0 && (((uintptr_t)points & 15) == 0);
  if (rows_ok) {
    // up to 8 rows per workgroup in at most 64 KiB of LDS (idx/weight are 24 B per point, read once per row
    // group); tile-owner and direct-gather variants measured slower and were removed
    int cpb = 8;
    while (cpb > 1 && (size_t)cpb * m * 4 > 64 * 1024) cpb >>= 1;
    while (cpb > 1 && cpb > c) cpb >>= 1;
    const int rows = pvn3d_ceil_div(c, cpb);
    int jch = pvn3d_ceil_div(4096, rows * b);
    if (jch < 1) jch = 1;
    int jchunk = pvn3d_ceil_div(pvn3d_ceil_div(n, jch), 1024) * 1024;
    if (jchunk < 2048) jchunk = 2048;
    jch = pvn3d_ceil_div(n, jchunk);
    const size_t lds = (size_t)cpb * m * sizeof(float);
    switch (cpb) {
      case 8:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_rows_kernel<8>));
        hipLaunchKernelGGL(three_interpolate_rows_kernel<8>, dim3(jch, rows, b), dim3(256), lds, st, c, m, n, jchunk,
                           points, idx, weight, out);
        break;
      case 4:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_rows_kernel<4>));
        hipLaunchKernelGGL(three_interpolate_rows_kernel<4>, dim3(jch, rows, b), dim3(256), lds, st, c, m, n, jchunk,
                           points, idx, weight, out);
        break;
      case 2:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_rows_kernel<2>));
        hipLaunchKernelGGL(three_interpolate_rows_kernel<2>, dim3(jch, rows, b), dim3(256), lds, st, c, m, n, jchunk,
                           points, idx, weight, out);
        break;
      default:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_rows_kernel<1>));
        hipLaunchKernelGGL(three_interpolate_rows_kernel<1>, dim3(jch, rows, b), dim3(256), lds, st, c, m, n, jchunk,
                           points, idx, weight, out);
        break;
    }
  } else if (aligned) {
    const int gx = pvn3d_ceil_div(n, 1024);
    int chunks = pvn3d_ceil_div(2048, gx * b);
    if (chunks < 1) chunks = 1;
    if (chunks > c) chunks = c;
    const int cch = pvn3d_ceil_div(c, chunks);
    chunks = pvn3d_ceil_div(c, cch);
    hipLaunchKernelGGL(three_interpolate_vec4_kernel, dim3(gx, chunks, b), dim3(256), 0, st, c,
                       m, n, cch, points, idx, weight, out);
  } else {
    hipLaunchKernelGGL(three_interpolate_scalar_kernel, dim3(pvn3d_ceil_div(n, 256), c, b),
                       dim3(256), 0, st, c, m, n, (size_t)n * 3, points, idx, weight, out);
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}

extern "C" int pvn3d_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                            const int* idx, const float* weight,
                                            float* grad_points, int refbug_compat,
                                            void* stream) {
  if (b <= 0 || c <= 0 || m <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (refbug_compat) {
    // interpolate.cpp:89-93: forward kernel with (m_arg = n, n_arg = m); idx/weight batch
    // stride becomes n_arg*3 = m*3.  Reads stay in bounds only if m <= n (FP modules: m < n).
    if (n <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(three_interpolate_scalar_kernel, dim3(pvn3d_ceil_div(m, 256), c, b),
                       dim3(256), 0, st, c, n, m, (size_t)m * 3, grad_out, idx, weight,
                       grad_points);
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
  if ((size_t)m * 4 <= 64 * 1024 && n > 0) {     // rows of known points fit the LDS: row-owner scatter
    int cpb = 4;
    while (cpb > 1 && (size_t)cpb * m * 4 > 64 * 1024) cpb >>= 1;
    while (cpb > 1 && cpb > c) cpb >>= 1;
    const int rows = pvn3d_ceil_div(c, cpb);
    // split the unknown points only when the rows alone do not fill the chip
    int jch = pvn3d_ceil_div(2048, rows * b);
    if (jch < 1) jch = 1;
    int jchunk = pvn3d_ceil_div(pvn3d_ceil_div(n, jch), 256) * 256;
    if (jchunk < 1024) jchunk = 1024;
    jch = pvn3d_ceil_div(n, jchunk);
    if (jch > 1) PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * m, st));
    const size_t lds = (size_t)cpb * m * sizeof(float);
    switch (cpb) {
      case 4:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_grad_rows_kernel<4>));
        hipLaunchKernelGGL(three_interpolate_grad_rows_kernel<4>, dim3(jch, rows, b), dim3(256), lds, st, c, n, m,
                           jchunk, grad_out, idx, weight, grad_points);
        break;
      case 2:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_grad_rows_kernel<2>));
        hipLaunchKernelGGL(three_interpolate_grad_rows_kernel<2>, dim3(jch, rows, b), dim3(256), lds, st, c, n, m,
                           jchunk, grad_out, idx, weight, grad_points);
        break;
      default:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_interpolate_grad_rows_kernel<1>));
        hipLaunchKernelGGL(three_interpolate_grad_rows_kernel<1>, dim3(jch, rows, b), dim3(256), lds, st, c, n, m,
                           jchunk, grad_out, idx, weight, grad_points);
        break;
    }
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
  PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * m, st));
  if (n <= 0) return 0;
  hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3(pvn3d_ceil_div(n, 256), c, b),
                     dim3(256), 0, st, c, n, m, grad_out, idx, weight, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}

// Inverse-distance weights of PointnetFPModule.forward (pointnet2_modules.py:184-186) from three_nn's squared distances,
// in the reference's fp32 operation order: dist = sqrt(dist2); dist_recip = 1.0 / (dist + 1e-8);
// norm = sum(dist_recip, dim=2); weight = dist_recip / norm.  One launch instead of five elementwise torch kernels.
namespace {
__global__ __launch_bounds__(256) void three_nn_weights_kernel(long long rows, const float* __restrict__ dist2,
                                                               float* __restrict__ weight) {
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i >= rows) return;
  const float r0 = 1.0f / (sqrtf(dist2[i * 3 + 0]) + 1e-8f);
  const float r1 = 1.0f / (sqrtf(dist2[i * 3 + 1]) + 1e-8f);
  const float r2 = 1.0f / (sqrtf(dist2[i * 3 + 2]) + 1e-8f);
  const float norm = (r0 + r1) + r2;
  weight[i * 3 + 0] = r0 / norm;
  weight[i * 3 + 1] = r1 / norm;
  weight[i * 3 + 2] = r2 / norm;
}
}  // namespace

extern "C" int pvn3d_three_nn_weights(long long rows, const float* dist2, float* weight, void* stream) {
  if (rows <= 0) return 0;
  if (!dist2 || !weight || rows > 0x7fffffffLL * 256) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(three_nn_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, rows,
                     dist2, weight);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
