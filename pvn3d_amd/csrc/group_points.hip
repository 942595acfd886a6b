// group_points.hip -- neighbourhood gather (the HBM-bound half of set abstraction), gfx950.
//
// Replaces group_points_kernel / group_points_grad_kernel,
// pvn3d/_ext-src/src/group_points_gpu.cu:8-28, 43-64 (reference), and fuses the caller's
// `grouped_xyz -= new_xyz` and `torch.cat` (pvn3d/lib/pointnet2_utils/pointnet2_utils.py:311-321)
// into the same pass by writing straight into the concatenated (b, 3+c, npoint, nsample) tensor.
//
// Traffic: the output (4*C*npoint*nsample bytes) dominates; inputs (4*C*n) are re-read from
// L2.  Layout: out[b][l][j][s] -- for a fixed channel l the (j,s) plane is contiguous, so a
// thread owns 4 consecutive (j,s) positions, keeps their 4 neighbour indices in registers for
// its whole channel loop and issues one 16-byte store per channel (1 KiB per wave instruction).
// The reference's thread writes with stride nsample and loops k serially.
#include "common.h"

namespace {

// grid: (ceil(P/1024), n_chunks, b); P = npoints*nsample, P % 4 == 0
__global__ __launch_bounds__(256) void group_points_vec4_kernel(
    int c, int n, int P, int cch, const float* __restrict__ points,
    const int* __restrict__ idx, float* __restrict__ out, size_t out_batch_stride) {
  const int p0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p0 >= P) return;
  const int bi = blockIdx.z;
  const int c0 = blockIdx.y * cch;
  const int c1 = min(c0 + cch, c);
  const int4 id = *reinterpret_cast<const int4*>(idx + (size_t)bi * P + p0);
  const float* row = points + ((size_t)bi * c + c0) * n;
  float* o = out + (size_t)bi * out_batch_stride + (size_t)c0 * P + p0;
  int l = c0;
  for (; l + 4 <= c1; l += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* r = row + (size_t)u * n;
      v[u] = make_float4(r[id.x], r[id.y], r[id.z], r[id.w]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<float4*>(o + (size_t)u * P) = v[u];
    row += (size_t)4 * n;
    o += (size_t)4 * P;
  }
  for (; l < c1; ++l) {
    *reinterpret_cast<float4*>(o) = make_float4(row[id.x], row[id.y], row[id.z], row[id.w]);
    row += n;
    o += P;
  }
}

// Row-owner variant: a workgroup owns CPB whole channel rows of one cloud (staged once in LDS)
// and streams over a span of positions, reading idx with coalesced 16-byte loads and writing
// CPB contiguous output streams -- long sequential HBM write bursts per workgroup, and idx is
// re-read from L2 only once per CPB channels.
// grid: (n_pchunks, ceil(c/CPB), b); dynamic LDS = CPB*n floats; n % 4 == 0, P % 4 == 0.
// QueryAndGroup's relative-xyz channels ride in the same launch: with xyz != nullptr the grid has three
// more row blocks (gridDim.y = ceil(c/CPB) + 3), block ceil(c/CPB) + k stages coordinate k of the cloud
// (AoS (b,n,3), strided read) and writes out[b][k][p] = xyz[idx[p]][k] - new_xyz[p / nsample][k]; the
// feature rows then start at channel 3 (pointnet2_utils.py:311-321: grouped_xyz -= new_xyz; cat).
// One position stream of a launch: a QueryAndGroup scale (idx, its output, P = npoint * nsample positions cut into
// spans of pchunk).  A launch carries one or two of them: the two scales of an MSG level gather from the SAME staged
// rows, so the pair launch stages every row group once instead of twice (and is one launch instead of two).
struct GpScale {
  const int* idx;
  float* out;
  size_t out_batch_stride;
  int P, pchunk, nsample;
};

template <int CPB, int NTHR>
__global__ __launch_bounds__(NTHR) void group_points_rows_kernel(
    int c, int n, int nscale, GpScale sc0, GpScale sc1, const float* __restrict__ points,
    const float* __restrict__ xyz, const float* __restrict__ new_xyz) {
  extern __shared__ float s_row[];  // [CPB][n]
  const int tid = threadIdx.x;
  // XCD-aware mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in
  // linear-id order; keep every workgroup of a cloud on one XCD so that the cloud's idx array,
  // re-read once per CPB channels, is fetched from HBM once instead of once per XCD.
  int bi = blockIdx.z, by = blockIdx.y, bx = blockIdx.x;
  if ((gridDim.z & 7) == 0) {
    const unsigned per = gridDim.x * gridDim.y;
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned q = lin >> 3, w = q % per;
    bi = (int)(lin & 7) + 8 * (int)(q / per);
    by = (int)(w / gridDim.x);
    bx = (int)(w % gridDim.x);
  }
  typedef float v4f __attribute__((ext_vector_type(4)));
  const int feat_rows = (c + CPB - 1) / CPB;
  if (by >= feat_rows) {          // relative-xyz channel k (only launched with xyz != nullptr)
    const int k = by - feat_rows;
    const float* xb = xyz + (size_t)bi * n * 3 + k;
    for (int q = tid; q < n; q += NTHR) s_row[q] = xb[(size_t)q * 3];
    __syncthreads();
    for (int si = 0; si < nscale; ++si) {
      // (field by field: a reference to `si ? sc1 : sc0` would be a pointer select and push the arguments to scratch)
      const int P = si ? sc1.P : sc0.P, pchunk = si ? sc1.pchunk : sc0.pchunk, nsample = si ? sc1.nsample : sc0.nsample;
      const int* ip = (si ? sc1.idx : sc0.idx) + (size_t)bi * P;
      const float* cb = new_xyz + (size_t)bi * (P / nsample) * 3 + k;
      float* o = (si ? sc1.out : sc0.out) + (size_t)bi * (si ? sc1.out_batch_stride : sc0.out_batch_stride) + (size_t)k * P;
      const int p_end = min(bx * pchunk + pchunk, P);
      for (int p = bx * pchunk + tid * 4; p < p_end; p += 4 * NTHR) {
        const int4 id = *reinterpret_cast<const int4*>(ip + p);
        v4f val;
        if ((nsample & 3) == 0) {      // the four positions share one centre
          const float cc = cb[(size_t)(p / nsample) * 3];
          val = v4f{s_row[id.x] - cc, s_row[id.y] - cc, s_row[id.z] - cc, s_row[id.w] - cc};
        } else {
          val = v4f{s_row[id.x] - cb[(size_t)(p / nsample) * 3], s_row[id.y] - cb[(size_t)((p + 1) / nsample) * 3],
                    s_row[id.z] - cb[(size_t)((p + 2) / nsample) * 3], s_row[id.w] - cb[(size_t)((p + 3) / nsample) * 3]};
        }
        __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + p));
      }
    }
    return;
  }
  const int c0 = by * CPB;
  const int nc = min(CPB, c - c0);
  const int n4 = n >> 2;
  const float4* row = reinterpret_cast<const float4*>(points + ((size_t)bi * c + c0) * n);
  float4* s4 = reinterpret_cast<float4*>(s_row);
  for (int q = tid; q < nc * n4; q += NTHR) s4[q] = row[q];
  __syncthreads();
  for (int si = 0; si < nscale; ++si) {
  const int P = si ? sc1.P : sc0.P, pchunk = si ? sc1.pchunk : sc0.pchunk;
  const int p_begin = bx * pchunk;
  if (p_begin >= P) continue;
  const int p_end = min(p_begin + pchunk, P);
  const int* ip = (si ? sc1.idx : sc0.idx) + (size_t)bi * P;
  float* o = (si ? sc1.out : sc0.out) + (size_t)bi * (si ? sc1.out_batch_stride : sc0.out_batch_stride) +
             (size_t)(c0 + (xyz ? 3 : 0)) * P;
  // idx is loaded one iteration ahead: gfx950 counts loads and stores in the same in-order
  // vmcnt, so waiting for an idx load issued AFTER the previous iteration's stores would wait
  // for their write acknowledgements too; issued before them it only needs vmcnt(#stores).
  int p = p_begin + tid * 4;
  int4 id = *reinterpret_cast<const int4*>(ip + min(p, P - 4));
  if (CPB == 1) {
    // One row of a large cloud (48 KiB of LDS at n = 12288).  gfx950 counts loads and stores in one in-order
    // vmcnt, so waiting for the idx vectors requested ahead also waits for the PREVIOUS trip's stores to be
    // acknowledged by HBM (~2.5 us per trip, measured): the bytes a CU keeps in flight are what bounds this
    // kernel, hence 1024 threads on the staged row (two workgroups = 32 waves per CU) and two position vectors
    // per thread and trip.
    typedef int v4i __attribute__((ext_vector_type(4)));
    constexpr int U = 2, STEP = 4 * NTHR;
    v4i idv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) idv[u] = *reinterpret_cast<const v4i*>(ip + min(p + STEP * u, P - 4));
    while (p + STEP * (U - 1) < p_end) {
      const int pn = p + STEP * U;
      v4i idn[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int* nptr = ip + min(pn + STEP * u, P - 4);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(idn[u]) : "v"(nptr) : "memory");
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        v4f val = {s_row[idv[u].x], s_row[idv[u].y], s_row[idv[u].z], s_row[idv[u].w]};
        __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + p + STEP * u));
      }
      asm volatile("s_waitcnt vmcnt(2)" : "+v"(idn[0]), "+v"(idn[1]) : : "memory");
#pragma unroll
      for (int u = 0; u < U; ++u) idv[u] = idn[u];
      p = pn;
    }
    for (; p < p_end; p += STEP) {
      id = *reinterpret_cast<const int4*>(ip + p);
      v4f val = {s_row[id.x], s_row[id.y], s_row[id.z], s_row[id.w]};
      __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + p));
    }
  } else if (nc == CPB) {
    // Full row group: the next idx vector is requested BEFORE this iteration's stores and awaited
    // after them with s_waitcnt vmcnt(CPB) -- loads and stores share one in-order counter on
    // gfx950, so "CPB younger operations may be outstanding" is exactly "the idx load has
    // landed, the CPB stores need not have".  The compiler's own bookkeeping merges the loop
    // entry state into the loop and would wait for the stores' acknowledgements every
    // iteration (vmcnt(1)); hence the load and the wait are issued by hand.
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i idv = {id.x, id.y, id.z, id.w};
    while (p < p_end) {
      const int pn = p + 4 * NTHR;
      const int* nptr = ip + min(pn, P - 4);
      v4i idn;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(idn) : "v"(nptr) : "memory");
#pragma unroll
      for (int u = 0; u < CPB; ++u) {
        const float* sr = s_row + u * n;
        v4f val = {sr[idv.x], sr[idv.y], sr[idv.z], sr[idv.w]};
        // streaming output: written once, never re-read by this kernel
        __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + (size_t)u * P + p));
      }
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(idn) : "n"(CPB) : "memory");
      idv = idn;
      p = pn;
    }
  } else {
    for (; p < p_end; p += 4 * NTHR) {
      id = *reinterpret_cast<const int4*>(ip + p);
      for (int u = 0; u < nc; ++u) {
        const float* sr = s_row + u * n;
        v4f val = {sr[id.x], sr[id.y], sr[id.z], sr[id.w]};
        __builtin_nontemporal_store(val, reinterpret_cast<v4f*>(o + (size_t)u * P + p));
      }
    }
  }
  }   // scales
}

// any-P fallback, one output element per thread.  grid: (ceil(P/256), c, b)
__global__ __launch_bounds__(256) void group_points_scalar_kernel(
    int c, int n, int P, const float* __restrict__ points, const int* __restrict__ idx,
    float* __restrict__ out, size_t out_batch_stride) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  out[(size_t)bi * out_batch_stride + (size_t)l * P + p] =
      points[((size_t)bi * c + l) * n + idx[(size_t)bi * P + p]];
}

// channels [0,3) of QueryAndGroup's output: xyz[idx] - new_xyz[j].  xyz is (b,n,3) AoS.
// grid: (ceil(P/256), 1, b)
__global__ __launch_bounds__(256) void group_xyz_rel_kernel(
    int n, int m, int nsample, const float* __restrict__ xyz, const float* __restrict__ new_xyz,
    const int* __restrict__ idx, float* __restrict__ out, size_t out_batch_stride) {
  const int P = m * nsample;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int bi = blockIdx.z;
  const int j = p / nsample;
  const int k = idx[(size_t)bi * P + p];
  const float* q = xyz + ((size_t)bi * n + k) * 3;
  const float* cq = new_xyz + ((size_t)bi * m + j) * 3;
  float* o = out + (size_t)bi * out_batch_stride + p;
  o[0] = q[0] - cq[0];
  o[(size_t)P] = q[1] - cq[1];
  o[(size_t)2 * P] = q[2] - cq[2];
}

// grid: (ceil(P/256), c, b)
__global__ __launch_bounds__(256) void group_points_grad_kernel(
    int c, int n, int P, const float* __restrict__ grad_out, const int* __restrict__ idx,
    float* __restrict__ grad_points) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int l = blockIdx.y, bi = blockIdx.z;
  atomicAdd(grad_points + ((size_t)bi * c + l) * n + idx[(size_t)bi * P + p],
            grad_out[((size_t)bi * c + l) * P + p]);
}

// xyz / new_xyz != nullptr: also write the three relative-xyz channels in front of the c feature channels
// (`out` is then the (b, 3+c, P) tensor); returns 1 if that request could not be served by the row-owner
// kernel (the caller then uses the separate xyz kernel + a plain feature launch), 0 on success.
// Row-owner scatter, the rows kernel above run backwards: a workgroup owns CPB channel rows of one
// cloud's grad_points as LDS accumulators, streams a span of positions (idx and grad_out read with
// coalesced 16-byte loads) and adds into LDS with float atomics; rows leave with coalesced stores
// (one span) or one global atomic per touched element (several spans).  The element-per-thread kernel
// issues c * P global atomics per cloud -- 17 ms of a 65 ms training step over the four levels.
// grid: (n_pchunks, ceil(c/CPB), b); dynamic LDS = CPB*n floats; P % 4 == 0.
template <int CPB>
__global__ __launch_bounds__(256) void group_points_grad_rows_kernel(
    int c, int n, int P, int pchunk, const float* __restrict__ grad_out, const int* __restrict__ idx,
    float* __restrict__ grad_points) {
  extern __shared__ float s_acc[];  // [CPB][n]
  const int tid = threadIdx.x;
  const int bi = blockIdx.z, c0 = blockIdx.y * CPB;
  const int nc = min(CPB, c - c0);
  for (int q = tid; q < nc * n; q += 256) s_acc[q] = 0.f;
  __syncthreads();
  const int p_end = min(blockIdx.x * pchunk + pchunk, P);
  const int* ip = idx + (size_t)bi * P;
  const float* g = grad_out + ((size_t)bi * c + c0) * P;
  for (int p = blockIdx.x * pchunk + tid * 4; p < p_end; p += 1024) {
    const int4 id = *reinterpret_cast<const int4*>(ip + p);
#pragma unroll
    for (int u = 0; u < CPB; ++u) {
      if (u < nc) {
        const float4 gv = *reinterpret_cast<const float4*>(g + (size_t)u * P + p);
        float* row = s_acc + u * n;
        atomicAdd(row + id.x, gv.x);
        atomicAdd(row + id.y, gv.y);
        atomicAdd(row + id.z, gv.z);
        atomicAdd(row + id.w, gv.w);
      }
    }
  }
  __syncthreads();
  float* o = grad_points + ((size_t)bi * c + c0) * n;
  if (gridDim.x == 1) {
    for (int q = tid; q < nc * n; q += 256) o[q] = s_acc[q];
  } else {
    for (int q = tid; q < nc * n; q += 256) {
      const float v = s_acc[q];
      if (v != 0.f) atomicAdd(o + q, v);
    }
  }
}

// second scale (idx1 != nullptr): same points / xyz / new_xyz, its own idx, output and nsample
int launch_group(int b, int c, int n, int P, const float* points, const int* idx, float* out,
                 size_t out_batch_stride, hipStream_t st, const float* xyz = nullptr,
                 const float* new_xyz = nullptr, int nsample = 1, int P1 = 0, const int* idx1 = nullptr,
                 float* out1 = nullptr, size_t out_batch_stride1 = 0, int nsample1 = 1) {
  if (b <= 0 || c <= 0 || P <= 0) return 0;
  const bool pair = idx1 != nullptr;
  bool aligned = (P % 4 == 0) && (out_batch_stride % 4 == 0) &&
                 (((uintptr_t)out & 15) == 0) && (((uintptr_t)idx & 15) == 0);
  if (pair)
    aligned = aligned && (P1 % 4 == 0) && (out_batch_stride1 % 4 == 0) && (((uintptr_t)out1 & 15) == 0) &&
              (((uintptr_t)idx1 & 15) == 0);
  // (at n = 12288 -- level 0, 9 channels -- a workgroup stages a 48 KiB row for 16-32 KiB of output; the
  // direct-gather kernel below was measured there too and is slower still: 66 + 104 us vs 49 + 62 us)
  const bool rows_ok = aligned && (n % 4 == 0) && (size_t)n * 4 <= 128 * 1024 && (((uintptr_t)points & 15) == 0);
  if (pair && !rows_ok) return 2;      // the caller falls back to two single launches
  if (rows_ok) {
    // rows per workgroup: up to 8 rows in at most 32 KiB of LDS (idx is re-read once per row group, so
    // more rows per group = less L2 traffic per output byte; past 32 KiB the lost occupancy costs more).
    // Measured on MI355X, 64 clouds, out bytes / time: n=2048 -> 4 rows 5.3 TB/s (2 rows: 5.0-5.1),
    // n=1024 -> 8 rows 5.4-5.7 (4 rows: 5.1-5.5), n=512 -> 8 rows 4.4-5.2; tile-owner and direct-gather
    // variants measured 1.8-3.0 TB/s and were removed
    int cpb = 8;
    while (cpb > 1 && (size_t)cpb * n * 4 > 32 * 1024) cpb >>= 1;
    while (cpb > 1 && cpb > c) cpb >>= 1;
    const int rows = pvn3d_ceil_div(c, cpb) + (xyz ? 3 : 0);
    // split the position range until there are a few thousand workgroups (of 256 threads; the single-row
    // kernel runs 1024 threads on a row and wants long position spans per staged row)
    const int nthr = cpb == 1 ? 1024 : 256;
    int pch = pvn3d_ceil_div(cpb == 1 ? 1024 : 4096, rows * b);
    if (pch < 1) pch = 1;
    auto span = [&](int Ps, int& pchunk_s) {          // -> number of spans of this scale
      pchunk_s = pvn3d_ceil_div(pvn3d_ceil_div(Ps, pch), 4 * nthr) * 4 * nthr;
      if (pchunk_s < 16 * nthr) pchunk_s = 16 * nthr;
      return pvn3d_ceil_div(Ps, pchunk_s);
    };
    GpScale s0{idx, out, out_batch_stride, P, 0, nsample}, s1{idx1, out1, out_batch_stride1, P1, 0, nsample1};
    int gx = span(P, s0.pchunk);
    if (pair) gx = std::max(gx, span(P1, s1.pchunk));
    const size_t lds = (size_t)cpb * n * sizeof(float);
#define GP_LAUNCH(CPB, NTHR)                                                                      \
  do {                                                                                            \
    auto gk = group_points_rows_kernel<CPB, NTHR>;                                                \
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(gk));                                     \
    hipLaunchKernelGGL(gk, dim3(gx, rows, b), dim3(NTHR), lds, st, c, n, pair ? 2 : 1, s0, s1, points, xyz, new_xyz); \
  } while (0)
    if (cpb == 8) GP_LAUNCH(8, 256); else if (cpb == 4) GP_LAUNCH(4, 256); else if (cpb == 2) GP_LAUNCH(2, 256); else GP_LAUNCH(1, 1024);
#undef GP_LAUNCH
  } else if (xyz) {
    return 1;
  } else if (aligned) {
    const int gx = pvn3d_ceil_div(P, 1024);
    // split channels until there are a few thousand workgroups (256 CUs x 8)
    int chunks = pvn3d_ceil_div(2048, gx * b);
    if (chunks < 1) chunks = 1;
    if (chunks > c) chunks = c;
    const int cch = pvn3d_ceil_div(c, chunks);
    chunks = pvn3d_ceil_div(c, cch);
    hipLaunchKernelGGL(group_points_vec4_kernel, dim3(gx, chunks, b), dim3(256), 0, st, c, n, P,
                       cch, points, idx, out, out_batch_stride);
  } else {
    hipLaunchKernelGGL(group_points_scalar_kernel, dim3(pvn3d_ceil_div(P, 256), c, b),
                       dim3(256), 0, st, c, n, P, points, idx, out, out_batch_stride);
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int pvn3d_group_points(int b, int c, int n, int npoints, int nsample,
                                  const float* points, const int* idx, float* out,
                                  void* stream) {
  const int P = npoints * nsample;
  return launch_group(b, c, n, P, points, idx, out, (size_t)c * P, (hipStream_t)stream);
}

extern "C" int pvn3d_group_xyz_features(int b, int n, int m, int c, int nsample, int use_xyz,
                                        const float* xyz, const float* new_xyz,
                                        const float* features, const int* idx, float* out,
                                        void* stream) {
  if (b <= 0 || m <= 0 || nsample <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int P = m * nsample;
  const int c_out = (use_xyz ? 3 : 0) + (features ? c : 0);
  if (c_out == 0) return (int)hipErrorInvalidValue;
  const size_t bstride = (size_t)c_out * P;
  if (use_xyz && features && c > 0) {      // one launch: feature rows + the three relative-xyz rows
    const int rc = launch_group(b, c, n, P, features, idx, out, bstride, st, xyz, new_xyz, nsample);
    if (rc != 1) return rc;
  }
  if (use_xyz) {
    hipLaunchKernelGGL(group_xyz_rel_kernel, dim3(pvn3d_ceil_div(P, 256), 1, b), dim3(256), 0, st,
                       n, m, nsample, xyz, new_xyz, idx, out, bstride);
    PVN3D_LAUNCH_CHECK();
  }
  if (features && c > 0)
    return launch_group(b, c, n, P, features, idx, out + (use_xyz ? (size_t)3 * P : 0), bstride,
                        st);
  return 0;
}

extern "C" int pvn3d_group_xyz_features_pair(int b, int n, int m, int c, int nsample0, int nsample1,
                                             const float* xyz, const float* new_xyz, const float* features,
                                             const int* idx0, const int* idx1, float* out0, float* out1,
                                             void* stream) {
  if (b <= 0 || m <= 0 || nsample0 <= 0 || nsample1 <= 0) return 0;
  if (!xyz || !new_xyz || !features || c <= 0 || !idx0 || !idx1 || !out0 || !out1) return (int)hipErrorInvalidValue;
  const int rc = launch_group(b, c, n, m * nsample0, features, idx0, out0, (size_t)(3 + c) * m * nsample0,
                              (hipStream_t)stream, xyz, new_xyz, nsample0, m * nsample1, idx1, out1,
                              (size_t)(3 + c) * m * nsample1, nsample1);
  if (rc != 1 && rc != 2) return rc;
  // shapes the row kernel does not take: the two single-scale calls
  const int r0 = pvn3d_group_xyz_features(b, n, m, c, nsample0, 1, xyz, new_xyz, features, idx0, out0, stream);
  if (r0) return r0;
  return pvn3d_group_xyz_features(b, n, m, c, nsample1, 1, xyz, new_xyz, features, idx1, out1, stream);
}

extern "C" int pvn3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                       const float* grad_out, const int* idx,
                                       float* grad_points, void* stream) {
  if (b <= 0 || c <= 0 || n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int P = npoints * nsample;
  if (P > 0 && (P % 4 == 0) && (size_t)n * 4 <= 128 * 1024 && (((uintptr_t)grad_out & 15) == 0) &&
      (((uintptr_t)idx & 15) == 0)) {
    int cpb = 4;
    while (cpb > 1 && (size_t)cpb * n * 4 > 32 * 1024) cpb >>= 1;
    while (cpb > 1 && cpb > c) cpb >>= 1;
    const int rows = pvn3d_ceil_div(c, cpb);
    // split the positions only when the rows alone do not fill the chip
    int pch = pvn3d_ceil_div(2048, rows * b);
    if (pch < 1) pch = 1;
    int pchunk = pvn3d_ceil_div(pvn3d_ceil_div(P, pch), 1024) * 1024;
    if (pchunk < 4096) pchunk = 4096;
    pch = pvn3d_ceil_div(P, pchunk);
    if (pch > 1) PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, st));
    const size_t lds = (size_t)cpb * n * sizeof(float);
    switch (cpb) {
      case 4:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(group_points_grad_rows_kernel<4>));
        hipLaunchKernelGGL(group_points_grad_rows_kernel<4>, dim3(pch, rows, b), dim3(256), lds, st, c, n, P, pchunk,
                           grad_out, idx, grad_points);
        break;
      case 2:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(group_points_grad_rows_kernel<2>));
        hipLaunchKernelGGL(group_points_grad_rows_kernel<2>, dim3(pch, rows, b), dim3(256), lds, st, c, n, P, pchunk,
                           grad_out, idx, grad_points);
        break;
      default:
        PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(group_points_grad_rows_kernel<1>));
        hipLaunchKernelGGL(group_points_grad_rows_kernel<1>, dim3(pch, rows, b), dim3(256), lds, st, c, n, P, pchunk,
                           grad_out, idx, grad_points);
        break;
    }
    PVN3D_LAUNCH_CHECK();
    return 0;
  }
  PVN3D_RETURN_IF_ERR(hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, st));
  if (P <= 0) return 0;
  hipLaunchKernelGGL(group_points_grad_kernel, dim3(pvn3d_ceil_div(P, 256), c, b), dim3(256), 0,
                     st, c, n, P, grad_out, idx, grad_points);
  PVN3D_LAUNCH_CHECK();
  return 0;
}
