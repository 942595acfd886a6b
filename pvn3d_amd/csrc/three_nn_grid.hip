// three_nn_grid.hip -- exact 3-nearest-neighbour search through a uniform grid of the KNOWN points.
//
// Same results, bit for bit, as three_nn_kernel in interpolate.hip (and therefore as
// three_nn_kernel, pvn3d/_ext-src/src/interpolate_gpu.cu:9-59): for every unknown point the three
// known points with the smallest fp32 d = ((dx*dx + dy*dy) + dz*dz), ties to the smaller index
// (the reference scans k in ascending order with strict '<').  The brute-force kernel evaluates
// n*m pairs (25 M per frame at PVN3D's first feature-propagation level); here
//   build  (one workgroup per cloud): bounding box -> cell size h = 1.6 * sqrt(A_max / m)
//          (A_max = largest bounding-box face: the known points are a furthest-point sample of a
//          surface, whose spacing is ~sqrt(area / m)); 16^3 toroidal bucket table as in
//          ball_query_grid.hip: LDS histogram -> scan -> scatter of (x, y, z, k);
//   query  (one lane per unknown point; the sorted cloud and the bucket table sit in LDS): the 27
//          neighbour buckets are scanned with exactly the brute-force arithmetic and a
//          lexicographic (d, k) insertion, so the visiting order does not matter.  Every known
//          point NOT in those buckets is more than h away along some axis, so if the third best
//          distance is <= (0.999 h)^2 the result is already exact; otherwise (sparse regions,
//          volumetric clouds) the lane falls back to scanning all m points.  Aliased far cells of
//          the toroidal table only add candidates.
// Scratch is passed in by the caller; m <= 2048 (the sorted cloud must fit in LDS).

#include "common.h"

namespace {

constexpr int NG_AL = 4;
constexpr int NG_T = 1 << (3 * NG_AL);       // 4096 buckets
constexpr int NG_MAX_M = 2048;
constexpr int NG_MIN_M = 64;

struct NgWs {
  int* cell_start;    // [b][NG_T + 1]
  float4* sorted;     // [b][m]  (x, y, z, bits(k))
  float* hinfo;       // [b][2]  inv_h, (0.999 h)^2
};

inline size_t ng_layout(int b, int m, char* base, NgWs* ws) {
  const size_t cs = ((size_t)b * (NG_T + 1) * sizeof(int) + 255) / 256 * 256;
  const size_t so = ((size_t)b * m * sizeof(float4) + 255) / 256 * 256;
  const size_t hi = (size_t)b * 2 * sizeof(float);
  if (ws) {
    ws->cell_start = (int*)base;
    ws->sorted = (float4*)(base + cs);
    ws->hinfo = (float*)(base + cs + so);
  }
  return cs + so + hi;
}

__device__ __forceinline__ int ng_bucket_c(int cx, int cy, int cz) {
  constexpr int M = (1 << NG_AL) - 1;
  return (cx & M) | ((cy & M) << NG_AL) | ((cz & M) << (2 * NG_AL));
}
__device__ __forceinline__ int ng_cell(float x, float inv_h) { return (int)floorf(x * inv_h); }

// one workgroup (1024 threads) per cloud
__global__ __launch_bounds__(1024) void nn_grid_build_kernel(int m, float h_scale, const float* __restrict__ known,
                                                             int* __restrict__ cell_start,
                                                             float4* __restrict__ sorted,
                                                             float* __restrict__ hinfo) {
  constexpr int PER = NG_T / 1024;
  __shared__ int s_cnt[NG_T + (NG_T >> 5)];
  __shared__ int s_part[16];
  __shared__ float s_box[6][16];
  __shared__ float s_invh;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  known += (size_t)blockIdx.x * m * 3;
  cell_start += (size_t)blockIdx.x * (NG_T + 1);
  sorted += (size_t)blockIdx.x * m;
  auto pad = [](int c) { return c + (c >> 5); };
  for (int i = tid; i < NG_T + (NG_T >> 5); i += 1024) s_cnt[i] = 0;
  // bounding box
  float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
  for (int k = tid; k < m; k += 1024) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = known[k * 3 + a];
      lo[a] = fminf(lo[a], v);
      hi[a] = fmaxf(hi[a], v);
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], o, 64));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o, 64));
    }
    if (lane == 0) { s_box[a][wv] = lo[a]; s_box[3 + a][wv] = hi[a]; }
  }
  __syncthreads();
  if (tid == 0) {
    float e[3];
    for (int a = 0; a < 3; ++a) {
      float l = s_box[a][0], h = s_box[3 + a][0];
      for (int w = 1; w < 16; ++w) { l = fminf(l, s_box[a][w]); h = fmaxf(h, s_box[3 + a][w]); }
      e[a] = fmaxf(h - l, 0.f);
    }
    const float area = fmaxf(e[0] * e[1], fmaxf(e[1] * e[2], e[0] * e[2]));
    float h = h_scale * sqrtf(area / (float)m);
    const float emax = fmaxf(e[0], fmaxf(e[1], e[2]));
    if (!(h > 1e-12f)) h = fmaxf(emax * 0.125f, 1e-6f);     // collinear / coincident clouds
    if (!(h < 3.0e37f)) h = 1.0f;                           // non-finite coordinates: any cell size
    s_invh = 1.0f / h;
    hinfo[blockIdx.x * 2 + 0] = 1.0f / h;
    hinfo[blockIdx.x * 2 + 1] = (0.999f * h) * (0.999f * h);
  }
  __syncthreads();
  const float inv_h = s_invh;
  for (int k = tid; k < m; k += 1024)
    atomicAdd(&s_cnt[pad(ng_bucket_c(ng_cell(known[k * 3], inv_h), ng_cell(known[k * 3 + 1], inv_h),
                                     ng_cell(known[k * 3 + 2], inv_h)))], 1);
  __syncthreads();
  int local = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) local += s_cnt[pad(tid * PER + i)];
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_part[wv] = incl;
  __syncthreads();
  int wave_off = 0;
  for (int w = 0; w < wv; ++w) wave_off += s_part[w];
  int run = wave_off + incl - local;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = s_cnt[pad(tid * PER + i)];
    s_cnt[pad(tid * PER + i)] = run;      // becomes the scatter cursor
    run += c;
  }
  if (tid == 1023) cell_start[NG_T] = run;
  __syncthreads();
  // (bucket order = coalesced stores; see grid_build_kernel in ball_query_grid.hip)
#pragma unroll
  for (int i = 0; i < PER; ++i) cell_start[tid + 1024 * i] = s_cnt[pad(tid + 1024 * i)];
  __syncthreads();
  for (int k = tid; k < m; k += 1024) {
    const float x = known[k * 3], y = known[k * 3 + 1], z = known[k * 3 + 2];
    const int pos = atomicAdd(&s_cnt[pad(ng_bucket_c(ng_cell(x, inv_h), ng_cell(y, inv_h), ng_cell(z, inv_h)))], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(k));
  }
}

struct Top3 {
  float b1, b2, b3;
  int i1, i2, i3;
  __device__ __forceinline__ void reset() {
    b1 = b2 = b3 = __builtin_inff();
    i1 = i2 = i3 = 0;
  }
  // (d, k) lexicographic insertion == the reference's strict '<' scan in ascending k
  __device__ __forceinline__ void insert(float d, int k) {
    const bool lt3 = d < b3 || (d == b3 && k < i3);
    if (!lt3) return;
    const bool lt2 = d < b2 || (d == b2 && k < i2);
    const bool lt1 = d < b1 || (d == b1 && k < i1);
    if (lt1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
    else if (lt2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
    else { b3 = d; i3 = k; }
  }
};

// grid (ceil(n / (256*QPT)), b), block 256; dynamic LDS = (NG_T + 1) ints + m float4
template <int QPT>
__global__ __launch_bounds__(256) void three_nn_grid_kernel(int n, int m, const float* __restrict__ unknown,
                                                            const int* __restrict__ cell_start,
                                                            const float4* __restrict__ sorted,
                                                            const float* __restrict__ hinfo,
                                                            float* __restrict__ dist2, int* __restrict__ idx) {
  extern __shared__ float4 s_dyn[];
  float4* s_pts = s_dyn;                                          // [m]
  int* s_start = reinterpret_cast<int*>(s_dyn + m);               // [NG_T + 1]
  const int tid = threadIdx.x, lane = tid & 63;
  int bi, bx;
  pvn3d_xcd_frame_map(bi, bx);          // a cloud's workgroups on one XCD: its bucket table is fetched from HBM once
  cell_start += (size_t)bi * (NG_T + 1);
  sorted += (size_t)bi * m;
  unknown += (size_t)bi * n * 3;
  for (int i = tid; i < m; i += 256) s_pts[i] = sorted[i];
  for (int i = tid; i <= NG_T; i += 256) s_start[i] = cell_start[i];
  const float inv_h = hinfo[bi * 2 + 0], safe2 = hinfo[bi * 2 + 1];
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < QPT; ++q) {
    const int j = (bx * QPT + q) * 256 + tid;
    const bool live = j < n;          // dead lanes still help in the cooperative fallback below
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (live) { ux = unknown[j * 3 + 0]; uy = unknown[j * 3 + 1]; uz = unknown[j * 3 + 2]; }
    const int cx = ng_cell(ux, inv_h), cy = ng_cell(uy, inv_h), cz = ng_cell(uz, inv_h);
    Top3 t;
    t.reset();
    if (!live) t.b3 = 0.f;            // never "bad"
    if (live) {
      for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int bk = ng_bucket_c(cx + dx, cy + dy, cz + dz);
            const int e = s_start[bk + 1];
            for (int p = s_start[bk]; p < e; ++p) {
              const float4 a = s_pts[p];
              const float ex = ux - a.x, ey = uy - a.y, ez = uz - a.z;
              t.insert(ex * ex + ey * ey + ez * ez, __float_as_int(a.w));
            }
          }
    }
    // Not provably complete (sparse region, NaN query): the WAVE scans all m points for each such
    // query cooperatively -- lane l takes points l, l+64, ..., then the three winners are pulled
    // out by three lexicographic (d, k) wave-minimum rounds.  A per-lane scan would make the whole
    // wave wait 2048 iterations for a single failing lane.
    unsigned long long bad = __ballot(!(t.b3 <= safe2));
    while (bad) {
      const int src = __builtin_ctzll(bad);
      bad &= bad - 1;
      const float qx = __shfl(ux, src, 64), qy = __shfl(uy, src, 64), qz = __shfl(uz, src, 64);
      Top3 loc;
      loc.reset();
      for (int p = lane; p < m; p += 64) {
        const float4 a = s_pts[p];
        const float ex = qx - a.x, ey = qy - a.y, ez = qz - a.z;
        loc.insert(ex * ex + ey * ey + ez * ez, __float_as_int(a.w));
      }
      Top3 res;
      res.reset();
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        float d = loc.b1;
        int k = loc.b1 < __builtin_inff() ? loc.i1 : 0x7fffffff;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          const float od_ = __shfl_xor(d, o, 64);
          const int ok_ = __shfl_xor(k, o, 64);
          if (od_ < d || (od_ == d && ok_ < k)) { d = od_; k = ok_; }
        }
        // every lane now holds the wave minimum; its owner pops it
        if (d < __builtin_inff()) {
          if (loc.b1 == d && loc.i1 == k) { loc.b1 = loc.b2; loc.i1 = loc.i2; loc.b2 = loc.b3; loc.i2 = loc.i3; loc.b3 = __builtin_inff(); loc.i3 = 0; }
          if (r == 0) { res.b1 = d; res.i1 = k; } else if (r == 1) { res.b2 = d; res.i2 = k; } else { res.b3 = d; res.i3 = k; }
        }
      }
      if (lane == src) t = res;
    }
    if (live) {
      float* od = dist2 + ((size_t)bi * n + j) * 3;
      int* oi = idx + ((size_t)bi * n + j) * 3;
      od[0] = t.b1; od[1] = t.b2; od[2] = t.b3;
      oi[0] = t.i1; oi[1] = t.i2; oi[2] = t.i3;
    }
  }
}

}  // namespace

extern "C" size_t pvn3d_three_nn_grid_workspace_bytes(int b, int m) {
  if (b <= 0 || m <= 0) return 0;
  return ng_layout(b, m, nullptr, nullptr);
}

extern "C" int pvn3d_three_nn_grid(int b, int n, int m, const float* unknown, const float* known,
                                   float* dist2, int* idx, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (b <= 0 || n <= 0) return 0;
  if (m < NG_MIN_M || m > NG_MAX_M || !unknown || !known || !dist2 || !idx || !workspace ||
      workspace_bytes < pvn3d_three_nn_grid_workspace_bytes(b, m))
    return (int)hipErrorInvalidValue;
  hipStream_t st = (hipStream_t)stream;
  NgWs ws;
  ng_layout(b, m, (char*)workspace, &ws);
  // cell size in units of the estimated point spacing; measured 1.2 / 1.5 / 1.8 / 2.2 / 2.8 ->
  // 0.63 / 0.38 / 0.39 / 0.43 / 0.49 ms per 64-frame step
  const float h_scale = 1.6f;
  hipLaunchKernelGGL(nn_grid_build_kernel, dim3(b), dim3(1024), 0, st, m, h_scale, known, ws.cell_start,
                     ws.sorted, ws.hinfo);
  PVN3D_LAUNCH_CHECK();
  const size_t lds = (size_t)m * sizeof(float4) + (NG_T + 1) * sizeof(int);
  if (n >= 4096) {
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_nn_grid_kernel<4>));
    hipLaunchKernelGGL(three_nn_grid_kernel<4>, dim3(pvn3d_ceil_div(n, 1024), b), dim3(256), lds, st, n, m, unknown,
                       ws.cell_start, ws.sorted, ws.hinfo, dist2, idx);
  } else {
    PVN3D_RETURN_IF_ERR((hipError_t)pvn3d_allow_big_lds(three_nn_grid_kernel<1>));
    hipLaunchKernelGGL(three_nn_grid_kernel<1>, dim3(pvn3d_ceil_div(n, 256), b), dim3(256), lds, st, n, m, unknown,
                       ws.cell_start, ws.sorted, ws.hinfo, dist2, idx);
  }
  PVN3D_LAUNCH_CHECK();
  return 0;
}
