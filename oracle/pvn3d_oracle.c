/*
 * pvn3d_oracle.c -- CPU restatement of the PVN3D hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the *checker*, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call it.  The shipped path is the HIP
 * library (pvn3d_amd/csrc); it never falls back to this code.
 *
 * Every function restates, in plain C, the algorithm of one reference function and
 * cites the reference file:line it follows (paths relative to the reference root).
 *
 * Canonical arithmetic: IEEE fp32, one rounding per source-level operation, evaluated
 * in the reference's source order with NO fused multiply-add.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math      (see oracle/Makefile)
 * The reference's CUDA 9 binary (nvcc default -fmad=true) may have contracted some of
 * these expressions into FMAs; that cannot be observed here (no NVIDIA GPU, no nvcc),
 * so the C-source semantics are the pinned form.  See DESIGN.md "Canonical arithmetic".
 *
 * Parity status of the native ops: the reference ships no golden vectors / asserting
 * tests for pvn3d/_ext-src ("parity unpinned" by the reference's own tests); these
 * restatements are additionally cross-checked against independent brute-force
 * formulations in tests/test_oracle_native.py.  MeanShift and best_fit_transform ARE
 * pinned: the npz files under tests/golden hold outputs of the reference's own Python
 * (meanshift_pytorch.py / basic_utils.py) run in the build container.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------
 * Launch heuristic: pvn3d/_ext-src/include/cuda_utils.h:13-19  (opt_n_threads)
 *   pow_2 = (int)(log(work_size)/log(2.0));  return max(min(1<<pow_2, 512), 1)
 * ---------------------------------------------------------------------------------- */
int orc_opt_n_threads(int work_size) {
  const int pow_2 = (int)(log((double)work_size) / log(2.0));
  int v = 1 << pow_2;
  if (v > 512) v = 512;
  if (v < 1) v = 1;
  return v;
}

/* ------------------------------------------------------------------------------------
 * furthest_point_sampling
 *   init      : pvn3d/_ext-src/src/sampling.cpp:65-86  (idx zeros, temp filled 1e10)
 *   kernel    : pvn3d/_ext-src/src/sampling_gpu.cu:69-173 (+ __update :59-65)
 *   block size: sampling_gpu.cu:178 -> opt_n_threads(n)
 * Literal emulation of the thread block: `bs` virtual threads, thread t scans
 * k = t, t+bs, ... with strict '>' (first k wins), then the shared-memory halving
 * tree where ties keep the lower slot.  The tie-break therefore depends on bs and
 * the tree shape, which is why the block is emulated rather than replaced by an argmax.
 * dataset (b,n,3) ; temp (b,n) scratch, overwritten ; idxs (b,m)
 * ---------------------------------------------------------------------------------- */
void orc_furthest_point_sampling(int b, int n, int m, const float *dataset, float *temp,
                                 int *idxs) {
  if (m <= 0) return;
  const int bs = orc_opt_n_threads(n);
  float *dists = (float *)malloc(sizeof(float) * (size_t)bs);
  int *dists_i = (int *)malloc(sizeof(int) * (size_t)bs);
  for (int bi = 0; bi < b; ++bi) {
    const float *ds = dataset + (size_t)bi * n * 3;
    float *tp = temp + (size_t)bi * n;
    int *out = idxs + (size_t)bi * m;
    for (int k = 0; k < n; ++k) tp[k] = 1e10f; /* sampling.cpp:73-75 */
    int old = 0;
    out[0] = old;
    for (int j = 1; j < m; ++j) {
      const float x1 = ds[old * 3 + 0], y1 = ds[old * 3 + 1], z1 = ds[old * 3 + 2];
      for (int tid = 0; tid < bs; ++tid) {
        int besti = 0;
        float best = -1.0f;
        for (int k = tid; k < n; k += bs) {
          const float x2 = ds[k * 3 + 0], y2 = ds[k * 3 + 1], z2 = ds[k * 3 + 2];
          const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
          /* sampling_gpu.cu:101 compares float mag with the double literal 1e-3 */
          if ((double)mag <= 1e-3) continue;
          const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) +
                          (z2 - z1) * (z2 - z1);
          const float d2 = d < tp[k] ? d : tp[k]; /* min(d, temp[k]) */
          tp[k] = d2;
          besti = d2 > best ? k : besti;
          best = d2 > best ? d2 : best;
        }
        dists[tid] = best;
        dists_i[tid] = besti;
      }
      for (int s = bs / 2; s >= 1; s >>= 1) { /* sampling_gpu.cu:114-168 */
        for (int tid = 0; tid < s; ++tid) {
          const float v1 = dists[tid], v2 = dists[tid + s];
          const int i1 = dists_i[tid], i2 = dists_i[tid + s];
          dists[tid] = v1 > v2 ? v1 : v2;
          dists_i[tid] = v2 > v1 ? i2 : i1;
        }
      }
      old = dists_i[0];
      out[j] = old;
    }
  }
  free(dists);
  free(dists_i);
}

/* ------------------------------------------------------------------------------------
 * gather_points / gather_points_grad : pvn3d/_ext-src/src/sampling_gpu.cu:8-20, 34-47
 * points (b,c,n) idx (b,m) -> out (b,c,m) ;  grad: out (b,c,n) must be pre-zeroed
 * (sampling.cpp:54-56), accumulated in (j ascending) order here.
 * ---------------------------------------------------------------------------------- */
void orc_gather_points(int b, int c, int n, int m, const float *points, const int *idx,
                       float *out) {
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        out[((size_t)i * c + l) * m + j] = points[((size_t)i * c + l) * n + a];
      }
}

void orc_gather_points_grad(int b, int c, int n, int m, const float *grad_out,
                            const int *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int i = 0; i < b; ++i)
    for (int l = 0; l < c; ++l)
      for (int j = 0; j < m; ++j) {
        const int a = idx[(size_t)i * m + j];
        grad_points[((size_t)i * c + l) * n + a] += grad_out[((size_t)i * c + l) * m + j];
      }
}

/* ------------------------------------------------------------------------------------
 * ball_query : pvn3d/_ext-src/src/ball_query_gpu.cu:9-44 ; zero-init ball_query.cpp:19-21
 * new_xyz (b,m,3) xyz (b,n,3) -> idx (b,m,nsample).  radius2 = radius*radius in fp32.
 * ---------------------------------------------------------------------------------- */
void orc_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                    const float *xyz, int *idx) {
  const float radius2 = radius * radius;
  memset(idx, 0, sizeof(int) * (size_t)b * m * nsample);
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    for (int j = 0; j < m; ++j) {
      const float *px = xyz + (size_t)bi * n * 3;
      const float *pc = new_xyz + ((size_t)bi * m + j) * 3;
      int *o = idx + ((size_t)bi * m + j) * nsample;
      const float new_x = pc[0], new_y = pc[1], new_z = pc[2];
      for (int k = 0, cnt = 0; k < n && cnt < nsample; ++k) {
        const float x = px[k * 3 + 0], y = px[k * 3 + 1], z = px[k * 3 + 2];
        const float d2 = (new_x - x) * (new_x - x) + (new_y - y) * (new_y - y) +
                         (new_z - z) * (new_z - z);
        if (d2 < radius2) {
          if (cnt == 0)
            for (int l = 0; l < nsample; ++l) o[l] = k;
          o[cnt] = k;
          ++cnt;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------
 * group_points / group_points_grad : pvn3d/_ext-src/src/group_points_gpu.cu:8-28, 43-64
 * points (b,c,n) idx (b,npoints,nsample) -> out (b,c,npoints,nsample)
 * ---------------------------------------------------------------------------------- */
void orc_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                      const int *idx, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * npoints * nsample;
      float *o = out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          o[(size_t)j * nsample + k] = p[id[(size_t)j * nsample + k]];
    }
}

void orc_group_points_grad(int b, int c, int n, int npoints, int nsample,
                           const float *grad_out, const int *idx, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * n);
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      float *g = grad_points + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * npoints * nsample;
      const float *go = grad_out + ((size_t)bi * c + l) * npoints * nsample;
      for (int j = 0; j < npoints; ++j)
        for (int k = 0; k < nsample; ++k)
          g[id[(size_t)j * nsample + k]] += go[(size_t)j * nsample + k];
    }
}

/* ------------------------------------------------------------------------------------
 * three_nn : pvn3d/_ext-src/src/interpolate_gpu.cu:9-59
 * unknown (b,n,3) known (b,m,3) -> dist2 (b,n,3) fp32, idx (b,n,3).
 * Running bests are double initialised to 1e40 (interpolate_gpu.cu:29); d is fp32.
 * ---------------------------------------------------------------------------------- */
void orc_three_nn(int b, int n, int m, const float *unknown, const float *known,
                  float *dist2, int *idx) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi) {
    for (int j = 0; j < n; ++j) {
      const float *un = unknown + ((size_t)bi * n + j) * 3;
      const float *kn = known + (size_t)bi * m * 3;
      const float ux = un[0], uy = un[1], uz = un[2];
      double best1 = 1e40, best2 = 1e40, best3 = 1e40;
      int besti1 = 0, besti2 = 0, besti3 = 0;
      for (int k = 0; k < m; ++k) {
        const float x = kn[k * 3 + 0], y = kn[k * 3 + 1], z = kn[k * 3 + 2];
        const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
        if (d < best1) {
          best3 = best2; besti3 = besti2;
          best2 = best1; besti2 = besti1;
          best1 = d;     besti1 = k;
        } else if (d < best2) {
          best3 = best2; besti3 = besti2;
          best2 = d;     besti2 = k;
        } else if (d < best3) {
          best3 = d;     besti3 = k;
        }
      }
      float *od = dist2 + ((size_t)bi * n + j) * 3;
      int *oi = idx + ((size_t)bi * n + j) * 3;
      od[0] = (float)best1; od[1] = (float)best2; od[2] = (float)best3;
      oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
    }
  }
}

/* ------------------------------------------------------------------------------------
 * three_interpolate : pvn3d/_ext-src/src/interpolate_gpu.cu:72-101
 * points (b,c,m) idx (b,n,3) weight (b,n,3) -> out (b,c,n)
 * ---------------------------------------------------------------------------------- */
void orc_three_interpolate(int b, int c, int m, int n, const float *points, const int *idx,
                           const float *weight, float *out) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = points + ((size_t)bi * c + l) * m;
      const int *id = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *o = out + ((size_t)bi * c + l) * n;
      for (int j = 0; j < n; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        o[j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
      }
    }
}

/* Mathematically correct gradient: the kernel the reference defines but never launches,
 * pvn3d/_ext-src/src/interpolate_gpu.cu:116-143.  grad_out (b,c,n) -> grad_points (b,c,m) */
void orc_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                const int *idx, const float *weight, float *grad_points) {
  memset(grad_points, 0, sizeof(float) * (size_t)b * c * m);
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *go = grad_out + ((size_t)bi * c + l) * n;
      const int *id = idx + (size_t)bi * n * 3;
      const float *w = weight + (size_t)bi * n * 3;
      float *g = grad_points + ((size_t)bi * c + l) * m;
      for (int j = 0; j < n; ++j)
        for (int t = 0; t < 3; ++t) g[id[j * 3 + t]] += go[j] * w[j * 3 + t];
    }
}

/* What the reference binary actually returns from three_interpolate_grad:
 * pvn3d/_ext-src/src/interpolate.cpp:89-93 calls the FORWARD wrapper with
 * (b, c, m:=n, n:=m, grad_out, idx, weight, out): out[b,c,j<m] = sum_t grad_out[b,c,idx[j,t]]*w[j,t]
 * with idx/weight strides still those of a (b,m,3) view of the first m rows per batch...
 * NOTE the forward kernel offsets idx/weight by batch_index*n_arg*3 with n_arg = m, so for
 * b>1 it reads rows [bi*m, bi*m+m) of the flattened (b*n,3) arrays, not batch bi's rows. */
void orc_three_interpolate_grad_refbug(int b, int c, int n, int m, const float *grad_out,
                                       const int *idx, const float *weight, float *out) {
  for (int bi = 0; bi < b; ++bi)
    for (int l = 0; l < c; ++l) {
      const float *p = grad_out + ((size_t)bi * c + l) * n; /* "points" (b,c,m_arg=n) */
      const int *id = idx + (size_t)bi * m * 3;             /* batch stride n_arg*3 = m*3 */
      const float *w = weight + (size_t)bi * m * 3;
      float *o = out + ((size_t)bi * c + l) * m;
      for (int j = 0; j < m; ++j) {
        const float w1 = w[j * 3 + 0], w2 = w[j * 3 + 1], w3 = w[j * 3 + 2];
        const int i1 = id[j * 3 + 0], i2 = id[j * 3 + 1], i3 = id[j * 3 + 2];
        o[j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
      }
    }
}

/* ------------------------------------------------------------------------------------
 * MeanShiftTorch.fit : pvn3d/lib/utils/meanshift_pytorch.py:13-51
 *   stop_thresh = bandwidth*1e-3 (:21) ; loop (:31-44) ; cluster pick (:45-51)
 *   gaussian_kernel (:13-15): (1/(bw*sqrt(2*pi))) * exp(-0.5*(dis/bw)^2)
 * A (n,3) fp32.  ctr_out[3], labels_out[n] (0/1), iters_out = number of iterations run.
 * The per-pair weight is evaluated in fp32 exactly as the torch expression is written
 * (norm -> /bw -> **2 -> *-0.5 -> exp -> *const); the two reductions over j accumulate in
 * double (torch's fp32 CPU sum is a vectorised cascade sum whose order is an
 * implementation detail of the installed torch; double accumulation is the order-free
 * stand-in).  Pinned against the reference's own outputs in tests/golden/meanshift_*.npz.
 * If C_final != NULL it receives the seed positions after the last iteration (n,3);
 * last_max_shift (optional) receives max(Adis) of the last iteration -- tests use it to tell
 * when the stop decision is marginal (a slowly creeping outlier within a few % of the
 * threshold makes the iteration COUNT chaotic in any fp32 implementation; the centre is not).
 * ---------------------------------------------------------------------------------- */
void orc_meanshift_fit(const float *A, int n, float bandwidth, int max_iter, float *ctr_out,
                       uint8_t *labels_out, int *iters_out, float *C_final,
                       float *last_max_shift) {
  const float stop_thresh = (float)((double)bandwidth * 1e-3);
  const float kconst = 1.0f / (bandwidth * sqrtf(2.0f * (float)M_PI));
  float *C = (float *)malloc(sizeof(float) * (size_t)n * 3);
  float *Cn = (float *)malloc(sizeof(float) * (size_t)n * 3);
  memcpy(C, A, sizeof(float) * (size_t)n * 3);
  int it = 0;
  float final_shift = 0.0f;
  while (1) {
    ++it;
    float max_adis = 0.0f;
#pragma omp parallel for schedule(static) reduction(max : max_adis)
    for (int i = 0; i < n; ++i) {
      const float cx = C[i * 3 + 0], cy = C[i * 3 + 1], cz = C[i * 3 + 2];
      double sw = 0.0, sx = 0.0, sy = 0.0, sz = 0.0;
      for (int j = 0; j < n; ++j) {
        const float ax = A[j * 3 + 0], ay = A[j * 3 + 1], az = A[j * 3 + 2];
        const float dx = cx - ax, dy = cy - ay, dz = cz - az;
        const float dis = sqrtf(dx * dx + dy * dy + dz * dz);
        const float q = dis / bandwidth;
        const float w = kconst * expf(-0.5f * (q * q));
        sw += (double)w;
        sx += (double)(w * ax);
        sy += (double)(w * ay);
        sz += (double)(w * az);
      }
      const float nx = (float)sx / (float)sw, ny = (float)sy / (float)sw,
                  nz = (float)sz / (float)sw;
      const float ex = nx - cx, ey = ny - cy, ez = nz - cz;
      const float adis = sqrtf(ex * ex + ey * ey + ez * ez);
      Cn[i * 3 + 0] = nx; Cn[i * 3 + 1] = ny; Cn[i * 3 + 2] = nz;
      if (adis > max_adis) max_adis = adis;
    }
    float *t = C; C = Cn; Cn = t;
    final_shift = max_adis;
    if (max_adis < stop_thresh || it > max_iter) break; /* :42 */
  }
  /* :46-51 -- note both operands are built from A: neighbour counts of ORIGINAL points */
  int best_cnt = -1, best_i = 0;
  for (int i = 0; i < n; ++i) {
    const float ax = A[i * 3 + 0], ay = A[i * 3 + 1], az = A[i * 3 + 2];
    int cnt = 0;
    for (int j = 0; j < n; ++j) {
      const float dx = A[j * 3 + 0] - ax, dy = A[j * 3 + 1] - ay, dz = A[j * 3 + 2] - az;
      const float dis = sqrtf(dx * dx + dy * dy + dz * dz);
      cnt += dis < bandwidth;
    }
    if (cnt > best_cnt) { best_cnt = cnt; best_i = i; } /* first max, torch>=1.7 */
  }
  {
    const float ax = A[best_i * 3 + 0], ay = A[best_i * 3 + 1], az = A[best_i * 3 + 2];
    for (int j = 0; j < n; ++j) {
      const float dx = A[j * 3 + 0] - ax, dy = A[j * 3 + 1] - ay, dz = A[j * 3 + 2] - az;
      const float dis = sqrtf(dx * dx + dy * dy + dz * dz);
      labels_out[j] = dis < bandwidth;
    }
  }
  ctr_out[0] = C[best_i * 3 + 0]; ctr_out[1] = C[best_i * 3 + 1]; ctr_out[2] = C[best_i * 3 + 2];
  if (iters_out) *iters_out = it;
  if (last_max_shift) *last_max_shift = final_shift; /* how close the stop decision was */
  if (C_final) memcpy(C_final, C, sizeof(float) * (size_t)n * 3);
  free(C);
  free(Cn);
}

/* ------------------------------------------------------------------------------------
 * best_fit_transform : pvn3d/lib/utils/basic_utils.py:47-80 (Kabsch).
 * A,B (n,3) fp32 -> T (3,4) double row-major = [R | t].
 * The reference calls LAPACK (np.linalg.svd) on the fp32 3x3 H; here the SVD is a
 * double-precision one-sided Jacobi with singular values sorted descending (LAPACK's
 * order, which decides WHICH row of Vt the reflection fix at :70-72 negates).
 * Pinned against the reference's outputs in tests/golden/kabsch_*.npz (tolerance 1e-5:
 * fp32 LAPACK vs fp64 Jacobi).
 * ---------------------------------------------------------------------------------- */
static void jacobi_svd3(const double H[9], double U[9], double S[3], double V[9]) {
  /* one-sided Jacobi on columns of G (= H), V accumulates rotations: H = U S V^T */
  double G[9];
  memcpy(G, H, sizeof(G));
  for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int r = 0; r < 3; ++r) {
          alpha += G[r * 3 + p] * G[r * 3 + p];
          beta += G[r * 3 + q] * G[r * 3 + q];
          gamma += G[r * 3 + p] * G[r * 3 + q];
        }
        off += fabs(gamma);
        if (fabs(gamma) <= 1e-300 || fabs(gamma) <= 1e-17 * sqrt(alpha * beta)) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int r = 0; r < 3; ++r) {
          const double gp = G[r * 3 + p], gq = G[r * 3 + q];
          G[r * 3 + p] = c * gp - s * gq;
          G[r * 3 + q] = s * gp + c * gq;
          const double vp = V[r * 3 + p], vq = V[r * 3 + q];
          V[r * 3 + p] = c * vp - s * vq;
          V[r * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-30) break;
  }
  int order[3] = {0, 1, 2};
  double nrm[3];
  for (int c = 0; c < 3; ++c)
    nrm[c] = sqrt(G[c] * G[c] + G[3 + c] * G[3 + c] + G[6 + c] * G[6 + c]);
  for (int a = 0; a < 2; ++a)
    for (int bb = a + 1; bb < 3; ++bb)
      if (nrm[order[bb]] > nrm[order[a]]) { int t = order[a]; order[a] = order[bb]; order[bb] = t; }
  double Vs[9];
  for (int c = 0; c < 3; ++c) {
    const int oc = order[c];
    S[c] = nrm[oc];
    for (int r = 0; r < 3; ++r) {
      Vs[r * 3 + c] = V[r * 3 + oc];
      U[r * 3 + c] = nrm[oc] > 0 ? G[r * 3 + oc] / nrm[oc] : 0.0;
    }
  }
  memcpy(V, Vs, sizeof(Vs));
  /* complete U to an orthonormal basis if rank-deficient (cross products) */
  if (S[2] <= 1e-14 * (S[0] > 0 ? S[0] : 1.0)) {
    if (S[1] <= 1e-14 * (S[0] > 0 ? S[0] : 1.0)) {
      /* rank <= 1: pick any vector orthogonal to U[:,0] */
      double u0[3] = {U[0], U[3], U[6]};
      if (S[0] <= 0) { u0[0] = 1; u0[1] = 0; u0[2] = 0; U[0] = 1; U[3] = 0; U[6] = 0; }
      double a[3] = {0, 0, 0};
      int mi = fabs(u0[0]) < fabs(u0[1]) ? (fabs(u0[0]) < fabs(u0[2]) ? 0 : 2)
                                         : (fabs(u0[1]) < fabs(u0[2]) ? 1 : 2);
      a[mi] = 1.0;
      double u1[3] = {u0[1] * a[2] - u0[2] * a[1], u0[2] * a[0] - u0[0] * a[2],
                      u0[0] * a[1] - u0[1] * a[0]};
      double l = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
      U[1] = u1[0] / l; U[4] = u1[1] / l; U[7] = u1[2] / l;
    }
    const double a0 = U[0], a1 = U[3], a2 = U[6], b0 = U[1], b1 = U[4], b2 = U[7];
    U[2] = a1 * b2 - a2 * b1; U[5] = a2 * b0 - a0 * b2; U[8] = a0 * b1 - a1 * b0;
  }
}

void orc_best_fit_transform(const float *A, const float *B, int n, double *T /*3x4*/) {
  double ca[3] = {0, 0, 0}, cb[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i)
    for (int d = 0; d < 3; ++d) { ca[d] += A[i * 3 + d]; cb[d] += B[i * 3 + d]; }
  for (int d = 0; d < 3; ++d) { ca[d] /= n; cb[d] /= n; }
  double H[9] = {0};
  for (int i = 0; i < n; ++i)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c)
        H[r * 3 + c] += ((double)A[i * 3 + r] - ca[r]) * ((double)B[i * 3 + c] - cb[c]);
  double U[9], S[3], V[9];
  jacobi_svd3(H, U, S, V);
  /* R = Vt^T U^T = V U^T */
  double R[9];
  for (int pass = 0; pass < 2; ++pass) {
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += V[r * 3 + k] * U[c * 3 + k];
        R[r * 3 + c] = s;
      }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) +
                       R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (pass == 1 || det >= 0) break;
    for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2]; /* Vt[m-1,:] *= -1 (:71) */
  }
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) T[r * 4 + c] = R[r * 3 + c];
    T[r * 4 + 3] = cb[r] - (R[r * 3 + 0] * ca[0] + R[r * 3 + 1] * ca[1] + R[r * 3 + 2] * ca[2]);
  }
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int t) {
#ifdef _OPENMP
  omp_set_num_threads(t);
#else
  (void)t;
#endif
}
