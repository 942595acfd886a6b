/*
 * pvn3d_hip.h -- C ABI of libpvn3d_hip.so: MI355X (gfx950) kernels for PVN3D's per-point
 * voting hot path.  This is the drop-in boundary: plain pointers and sizes, no torch types.
 *
 * Conventions (all entry points)
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - the library never allocates or frees device memory: outputs and scratch are passed in;
 *   - `stream` is a hipStream_t (passed as void*; NULL = the default stream).  Work is enqueued
 *     on it and the call returns without synchronising unless documented otherwise;
 *   - return value: 0 on success, otherwise a hipError_t code (launch / argument error).
 *     Nothing prints or exits (the reference's CUDA_CHECK_ERRORS, cuda_utils.h:30-39, calls
 *     exit(-1); the Python shim turns a non-zero return into RuntimeError instead);
 *   - re-entrant: no global mutable state; the device is the one current on the calling thread.
 *   - fp32 data, int32 indices, contiguous row-major tensors exactly as the reference lays
 *     them out (shapes are given per function).
 *
 * Section 1 mirrors, one to one, the `*_kernel_wrapper` functions that the reference's
 * pybind glue calls (pvn3d/_ext-src/src/{sampling,ball_query,group_points,interpolate}.cpp);
 * argument order is the reference's with `stream` appended.  Sections 2-3 are fused entry
 * points for the callers one level up (QueryAndGroup.forward, cal_frame_poses*).
 */
#ifndef PVN3D_HIP_H_
#define PVN3D_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PVN3D_ABI_VERSION 1
int pvn3d_abi_version(void);

/* =============================== 1. pointnet2 _ext ops ================================== */

/* opt_n_threads: pvn3d/_ext-src/include/cuda_utils.h:15-19.  Host helper; it fixes the FPS
 * tie-break order (see pvn3d_furthest_point_sampling). */
int pvn3d_opt_n_threads(int work_size);

/* replaces furthest_point_sampling_kernel_wrapper, pvn3d/_ext-src/src/sampling_gpu.cu:175-229
 * (declared sampling.cpp:12-14).  dataset (b,n,3) -> idxs (b,m).  `temp` (b,n) is the
 * reference's scratch; it may be NULL (distances live in registers here) -- if given it is
 * left untouched.  Index-exact with the reference block of opt_n_threads(n) threads,
 * including its tie-break order and the `x^2+y^2+z^2 <= 1e-3` skip rule. */
int pvn3d_furthest_point_sampling(int b, int n, int m, const float* dataset, float* temp,
                                  int* idxs, void* stream);

/* Nested sampling for PointNet++ pyramids (no reference counterpart; the reference runs
 * furthest_point_sampling_kernel_wrapper once per level, pointnet2_modules.py:47-55).  Level l+1 samples
 * the points level l selected, in the order it selected them; FPS is greedy, so that run returns
 * 0, 1, ..., m-1 unless a tie is broken differently under the next level's block shape.
 *   pvn3d_furthest_point_sampling_nested: as pvn3d_furthest_point_sampling, plus
 *     dmax_out (b,m) or NULL: the winning squared distance of every round (fp32 bit pattern; < 0: none);
 *     nest_flags (b,3) or NULL with nest_level in [0,3): R = nest_flags[cloud][nest_level] says that the
 *     cloud is in FPS order for its first R picks; rounds below R are not run (R >= m: idxs = 0..m-1 without
 *     a single round).  Index-exact either way.
 *   pvn3d_fps_nest_verify: ordered_xyz (b,n0,3) = the cloud gathered in the order of a run whose dmax is
 *     given; m_levels[l] (host array) = samples of the l-th following level (its cloud = the first
 *     m_levels[l-1], or n0, points).  flags[cloud][l] = the first round of that level's run that does NOT
 *     select its own number (a tie broken differently under that level's block shape, or a degenerate
 *     round); >= m_levels[l] if there is none; 1 if an earlier level already differs. */
int pvn3d_furthest_point_sampling_nested(int b, int n, int m, const float* dataset, float* temp, int* idxs,
                                         int* dmax_out, const int* nest_flags, int nest_level, void* stream);
int pvn3d_fps_nest_verify(int b, int n0, int n_levels, const int* m_levels, const float* ordered_xyz,
                          const int* dmax, int* flags, void* stream);

/* The same sampling with a caller-provided workspace (no reference counterpart: the reference's `temp` is
 * (b,n) floats and too small).  pvn3d_fps_ws_words(n) = 4-byte words per cloud that `ws` must hold (0: none
 * needed, ws may be NULL).  For 4096 < n <= 12288 the workspace lets the run use the spatially culled kernel
 * (csrc/fps_cells.hip: 64 equal-count cells per cloud, a round only recomputes the cells whose bounding box
 * is closer to the new sample than the cell's largest running distance -- index-exact, same tie order);
 * dmax_out / nest_flags / nest_level as in pvn3d_furthest_point_sampling_nested (a run with nest_flags uses
 * the register-resident kernel). */
int pvn3d_fps_ws_words(int n);
int pvn3d_furthest_point_sampling_ws(int b, int n, int m, const float* dataset, void* ws, int* idxs,
                                     int* dmax_out, const int* nest_flags, int nest_level, void* stream);
/* The same call with the number of waves that share a cloud's sampling rounds as a per-call argument (round 6).  The
 * reference gives a cloud a 512-thread block (sampling_gpu.cu:175-185).  waves_per_cloud: 0 or 1 = one wave per cloud
 * (what pvn3d_furthest_point_sampling_ws runs); >= 2, for 4096 < n <= 12288 = one wave per 64-point slot of the culled
 * kernel's 64 cells (2 waves up to 8192 points, 3 above): every wave keeps the cells' cached maxima, updates its own slot of
 * the touched cells and exchanges one (max, arg-max) entry per refreshed cell through LDS sequence words, no barrier.
 * Indices are identical whatever the value; the multi-wave form measured 1.45 x SLOWER (csrc/fps_cells.hip) and
 * is kept as an independently written cross-check. */
int pvn3d_furthest_point_sampling_ws_waves(int b, int n, int m, const float* dataset, void* ws, int* idxs,
                                           int* dmax_out, const int* nest_flags, int nest_level, int waves_per_cloud,
                                           void* stream);

/* replaces gather_points_kernel_wrapper, sampling_gpu.cu:22-29.
 * points (b,c,n), idx (b,npoints) -> out (b,c,npoints) */
int pvn3d_gather_points(int b, int c, int n, int npoints, const float* points, const int* idx,
                        float* out, void* stream);

/* replaces gather_points_grad_kernel_wrapper, sampling_gpu.cu:49-57.
 * grad_out (b,c,npoints), idx (b,npoints) -> grad_points (b,c,n); zero-fills grad_points
 * itself (the reference relies on torch::zeros, sampling.cpp:54-56). */
int pvn3d_gather_points_grad(int b, int c, int n, int npoints, const float* grad_out,
                             const int* idx, float* grad_points, void* stream);

/* replaces query_ball_point_kernel_wrapper, ball_query_gpu.cu:46-53.
 * new_xyz (b,m,3), xyz (b,n,3) -> idx (b,m,nsample); writes every slot (rows without a hit
 * are zero, as with the reference's torch::zeros, ball_query.cpp:19-21). */
int pvn3d_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz,
                     const float* xyz, int* idx, void* stream);

/* replaces group_points_kernel_wrapper, group_points_gpu.cu:30-38.
 * points (b,c,n), idx (b,npoints,nsample) -> out (b,c,npoints,nsample) */
int pvn3d_group_points(int b, int c, int n, int npoints, int nsample, const float* points,
                       const int* idx, float* out, void* stream);

/* replaces group_points_grad_kernel_wrapper, group_points_gpu.cu:66-75.
 * grad_out (b,c,npoints,nsample) -> grad_points (b,c,n); zero-fills grad_points itself. */
int pvn3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                            const float* grad_out, const int* idx, float* grad_points,
                            void* stream);

/* replaces three_nn_kernel_wrapper, interpolate_gpu.cu:61-68.
 * unknown (b,n,3), known (b,m,3) -> dist2 (b,n,3) squared distances, idx (b,n,3) */
int pvn3d_three_nn(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                   int* idx, void* stream);

/* Inverse-distance interpolation weights of PointnetFPModule.forward (pointnet2_modules.py:184-186) from three_nn's
 * dist2 (rows, 3): dist = sqrt(dist2); r = 1 / (dist + 1e-8); weight = r / (r0 + r1 + r2), fp32 in the reference's
 * operation order (one launch instead of five elementwise kernels). */
int pvn3d_three_nn_weights(long long rows, const float* dist2, float* weight, void* stream);

/* Same output as pvn3d_three_nn, bit for bit, through a uniform grid over the known points
 * (csrc/three_nn_grid.hip): 64 <= m <= 2048; workspace >= pvn3d_three_nn_grid_workspace_bytes(b, m)
 * bytes of device memory (bucket table + bucket-ordered copy of `known`).  ~30x fewer distance
 * evaluations when `known` samples a surface evenly (a furthest-point sample); degenerates to the
 * brute-force scan per query where the 27 neighbour cells do not provably contain the answer. */
size_t pvn3d_three_nn_grid_workspace_bytes(int b, int m);
int pvn3d_three_nn_grid(int b, int n, int m, const float* unknown, const float* known, float* dist2,
                        int* idx, void* workspace, size_t workspace_bytes, void* stream);

/* replaces three_interpolate_kernel_wrapper, interpolate_gpu.cu:103-111.
 * points (b,c,m), idx (b,n,3), weight (b,n,3) -> out (b,c,n) */
int pvn3d_three_interpolate(int b, int c, int m, int n, const float* points, const int* idx,
                            const float* weight, float* out, void* stream);

/* replaces three_interpolate_grad_kernel_wrapper, interpolate_gpu.cu:145-154.
 * grad_out (b,c,n), idx/weight (b,n,3) -> grad_points (b,c,m); zero-fills grad_points.
 * refbug_compat != 0 reproduces what the reference binary actually returns: its
 * interpolate.cpp:89-93 calls the FORWARD wrapper with m and n swapped. */
int pvn3d_three_interpolate_grad(int b, int c, int n, int m, const float* grad_out,
                                 const int* idx, const float* weight, float* grad_points,
                                 int refbug_compat, void* stream);

/* ============================ 2. fused set-abstraction ops ============================== */

/* Two ball queries sharing centres and points in one scan (the two scales of a
 * PointnetSAModuleMSG level, pvn3d/lib/pointnet2_utils/pointnet2_modules.py:57-60).
 * idx0 (b,m,nsample0) for radius0, idx1 (b,m,nsample1) for radius1; each identical to a
 * separate pvn3d_ball_query. */
int pvn3d_ball_query_pair(int b, int n, int m, float radius0, int nsample0, float radius1,
                          int nsample1, const float* new_xyz, const float* xyz, int* idx0,
                          int* idx1, void* stream);

/* Same outputs as pvn3d_ball_query_pair (bit-identical), computed through a toroidal uniform
 * grid so that only points in the 27 cells around a centre are distance-tested -- for large
 * clouds (PVN3D level 0: n = 12288) this examines ~100 candidates per centre instead of n.
 * nsample1 == 0 runs a single query (idx1 may be NULL).  n <= 32768.
 * workspace: >= pvn3d_ball_query_grid_workspace_bytes(b, n) bytes of device scratch. */
size_t pvn3d_ball_query_grid_workspace_bytes(int b, int n);
int pvn3d_ball_query_pair_grid(int b, int n, int m, float radius0, int nsample0, float radius1,
                               int nsample1, const float* new_xyz, const float* xyz, int* idx0,
                               int* idx1, void* workspace, size_t workspace_bytes, void* stream);

/* QueryAndGroup.forward (pvn3d/lib/pointnet2_utils/pointnet2_utils.py:293-330) minus the
 * ball query: writes the concatenated tensor directly.
 *   out (b, 3*use_xyz + c, m, nsample):
 *     channels [0,3)   = xyz[idx] - new_xyz        (grouping_operation(xyz^T) ; -= ; :313-314)
 *     channels [3,3+c) = features[idx]             (:317, torch.cat :319-321)
 * xyz (b,n,3), new_xyz (b,m,3), features (b,c,n) or NULL (c = 0), idx (b,m,nsample). */
int pvn3d_group_xyz_features(int b, int n, int m, int c, int nsample, int use_xyz,
                             const float* xyz, const float* new_xyz, const float* features,
                             const int* idx, float* out, void* stream);
/* Both scales of a multi-scale-grouping level in one launch (pointnet2_modules.py:47-71 runs QueryAndGroup once per
 * radius on the same xyz / new_xyz / features): every staged row group serves both index lists.  use_xyz = 1 and
 * features != NULL; out_s: (b, 3 + c, m, nsample_s).  Same values as two pvn3d_group_xyz_features calls. */
int pvn3d_group_xyz_features_pair(int b, int n, int m, int c, int nsample0, int nsample1, const float* xyz,
                                  const float* new_xyz, const float* features, const int* idx0, const int* idx1,
                                  float* out0, float* out1, void* stream);

/* Fused set abstraction for inference: gather (QueryAndGroup, pointnet2_utils.py:311-321) ->
 * SharedMLP = [1x1 conv -> BatchNorm (eval) -> ReLU] x n_layers (pytorch_utils.py:25-50) ->
 * max over nsample (pointnet2_modules.py:63-66), on fp32 MFMA; the (b, 3+c, m, nsample)
 * grouped tensor is never written.
 * Feature tensors are POINT-MAJOR at this boundary (row gathers are contiguous):
 *   features_pm  (b, n, ld_feat) floats, channels [0,c) of each row used, or NULL (c = 0);
 *                the reference's (b,c,n) tensor transposed (see pvn3d_transpose_bcn_to_bnc);
 *   out_pm       (b, m, ld_out): the dims[n_layers] pooled channels of centre j are written to
 *                out_pm[(b*m + j)*ld_out + out_coff ...] (the two scales of an MSG level share
 *                one buffer, which replaces torch.cat, pointnet2_modules.py:71).
 * dims_host: HOST int[n_layers+1], dims[0] = c + 3*use_xyz.  Layer-0 input channel order is
 * [c feature channels][x,y,z relative to the centre] -- i.e. the reference's conv weight
 * columns (xyz first, pointnet2_utils.py:319-321) rotated left by 3.  For layer l (K = dims[l],
 * M = dims[l+1]) with BatchNorm folded in (W' = W*g/sqrt(var+eps), b' = beta - mean*g/sqrt(var+eps)):
 *   w_packed[l]   DEVICE float[ceil(K/4)][ceil(M/32)][64][2], entry (k4, mt, lane, j) =
 *                 W'[mt*32 + (lane&31)][4*k4 + 2*j + (lane>>5)]  (0 outside M x K)
 *   bias_padded[l] DEVICE float[ceil(M/32)*32], zero padded.
 * w_packed / bias_padded are HOST arrays of device pointers.  nsample: power of two <= 64;
 * every dims[l>=1] <= 512.  The 16-byte row gather is used when features_pm is 16-byte aligned
 * and ld_feat % 4 == 0 (otherwise a scalar gather). */
int pvn3d_sa_mlp_maxpool(int b, int n, int m, int c, int nsample, int use_xyz, const float* xyz,
                         const float* new_xyz, const float* features_pm, int ld_feat,
                         const int* idx, int n_layers, const int* dims_host,
                         const float* const* w_packed, const float* const* bias_padded,
                         float* out_pm, int ld_out, int out_coff, void* stream);

/* Fused feature propagation for inference (PointnetFPModule.forward,
 * pointnet2_modules.py:183-206): three_interpolate(known_feats, idx, weight) ++ unknow_feats ->
 * SharedMLP.  known_pm (b, m, ld_known) and unknown_pm (b, n, ld_unknown) (or NULL, c1 = 0) are
 * point-major, channels [0,c2) / [0,c1) used; idx/weight (b,n,3); dims[0] = c2 + c1 in the
 * reference's order (interpolated first, :194-197); weights as in pvn3d_sa_mlp_maxpool.
 * out: out_point_major ? (b, n, ld_out) : (b, dims[n_layers], n)  (the module's API layout). */
int pvn3d_fp_interp_mlp(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                        const float* unknown_pm, int ld_unknown, const int* idx,
                        const float* weight, int n_layers, const int* dims_host,
                        const float* const* w_packed, const float* const* bias_padded, float* out,
                        int out_point_major, int ld_out, void* stream);

/* The same two fused chains with the fp32 contraction carried by the bf16 matrix pipe (csrc/sa_mlp_split.hip): every
 * fp32 operand is the exact sum of three bf16 pieces and a product is formed as the six partial products
 * w_i.x_j (i + j <= 4) accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- the three dropped terms are below 2^-24 of
 * the product, fp32's own rounding step, so the results carry fp32 accuracy at 6/16 of the fp32-MFMA cost.
 * Arguments as pvn3d_sa_mlp_maxpool (use_xyz = 1, features required) / pvn3d_fp_interp_mlp, except
 *   w_split[l]  DEVICE int16[ceil(K/16)][ceil(M/32)][3][64][8]: entry (slab, mt, piece, lane, j) = piece `piece` (bf16
 *               bits; hi = bf16(W'), mid = bf16(W' - hi), lo = bf16(W' - hi - mid), round to nearest) of
 *               W'[mt*32 + (lane & 31)][16*slab + 8*(lane >> 5) + j], 0 outside M x K.
 * Only the chain shapes the kernels are instantiated for are taken -- ask pvn3d_mlp_split_ok (1 = supported) with
 * c_a = channels of the first row source (SA: c, FP: c2), c_b = FP skip channels c1 (SA: 0); row tables must be
 * 16-byte aligned with ld % 4 == 0.  Unsupported shapes return hipErrorInvalidValue. */
int pvn3d_mlp_split_ok(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host);
int pvn3d_sa_mlp_maxpool_split(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                               const float* features_pm, int ld_feat, const int* idx, int n_layers,
                               const int* dims_host, const void* const* w_split, const float* const* bias_padded,
                               float* out_pm, int ld_out, int out_coff, void* stream);
int pvn3d_fp_interp_mlp_split(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                              const float* unknown_pm, int ld_unknown, const int* idx, const float* weight,
                              int n_layers, const int* dims_host, const void* const* w_split,
                              const float* const* bias_padded, float* out, int out_point_major, int ld_out,
                              void* stream);

/* The two fused chains on the fp16 matrix pipe with TWO pieces per fp32 operand (round 5; csrc/sa_mlp_split.hip, AR = 1):
 * x s = h + l, h = fp16(x s), l = fp16(x s - h), three partial products wh.xh + wh.xl + wl.xh per multiply accumulated
 * in fp32 by v_mfma_f32_32x32x16_f16 -- half the matrix-pipe time of the three-piece bf16 form.  A product of two fp16
 * numbers is exact in fp32; the pieces carry >= 20 bits of each operand and the dropped wl.xl is below 2^-20 of the
 * product; measured against fp64 (tools/mfma_fp16x2_bench.hip) the results are as close as the fp32 FMA chain's
 * (5.5e-7 vs 6.2e-7 of the output scale at K = 512): what remains is the fp32 accumulation's rounding, not the operands'.
 * fp16's 5 exponent bits are handled by exact power-of-two scales: the weights' (host, sw below), the layer-0 input's
 * from a bound on |input| that the kernel reads from DEVICE memory, the hidden layers' from the rigorous bound
 * B' = ||W||_inf B + max|b|; every scale is undone exactly where accumulators leave a layer.
 * Arguments as the _split entry points, except
 *   w_split2[l]   DEVICE int16[ceil(K/16)][ceil(M/32)][2][64][8]: piece 0 = fp16(sw_l W'), piece 1 = fp16(sw_l W' - piece 0)
 *                 (round to nearest), same (slab, mt, piece, lane, j) order as w_split;
 *   layer_meta    HOST float[3 * n_layers]: per layer sw_l (a power of two with max|sw_l W'| in [2^13, 2^14]),
 *                 ||W'||_inf (largest row sum of |W'|, true weights), max|bias|;
 *   *_absmax      DEVICE float: a bound on |x| over the row table it names (pvn3d_absmax); SA: xyz_absmax bounds the
 *                 coordinates of `xyz` (the relative coordinates are then within twice that); FP: unknown_absmax may be
 *                 NULL when c1 == 0.  The bounds must hold: a larger value costs nothing measurable, a smaller one
 *                 saturates operands at 65504 / scale.
 *   out_absmax    DEVICE float or NULL: atomic max of |output| over the launch (accumulates: set it to 0 before the first
 *                 launch that writes a table) -- the *_absmax of the level that consumes this output, without a pass of
 *                 its own.
 * pvn3d_mlp_split2_ok answers pvn3d_mlp_split_ok's question for these kernels (smaller LDS footprint; one more chain
 * shape: SA level 1).  pvn3d_absmax: *out_max = max(*out_max, max |src[r][ch]|, r < rows, ch < c) as an atomic max on
 * the bit pattern -- set *out_max to 0 (or to a previous bound) before the call.
 * Narrow set-abstraction chains (three layers of <= 128 channels, nsample 16 / 32; the shapes of the backbone's SA
 * levels 0-1: 9 -> 16 -> 16 -> 32, 9 -> 32 -> 32 -> 64, 99 -> 64 -> 64 | 96 -> 128) run a kernel of their own behind
 * pvn3d_sa_mlp_maxpool_split2 (weights of the whole chain resident in LDS, one wave per 32 columns through all layers);
 * for c == 6 it reads the six-float feature rows in place, so features_pm / ld_feat need no 16-byte alignment there.
 * Round 6, two more arguments on every *_split2 entry point:
 *   out_row_mul   DEVICE float[ceil(M_last / 32) * 32] or NULL: per-output-channel multiplier applied to the results (after
 *                 the ReLU, together with the chain's own power-of-two scale).  It lets the caller scale every ROW of a
 *                 weight matrix into fp16's range on its own: the host side (PackedMLP.split2) rescales the network
 *                 diagonally -- W~_l = D_l W_l D_(l-1)^-1, b~_l = D_l b_l with power-of-two diagonals chosen so that every
 *                 hidden channel's bound is ~1 -- which is exact, keeps rows whose BatchNorm scale is orders of magnitude
 *                 apart at full two-piece precision, and leaves D_L^-1 to be undone here.  out_absmax is taken on the
 *                 multiplied values.
 *   flags         0 or PVN3D_MLP_NO_NARROW: a PER-CALL switch that keeps the chain off the narrow-chain kernels (A/B
 *                 measurements: SA level 1 then runs the 4 + 4 wave kernel, SA level 0 is refused by pvn3d_mlp_split2_ok
 *                 and stays on pvn3d_sa_mlp_maxpool).  The library has no process-wide switch (no global mutable state). */
#define PVN3D_MLP_NO_NARROW 1
/* The caller's promise that layer 0's weight slabs over table A (SA: the gathered feature table, FP: the interpolated known
 * table) are an identity block, i.e. that their low fp16 pieces are zero -- what a pre-contracted chain is ([I | Wr]:
 * lib/pointnet2_utils/_fused_mlp.py::precontracted), scaled by the layer's power-of-two weight scale: the 4 + 4-wave kernel
 * then multiplies a row tile only with the chunk of the table that carries its unit block, with constant weight fragments
 * and without the zero low piece (FP level 1: 64 instead of 768 MFMAs per column block).  Results are the same bits with
 * and without the flag when the promise holds. */
#define PVN3D_MLP_IDENTITY_A 2
int pvn3d_mlp_split2_ok(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host, int flags);
/* The kernel family the fp16 x 2 entry points would run the chain on: 0 none, 1 the 4 + 4-wave kernel, 2 a narrow-chain
 * kernel (weights resident in LDS; instantiated for the backbone's widths: SA chains 6 / 96 features -> (16, 16, 32),
 * (32, 32, 64), (64, 64, 128), (64, 96, 128) at nsample 16 / 32; the pre-contracted FP chain 128 + 6 -> 128 -> 128 with a
 * channel-major output).  A diagnostic for hosts that build other networks (round 6: the review's "a backbone with one
 * different width silently loses 1.8 ms"). */
int pvn3d_mlp_split2_kernel(int is_sa, int c_a, int c_b, int nsample, int n_layers, const int* dims_host,
                            int out_point_major, int flags);
/* The FP chain in its pre-contracted form: the caller promises that the first c2 columns of layer 0's weights are the
 * identity (known_pm holds the first conv's interpolated half, already applied per KNOWN point: what
 * _ext.fp_interp_mlp does for FP level 0), i.e. layer 0 = relu(interp(known) + Wb.skip + b0).  Arguments and results as
 * pvn3d_fp_interp_mlp_split2, which computes the same thing by multiplying with that identity and is what this entry
 * point falls back to; for c2 = 128, c1 = 6, 128 -> 128, channel-major output (FP level 0 of the backbone) the
 * narrow-chain kernel adds the interpolated rows to the accumulators instead (flags & PVN3D_MLP_NO_NARROW: not). */
int pvn3d_fp_interp_add_mlp_split2(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                                   const float* unknown_pm, int ld_unknown, const int* idx, const float* weight,
                                   int n_layers, const int* dims_host, const void* const* w_split2,
                                   const float* const* bias_padded, const float* layer_meta,
                                   const float* known_absmax, const float* unknown_absmax, float* out,
                                   int out_point_major, int ld_out, float* out_absmax, const float* out_row_mul, int flags,
                                   void* stream);
int pvn3d_sa_mlp_maxpool_split2(int b, int n, int m, int c, int nsample, const float* xyz, const float* new_xyz,
                                const float* features_pm, int ld_feat, const int* idx, int n_layers,
                                const int* dims_host, const void* const* w_split2, const float* const* bias_padded,
                                const float* layer_meta, const float* features_absmax, const float* xyz_absmax,
                                float* out_pm, int ld_out, int out_coff, float* out_absmax, const float* out_row_mul,
                                int flags, void* stream);
int pvn3d_fp_interp_mlp_split2(int b, int n, int m, int c2, int c1, const float* known_pm, int ld_known,
                               const float* unknown_pm, int ld_unknown, const int* idx, const float* weight,
                               int n_layers, const int* dims_host, const void* const* w_split2,
                               const float* const* bias_padded, const float* layer_meta, const float* known_absmax,
                               const float* unknown_absmax, float* out, int out_point_major, int ld_out,
                               float* out_absmax, const float* out_row_mul, int flags, void* stream);
int pvn3d_absmax(long long rows, int c, const float* src, int ld_src, float* out_max, void* stream);

/* Layer-by-layer split-bf16 SharedMLP for chains whose hidden layer is too wide for the fused kernel (FP levels 2-3 of
 * PVN3D's backbone, lib/pvn3d.py:114-118; the module code is pointnet2_modules.py:188-206) -- csrc/split_gemm.hip.
 * "s16" layout of a matrix [rows][K]: rows x slabs (= 16 k each) x 3 pieces x 16 bf16, i.e. entry (r, k, piece) at
 * byte ((r * slabs + k / 16) * 3 + piece) * 32 + 2 * (k % 16); piece 0 + piece 1 + piece 2 == the fp32 value
 * (activations: truncation split, exact; weights: round-to-nearest pieces, |rest| <= 2^-26 |w|).
 *   pvn3d_split_rows: fp32 rows [rows][ld_src] (channels [0, c)) -> s16 [rows][slabs], channels >= c zero.
 *   pvn3d_split_gemm: out[p][o] = act( sum_k W[o][k] X[p][k]  (+ sum_t weight[p][t] * z[frame(p) * z_rows_per_frame +
 *                     idx[p][t]][o], frame(p) = p / z_points_per_frame, when z != NULL: three_interpolate of an fp32
 *                     table, pointnet2_utils.py:136-170)  (+ bias_padded[o]) ), act = relu or identity;
 *       x_s16 [n_points][slabs], w_s16 [roundup128(n_out)][slabs] with zero rows beyond n_out; slabs even;
 *       results as fp32 rows out_f32 [n_points][ld_out] (channels < n_out) and / or as s16 rows out_s16
 *       [n_points][slabs_out] (every channel < 16 * slabs_out <= roundup128(n_out) is written; channels >= n_out hold
 *       act(z-term + bias pad) -- exact zeros when z's and bias's pad columns are zero).
 * The chain H = relu(Wb.skip + interp(Wa.known) + b1), Y = relu(W2.H + b2) equals the reference's
 * conv([interp(known); skip]) -> ... by linearity of the interpolation; the fp32 rounding sequence differs (1e-6). */
int pvn3d_split_rows(long long rows, int c, const float* src, int ld_src, void* dst_s16, int slabs, void* stream);
int pvn3d_split_gemm(int n_points, int n_out, int slabs, const void* x_s16, const void* w_s16,
                     const float* bias_padded, int relu, const float* z, int ldz, int z_points_per_frame,
                     int z_rows_per_frame, const int* idx, const float* weight, float* out_f32, int ld_out,
                     void* out_s16, int slabs_out, void* stream);

/* The layer-by-layer GEMM in the fp16 x 2 arithmetic of pvn3d_*_split2 (round 5).  "h16" layout of a matrix [rows][K]:
 * rows x slabs x 2 pieces x 16 fp16, entry (r, k, piece) at byte ((r * slabs + k / 16) * 2 + piece) * 32 + 2 * (k % 16),
 * holding s * x split as piece 0 = fp16(s x), piece 1 = fp16(s x - piece 0), where s is the power of two that puts a
 * bound B on |x| at 2^14 -- B is a DEVICE float the caller names (pvn3d_absmax of the source, a GEMM's out_absmax, or
 * pvn3d_bound_affine of such bounds); the same pointer must be passed where the matrix is written and where it is read.
 *   pvn3d_split_rows2: fp32 rows -> h16 (scale from *src_bound).
 *   pvn3d_split_gemm2: pvn3d_split_gemm on h16 operands; w_h16 holds w_scale * W (w_scale a power of two, host);
 *       w_row_mul (round 6; DEVICE float[ceil(n_out / 128) * 128] or NULL): per-output-channel multiplier of the
 *       accumulators, applied with 1 / (w_scale * x scale) BEFORE the interpolated rows and the bias are added -- so that
 *       every ROW of W can carry its own power-of-two scale (w_h16 row o = w_scale * rs[o] * W[o], w_row_mul[o] = 1 / rs[o])
 *       and a row whose BatchNorm scale is far below the matrix maximum keeps both fp16 pieces normal;
 *       out_absmax (optional): atomic max of |out_f32| -- set it to 0 before; out_bound: the bound the h16 output is
 *       written with (required with out_h16).  Three partial products per multiply on v_mfma_f32_32x32x16_f16.
 *   pvn3d_bound_affine: *out = 1.01 (ca * *a + cb * *b + c0) (b may be NULL): the rigorous bound |W x + ...| <=
 *       ||W||_inf max|x| + ... of a layer's output from the bounds of its inputs. */
int pvn3d_split_rows2(long long rows, int c, const float* src, int ld_src, const float* src_bound, void* dst_h16, int slabs,
                      void* stream);
int pvn3d_split_gemm2(int n_points, int n_out, int slabs, const void* x_h16, const float* x_bound, const void* w_h16,
                      float w_scale, const float* w_row_mul, const float* bias_padded, int relu, const float* z, int ldz,
                      int z_points_per_frame, int z_rows_per_frame, const int* idx, const float* weight, float* out_f32,
                      int ld_out, float* out_absmax, void* out_h16, int slabs_out, const float* out_bound, void* stream);
/* pvn3d_split_gemm2 runs the LDS-DMA kernel (round 6: operands global -> LDS directly, three 16-k stages in a ring, tiles
 * of 128 channels x 256 points when those still fill the chip, else x 128 points); pvn3d_split_gemm2_tile128 is the same
 * product on the round-5 kernel (register-staged 128 x 128 tiles): the MFMA order per accumulator is the same, the results
 * are bit-identical -- the cross-check of the DMA ordering (tests/test_gpu_ops.py). */
int pvn3d_split_gemm2_tile128(int n_points, int n_out, int slabs, const void* x_h16, const float* x_bound, const void* w_h16,
                              float w_scale, const float* w_row_mul, const float* bias_padded, int relu, const float* z,
                              int ldz, int z_points_per_frame, int z_rows_per_frame, const int* idx, const float* weight,
                              float* out_f32, int ld_out, float* out_absmax, void* out_h16, int slabs_out,
                              const float* out_bound, void* stream);
int pvn3d_bound_affine(float* out, const float* a, float ca, const float* b, float cb, float c0, void* stream);

/* (b, c, n) -> (b, n, ld_out) with out[(b*n + j)*ld_out + ch] = in[(b*c + ch)*n + j]. */
int pvn3d_transpose_bcn_to_bnc(int b, int c, int n, const float* in, float* out, int ld_out,
                               void* stream);

/* ========================= 3. vote -> MeanShift -> pose (post-proc) ===================== */

/* Batched MeanShiftTorch.fit (pvn3d/lib/utils/meanshift_pytorch.py:18-51).
 * n_seg independent fits.  pts is a (total,4) float array (x,y,z,unused) holding the
 * segments back to back; segment s occupies rows [seg_off[s], seg_off[s]+seg_cnt[s]).
 * seg_off / seg_cnt are DEVICE int arrays (counts may come from an on-device compaction);
 * max_cnt_host >= every seg_cnt is the host-known bound that sizes the grid.
 * Outputs per segment: ctr (n_seg,3); labels (total) uint8 aligned with pts rows;
 * iters (n_seg) number of mean-shift iterations run (== the reference's `it` under PVN3D_MS_NO_WINNER_STOP;
 * otherwise <= it, see "Winner stop" below).
 * A segment with cnt == 0 yields ctr = 0, iters = 0.
 * workspace: >= pvn3d_meanshift_workspace_bytes(n_seg, total, max_iter) bytes of device
 * scratch.  poll_host: optional PINNED host int[2] used to stop enqueuing once every fit
 * has converged (the call then blocks on events every `poll_every` iterations, like the
 * reference's per-iteration host test, meanshift_pytorch.py:42); with poll_host == NULL the
 * call is fully asynchronous: poll_every <= 0 enqueues all max_iter+1 iterations (finished fits exit
 * at block start); poll_every = E > 0 enqueues at most E iterations -- a launch sequence of fixed
 * length, capturable in a HIP graph -- and a fit that would still run after them reports
 * iters[s] = -(iterations run): its centre is not final and the caller has to repeat the call
 * with a poll buffer (typical vote sets converge in 4-6 iterations).
 * flags: PVN3D_MS_ALIGNED32 -- the caller guarantees seg_off[s] % 32 == 0 and that rows
 * [seg_off[s], seg_off[s] + roundup32(seg_cnt[s])) belong to segment s (pvn3d_vote_compact's
 * layout does; informational).  The iteration kernel keeps two seeds per lane (packed fp32 math)
 * when max_cnt_host >= 256 and one otherwise, and splits a fit's points over the four waves of a
 * workgroup (always, unless PVN3D_MS_FORCE_WHOLE); all variants give identical bits, and the
 * PVN3D_MS_FORCE_* flags pin the choice (tests, A/B timing).
 * Exact early-out: from iteration 5 on, seeds whose update returned their own position bit for bit
 * (fixed points of the iteration function: shift 0 in every later iteration) are dropped from the
 * iterated set every fourth iteration -- same centres, labels and iteration counts, bit for bit;
 * PVN3D_MS_NO_EARLY_OUT iterates every seed every time (tests, A/B timing).
 * Winner stop (exact): the output is C[max_idx] only (meanshift_pytorch.py:46-51); max_idx depends on the original
 * points alone and a seed's trajectory on no other seed.  Once the update of seed max_idx returns its own position
 * bit for bit (a fixed point of the iteration function) the fit's centre is known; the remaining iterations, which
 * only wait for slower seeds to pass the reference's stop test, are not run.  PVN3D_MS_NO_WINNER_STOP runs them
 * (iters == the reference's `it`; tests); the centre is taken from the same record, so both modes return identical
 * bits.  Fits whose winner never reaches a bitwise fixed point stop by the reference's rule. */
#define PVN3D_MS_ALIGNED32 1
#define PVN3D_MS_NO_EARLY_OUT 2
#define PVN3D_MS_FORCE_SCALAR 4
#define PVN3D_MS_FORCE_PACKED 8
#define PVN3D_MS_FORCE_WHOLE 16
#define PVN3D_MS_FORCE_SPLIT 32
/* LDS-free iteration kernel (needs PVN3D_MS_ALIGNED32): points are wave-uniform SGPR operands streamed with scalar
 * loads, one wave per tile of seeds, no barrier -- same bits as the other variants.  It is the form to run BESIDE the
 * fused-MLP kernels (which keep the LDS pipe busy); PVN3D_MS_WAVE_CAP(n) bounds the launch to n waves that stride
 * over the work, i.e. chooses how many SIMD wave slots the iterations occupy (0 = one wave per tile). */
#define PVN3D_MS_SGPR_POINTS 64
#define PVN3D_MS_NO_WINNER_STOP 128
#define PVN3D_MS_WAVE_CAP(n) (((n) & 0xfffff) << 8)
/* The pruned neighbour count of the original points (the arg-max that names the winning seed) runs as ONE launch that
 * also sums the (core row, non-core column) hits per column (round 6); PVN3D_MS_COUNT_TWO_PASS selects the two launches
 * of rounds 2-5 (rows x non-core columns, then non-core rows x core columns: the same tests twice) -- identical counts,
 * kept as the cross-check (tests). */
#define PVN3D_MS_COUNT_TWO_PASS (1 << 28)
size_t pvn3d_meanshift_workspace_bytes(int n_seg, int total, int max_iter);
int pvn3d_meanshift_fit_batch(const float* pts, const int* seg_off, const int* seg_cnt,
                              int n_seg, int total, int max_cnt_host, float bandwidth,
                              int max_iter, float* ctr, uint8_t* labels, int* iters,
                              void* workspace, size_t workspace_bytes, int* poll_host,
                              int poll_every, int flags, void* stream);

/* Vote assembly + order-preserving mask compaction
 * (cal_frame_poses_lm, pvn3d/lib/utils/pvn3d_eval_utils.py:160-175; cal_frame_poses :41-42,83,91-92).
 * For instance i = (frame inst_frame[i], class inst_cls[i]) the rows p with
 * mask[frame, p] == cls are taken in ascending p ("row r" = r-th such point).  If sel != NULL,
 * row r is kept only when sel[i*sel_inst_stride + r] != 0 -- sel is typically the `labels`
 * output of an earlier centre fit on the same instance (in_pred_kp = cls_voted_kps[:, ctr_labels, :],
 * :91-92); when no row is selected row 0 is kept (`ctr_labels[0] = 1`, :86-87, 178-179).
 * For v in [v_first, v_first+v_count): vote v of row p = pcld[p] - off_v[p] with
 * off_v = pred_kp_of[frame, v] for v < n_kps and ctr_of[frame, 0] for v == n_kps.
 * pcld (F,n_pts,3); mask (F,n_pts) int32; ctr_of (F,1,n_pts,3); pred_kp_of (F,n_kps,n_pts,3).
 * Output: votes ((n_inst*(n_kps+1))*n_pts, 4) float -- segment s = i*(n_kps+1)+v starts at row
 * s*n_pts; seg_off[s] / seg_cnt[s] are filled for the produced segments only.
 * inst_frame / inst_cls: device int arrays (n_inst). */
int pvn3d_vote_compact(int n_frames, int n_pts, int n_kps, int n_inst, int v_first, int v_count,
                       const float* pcld, const int* mask, const float* ctr_of,
                       const float* pred_kp_of, const int* inst_frame, const int* inst_cls,
                       const uint8_t* sel, long long sel_inst_stride, float* votes,
                       int* seg_off, int* seg_cnt, void* stream);
/* The same with the rows per segment as an argument (round 6): segment s starts at row s * seg_stride_rows
 * (seg_stride_rows >= n_pts; a multiple of 32 keeps the PVN3D_MS_ALIGNED32 promise).  A stride of n_pts = 12288 rows puts
 * every segment 3 * 2^16 bytes after the previous one, and the iteration kernels' waves -- one fit each, walking their
 * points at the same pace -- then hit the same memory channels at the same time: 1.26 ms per iteration of the headline
 * batch against 1.00 ms with 12288 + 32 rows per segment (tools/ms_rate.py).  The host side (_vote_engine.seg_stride_rows)
 * adds 32 rows whenever the natural stride is a multiple of 1024 rows.  sel / sel_inst_stride: rows of the labels of an
 * earlier fit batch on the same layout, i.e. sel_inst_stride = (n_kps + 1) * seg_stride_rows. */
int pvn3d_vote_compact_strided(int n_frames, int n_pts, int seg_stride_rows, int n_kps, int n_inst, int v_first,
                               int v_count, const float* pcld, const int* mask, const float* ctr_of,
                               const float* pred_kp_of, const int* inst_frame, const int* inst_cls,
                               const uint8_t* sel, long long sel_inst_stride, float* votes, int* seg_off,
                               int* seg_cnt, void* stream);

/* Batched best_fit_transform (pvn3d/lib/utils/basic_utils.py:47-80): for each of n_sets,
 * A (npts,3) -> B (npts,3), T (3,4) float64 row-major [R|t].  valid (n_sets) int or NULL:
 * sets with valid == 0 get the identity pose (pvn3d_eval_utils.py:172-173). */
int pvn3d_best_fit_transform(int n_sets, int npts, const float* A, const float* B,
                             const int* valid, double* T, void* stream);

/* ADD / ADD-S pose distances of a batch of instances (Basic_Utils.cal_add_cuda / cal_adds_cuda,
 * pvn3d/lib/utils/basic_utils.py:617-635; called per object by eval_metric(_lm),
 * pvn3d_eval_utils.py:113-136, 204-221).  Instance i uses the mesh points
 * pts[pts_off[i] .. pts_off[i+1]) (pts (total,3) float, pts_off DEVICE int[n_inst+1]), its
 * predicted and ground-truth poses pred_RT / gt_RT (n_inst,3,4) float row-major [R|t]:
 *   add[i]  = mean_k |pred(x_k) - gt(x_k)|,  adds[i] = mean_k min_j |pred(x_j) - gt(x_k)|.
 * max_pts >= every instance's point count (host-known bound, sizes the grid);
 * workspace >= pvn3d_add_adds_workspace_bytes(n_inst, max_pts) bytes.  Deterministic. */
size_t pvn3d_add_adds_workspace_bytes(int n_inst, int max_pts);
int pvn3d_add_adds_batch(int n_inst, int max_pts, const float* pts, const int* pts_off,
                         const float* pred_RT, const float* gt_RT, void* workspace,
                         size_t workspace_bytes, float* add_out, float* adds_out, void* stream);

/* YCB centre-cluster re-labelling of cal_frame_poses (pvn3d_eval_utils.py:58-72), batched.
 * pcld, ctr_of (n_frames,n_pts,3) [ctr_of = the first (only) centre offset row]; mask
 * (n_frames,n_pts) int32; ctrs (n_frames,n_cls_m1,3) = MeanShift centre of class id c+1;
 * present (n_frames,n_cls_m1) int32 = class occurs in mask; thr (n_cls_m1) = fp32(0.8*ycb_r_lst).
 * new_mask (n_frames,n_pts): a labelled point takes the class of its nearest present centre
 * (first minimum in ascending class id) when that distance < thr[class]; present_new
 * (n_frames,n_cls_m1) int32 = classes occurring in new_mask. */
int pvn3d_relabel_by_centre(int n_frames, int n_pts, int n_cls_m1, const float* pcld,
                            const float* ctr_of, const int* mask, const float* ctrs,
                            const int* present, const float* thr, int* new_mask,
                            int* present_new, void* stream);

/* Vote (keypoint / centre offset) L1 loss, of_l1_loss with normalize=True (pvn3d/lib/loss.py:45-73):
 * pred_ofsts (bs,n_kpts,n_pts,3), kp_targ_ofst (bs,n_pts,n_kpts,3), labels (bs,n_pts) float
 * (w = labels > 1e-8) -> loss (bs,n_kpts) = sum_{i,c} w|pred-targ| / (sum_i w + 1e-3), and
 * wsum (bs,n_kpts) = sum_i w (kept for the backward).  Deterministic summation order. */
int pvn3d_of_l1_loss(int bs, int n_kpts, int n_pts, const float* pred_ofsts,
                     const float* kp_targ_ofst, const float* labels, float* loss, float* wsum,
                     void* stream);
/* grad_pred (bs,n_kpts,n_pts,3) = grad_loss[b,k] * w_i * sign(pred - targ) / (wsum[b,k] + 1e-3). */
int pvn3d_of_l1_loss_grad(int bs, int n_kpts, int n_pts, const float* pred_ofsts,
                          const float* kp_targ_ofst, const float* labels, const float* wsum,
                          const float* grad_loss, float* grad_pred, void* stream);

/* Bit-reproducible forms of the backward scatters: same results as
 * pvn3d_group_points_grad / pvn3d_gather_points_grad (nsample = 1) / pvn3d_three_interpolate_grad
 * up to fp32 rounding, but independent of the order in which the contributions are added
 * (64-bit fixed-point accumulation with integer atomics, csrc/scatter_det.hip).
 * workspace >= pvn3d_scatter_det_workspace_bytes(b, c, n_targets) bytes (n_targets = n resp. m). */
size_t pvn3d_scatter_det_workspace_bytes(int b, int c, int n_targets);
int pvn3d_group_points_grad_det(int b, int c, int n, int npoints, int nsample,
                                const float* grad_out, const int* idx, float* grad_points,
                                void* workspace, size_t workspace_bytes, void* stream);
int pvn3d_three_interpolate_grad_det(int b, int c, int n, int m, const float* grad_out,
                                     const int* idx, const float* weight, float* grad_points,
                                     void* workspace, size_t workspace_bytes, void* stream);

/* ---- Layer-by-layer SharedMLP for small launches (csrc/small_batch.hip) -------------------------------------
 * The fused chains (pvn3d_sa_mlp_maxpool / pvn3d_fp_interp_mlp) give a workgroup 64 columns and the whole layer
 * chain; with one frame per call the deep levels are 8 - 64 workgroups.  Same arithmetic (fp32 MFMA, eval
 * BatchNorm folded into W', b'), one layer per launch, one wave per 32 x 32 output tile, activations as
 * point-major fp32 matrices [columns][channels].
 * pvn3d_sb_linear: C[M][ldc] = act(A[M][K] . W[N][K]^T + bias[N]) (relu != 0: max(.,0)).  splits > 1 cuts K into
 * that many slices (partials in part[splits][M][N], caller's scratch; added in ascending slice order, then bias and
 * ReLU: deterministic) -- pvn3d_sb_linear_splits(M, N, K) is the library's choice for a launch with few output tiles.
 * pvn3d_sb_gather_sa / _fp: the layer-0 input rows (as pvn3d_mt_gather_*, fp32).  pvn3d_sb_pool_max: max over
 * the ns rows of every group -> out[g*out_ld + c]. */
int pvn3d_sb_linear_splits(int M, int N, int K);
int pvn3d_sb_linear(int M, int N, int K, const float* A, int lda, const float* W, int ldw, const float* bias, int relu,
                    float* C, int ldc, float* part, int splits, void* stream);
int pvn3d_sb_gather_sa(int b, int n, int m, int ns, int C, int use_xyz, const float* xyz, const float* new_xyz,
                       const float* feat, long long fsb, long long fsc, long long fsn, const int* idx, float* X0, int ld,
                       void* stream);
int pvn3d_sb_gather_fp(int b, int n, int mk, int C2, int C1, const float* known, long long ksb, long long ksc,
                       long long ksn, const float* unknown, long long usb, long long usc, long long usn, const int* idx,
                       const float* w, float* X0, int ld, void* stream);
int pvn3d_sb_pool_max(long long G, int ns, int ld, int C, const float* H, float* out, long long out_ld, void* stream);

/* ---- Training-mode SharedMLP on bf16 MFMA (csrc/mlp_train.hip; BASELINE config 5) ---------------------------
 * Replaces, for a module in training mode, the reference's grouped (B,C,npoint,nsample) tensor -> [Conv2d 1x1 ->
 * BatchNorm2d(batch statistics) -> ReLU] x L -> max_pool2d (pointnet2_modules.py:58-71, 188-206;
 * pytorch_utils.py:25-50) and its autograd backward.  Activations are POINT-MAJOR bf16 matrices [rows][ld]
 * (row = one (cloud, centre, sample) column, ld = channels rounded up to 16, pad columns zero).  bf16 = the upper
 * half of the fp32 bit pattern, round-to-nearest-even. */
/* C[M][N] = A[M][K] . B[N][K]^T (A, B bf16, K contiguous and a multiple of 16; lda, ldb multiples of 8) on
 * v_mfma_f32_32x32x16_bf16, fp32 accumulate.  pvn3d_mt_gemm_nt: C bf16 [M][ldc] (columns N..ldc-1 written as
 * zero); stat_sum / stat_sq, if given, [pvn3d_mt_gemm_nt_stat_rows(M)][stat_ld] = partial per-column sums of c
 * and c^2 from the fp32 accumulators (BatchNorm statistics; summed by pvn3d_mt_bn_finalize).
 * pvn3d_mt_gemm_nt_splitk: C fp32 [M][ldc] += A . B^T with K split `ksplit` ways (fp32 atomics) -- the weight
 * gradient dW = dY^T . H with K = rows. */
int pvn3d_mt_gemm_nt(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                     float* stat_sum, float* stat_sq, int stat_ld, void* stream);
int pvn3d_mt_gemm_nt_stat_rows(int M);
int pvn3d_mt_gemm_nt_splitk(int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                            int ksplit, void* stream);
/* X bf16 [rows][ld] -> XT [ld][ldt]; W fp32 [rows][cols] (row stride lds) -> bf16 [out_rows][ld] zero-padded,
 * transposed when `transpose`; bf16 [B*R][ld] channels [c_off, c_off+C) <-> fp32 channel-major [B][C][R]. */
/* Weight gradient without transposed copies, for layers of at most 512 x 544 channels (pvn3d_mt_wgrad_tn_ok):
 * dW (M, N) fp32 += dY^T . H over `rows` rows of the row-major bf16 matrices dY [rows][ldy] and H [rows][ldh]. */
int pvn3d_mt_wgrad_tn_ok(int M, int N);
int pvn3d_mt_wgrad_tn(long long rows, int M, int N, const void* dY, int ldy, const void* H, int ldh, float* dW, int ldw,
                      void* stream);
int pvn3d_mt_transpose(long long rows, int ld, const void* X, void* XT, long long ldt, void* stream);
int pvn3d_mt_pack_weight(int rows, int cols, const float* W, int lds, int transpose, void* out, int out_rows, int ld,
                         void* stream);
int pvn3d_mt_unpack_cm(int b, int R, int ld, int c_off, int C, const void* X, float* out, void* stream);
int pvn3d_mt_pack_cm(int b, int R, int ld, int C, const float* in, void* X, void* stream);
/* Backward of the layer-0 gathers without atomics on the data (the reference scatters with atomicAdd,
 * group_points_gpu.cu:49-75, interpolate_gpu.cu:137-170).  pvn3d_mt_csr_build inverts an index list once per call:
 * idx [b][E] with values in [0, n_src) (n_src <= 32768) -> start [b][n_src + 1], ent [b][E] (entry ids grouped by the
 * row they reference).  pvn3d_mt_inv_gather: out[(b * n_src + p) * out_ld + c] (+)= sum over the entries e of row p of
 * w[b][e] * dX[(b * E/div + e/div) * ld + c_off + c] for c < C <= 512 (bf16 dX, fp32 point-major out; wider tensors
 * in channel blocks; w may be NULL = 1; div = 1 for set abstraction, 3 for three_interpolate; accumulate != 0 adds). */
int pvn3d_mt_csr_build(int b, int n_src, int E, const int* idx, int* start, int* ent, void* stream);
int pvn3d_mt_inv_gather(int b, int n_src, int E, int div, int C, int c_off, int ld, const void* dX, const int* start,
                        const int* ent, const float* w, float* out, int out_ld, int accumulate, void* stream);
/* Layer-0 inputs.  SA: X0[(b*m+j)*ns+s][c] = relative xyz (c < 3 when use_xyz) ++ feat[b, c, idx[b,j,s]]
 * (QueryAndGroup, pointnet2_utils.py:293-330); FP: X0[b*n+i][c] = three_interpolate(known)[c < C2] ++ unknown
 * (pointnet2_modules.py:188-203).  feat / known / unknown: fp32, element (b,c,n) at base + b*sb + c*sc + n*sn. */
int pvn3d_mt_gather_sa(int b, int n, int m, int ns, int C, int use_xyz, const float* xyz, const float* new_xyz,
                       const float* feat, long long fsb, long long fsc, long long fsn, const int* idx, void* X0, int ld,
                       void* stream);
int pvn3d_mt_gather_fp(int b, int n, int mk, int C2, int C1, const float* known, long long ksb, long long ksc,
                       long long ksn, const float* unknown, long long usb, long long usc, long long usn, const int* idx,
                       const float* w, void* X0, int ld, void* stream);
/* BatchNorm2d in training mode.  finalize: partial sums [P][ld] -> mean, 1/std, a = gamma/std, b = beta - mean*a
 * (zero in pad channels), running_mean / running_var updated with `momentum` (unbiased variance) when given.
 * relu_apply: H = relu(a y + b).  pool_max: max over the ns rows of every group -> out[g*out_ld + c] fp32 and the
 * arg-index (first maximum), pool_bwd its backward; bn_relu_pool = relu_apply + pool_max straight from Y (the last
 * layer of a set-abstraction chain: its post-ReLU matrix is never written).  bwd_reduce: partial sums of
 * dz = dH.[H>0] and dz.yhat over pvn3d_mt_bn_bwd_partials(rows) row blocks -- the mask is recomputed from y
 * (H = bf16(relu(a y + b))), so H is not read; bwd_finalize: dgamma, dbeta and the affine form of the backward
 * dY = a.dz + k1.y + k0; bwd_apply applies it.  The *_pooled forms do both for a layer whose dH is a max-pool
 * backward (one nonzero per group and channel, taken from dpool / arg; P = pvn3d_mt_bn_bwd_partials(G)). */
int pvn3d_mt_bn_finalize(int P, int ld, int C, double count, const float* psum, const float* psq, const float* gamma,
                         const float* beta, float eps, float momentum, float* run_mean, float* run_var, float* mean,
                         float* invstd, float* a, float* b, void* stream);
int pvn3d_mt_bn_relu_apply(long long rows, int ld, const void* Y, const float* a, const float* b, void* H, void* stream);
int pvn3d_mt_pool_max(long long G, int ns, int ld, int C, const void* H, float* out, long long out_ld,
                      unsigned char* arg, void* stream);
int pvn3d_mt_pool_bwd(long long G, int ns, int ld, int C, const float* dout, long long out_ld, const unsigned char* arg,
                      void* dH, void* stream);
int pvn3d_mt_pack_grad(long long rows, int ld, int C, const float* g, long long gld, void* dH, void* stream);
int pvn3d_mt_unpack_out(long long rows, int ld, int C, const void* H, float* out, long long out_ld, void* stream);
int pvn3d_mt_bn_bwd_partials(long long rows);
int pvn3d_mt_bn_relu_pool(long long G, int ns, int ld, int C, const void* Y, const float* a, const float* b, float* out,
                          long long out_ld, void* arg, void* stream);
int pvn3d_mt_bn_bwd_reduce(long long rows, int ld, const void* dH, const void* Y, const float* a, const float* b,
                           const float* mean, const float* invstd, float* p1, float* p2, void* stream);
int pvn3d_mt_bn_bwd_reduce_pooled(long long G, int ns, int ld, int C, const float* dout, long long out_ld, const void* arg,
                                  const void* Y, const float* a, const float* b, const float* mean, const float* invstd,
                                  float* p1, float* p2, void* stream);
int pvn3d_mt_bn_bwd_apply_pooled(long long G, int ns, int ld, int C, const float* dout, long long out_ld, const void* arg,
                                 const void* Y, const float* a, const float* b, const float* k1, const float* k0,
                                 void* dY, void* stream);
int pvn3d_mt_bn_bwd_finalize(int P, int ld, int C, double count, const float* p1, const float* p2, const float* mean,
                             const float* invstd, const float* a, float* dgamma, float* dbeta, float* k1, float* k0,
                             void* stream);
int pvn3d_mt_bn_bwd_apply(long long rows, int ld, const void* dH, const void* Y, const float* a, const float* b,
                          const float* k1, const float* k0, void* dY, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PVN3D_HIP_H_ */
