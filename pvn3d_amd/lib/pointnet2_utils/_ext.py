"""Drop-in for the reference's compiled module ``pointnet2_utils._ext``.

Exports exactly the nine functions of pvn3d/_ext-src/src/bindings.cpp:6-19 with the same
argument order, tensor contract and error behaviour as the reference's ATen glue
(pvn3d/_ext-src/src/{sampling,ball_query,group_points,interpolate}.cpp):
  * inputs must be contiguous, fp32 / int32 (CHECK_CONTIGUOUS / CHECK_IS_FLOAT / CHECK_IS_INT,
    pvn3d/_ext-src/include/utils.h:10-25) and live on the GPU ("CPU not supported");
  * outputs are freshly allocated on the inputs' device by torch (caching allocator owns them);
  * kernels are enqueued on torch's current stream, no synchronisation.
All compute is in libpvn3d_hip.so (include/pvn3d_hip.h); this file only validates, allocates
and passes raw pointers.  Extra (non-reference) entry points used by the fused callers are
grouped at the bottom.
"""
import contextlib
import threading

import torch

from ..._lib import lib, check, on_device
from . import _fused_mlp

# Reproduce what the reference BINARY returns from three_interpolate_grad (it calls the forward
# kernel with swapped sizes, pvn3d/_ext-src/src/interpolate.cpp:89-93).  Default: the
# mathematically correct gradient.  See DESIGN.md "Reference bug: three_interpolate_grad".
REFERENCE_BUG_COMPAT = False


def _chk(t, name, dtype):
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a%s tensor" % (name, " float" if dtype == torch.float32 else "n int"))


def _same_dev(a, b, name):
    """After the dtype/contiguity checks of every argument, as in the reference glue: the first
    tensor decides CPU ("CPU not supported") vs GPU, the others must live on the same device."""
    if not a.is_cuda:
        raise RuntimeError("CPU not supported")
    if b.device != a.device:
        raise RuntimeError("%s must be a CUDA tensor" % name)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


# False: every size runs the register-resident kernel (one workgroup scans the whole cloud every round);
# True: 4096 < N <= 12288 runs the spatially culled kernel (csrc/fps_cells.hip).  Same indices either way.
FPS_CULLED = True
# waves that share a cloud's sampling rounds in the culled kernel: 0 / 1 = one wave per cloud (the default, and the
# faster form), >= 2 = one wave per 64-point slot of a cell (2-3 waves; measured 1.45 x slower, kept as an independently
# written cross-check).  Read per call; same indices either way.
FPS_WAVES = 0


def _fps_ws(B, N, dev):
    words = lib.pvn3d_fps_ws_words(N) if FPS_CULLED else (N if N > 16384 else 0)
    return torch.empty((B, words), dtype=torch.int32, device=dev) if words > 0 else None


def furthest_point_sampling(points, nsamples):
    """points (B,N,3) -> (B,nsamples) int32.  sampling.cpp:65-86"""
    _chk(points, "points", torch.float32)
    _same_dev(points, points, "points")
    B, N = points.size(0), points.size(1)
    out = torch.zeros((B, nsamples), dtype=torch.int32, device=points.device)
    # the reference's (B,N) 1e10 scratch is replaced by the workspace the library asks for
    ws = _fps_ws(B, N, points.device)
    with on_device(points.device):
        if FPS_CULLED:
            check(lib.pvn3d_furthest_point_sampling_ws_waves(B, N, int(nsamples), points.data_ptr(),
                                                             ws.data_ptr() if ws is not None else None,
                                                             out.data_ptr(), None, None, 0, int(FPS_WAVES), _stream(points)),
                  "furthest_point_sampling")
        else:
            check(lib.pvn3d_furthest_point_sampling(B, N, int(nsamples), points.data_ptr(),
                                                    ws.data_ptr() if ws is not None else None,
                                                    out.data_ptr(), _stream(points)),
                  "furthest_point_sampling")
    return out


def furthest_point_sampling_nested(points, nsamples, want_dmax=False, nest=None):
    """furthest_point_sampling for the levels of a PointNet++ pyramid (not a reference entry point).
    want_dmax: also return the winning squared distance of every round, (B, nsamples) int32 bit patterns --
    the input of fps_nest_verify.  nest = (first_rounds (B,3) int32 from fps_nest_verify, level): cloud b is
    known to be in FPS order for its first R = first_rounds[b][level] picks, so rounds below R are not run
    (R >= nsamples: the result is 0..nsamples-1 without a single round).  Returns (idx, dmax or None); idx is
    index-exact either way."""
    _chk(points, "points", torch.float32)
    _same_dev(points, points, "points")
    B, N = points.size(0), points.size(1)
    out = torch.zeros((B, nsamples), dtype=torch.int32, device=points.device)
    dmax = torch.empty((B, nsamples), dtype=torch.int32, device=points.device) if want_dmax else None
    flags, level = nest if nest is not None else (None, 0)
    if flags is not None:
        _chk(flags, "nest first_rounds", torch.int32)
        _same_dev(points, flags, "nest first_rounds")
        if tuple(flags.shape) != (B, 3):
            raise RuntimeError("nest first_rounds must be (B, 3)")
    ws = _fps_ws(B, N, points.device)
    with on_device(points.device):
        args = (B, N, int(nsamples), points.data_ptr(), ws.data_ptr() if ws is not None else None, out.data_ptr(),
                dmax.data_ptr() if dmax is not None else None, flags.data_ptr() if flags is not None else None, int(level))
        if FPS_CULLED:
            check(lib.pvn3d_furthest_point_sampling_ws_waves(*args, int(FPS_WAVES), _stream(points)),
                  "furthest_point_sampling_nested")
        else:
            check(lib.pvn3d_furthest_point_sampling_nested(*args, _stream(points)), "furthest_point_sampling_nested")
    return out, dmax


def fps_nest_verify(ordered_xyz, dmax, m_levels):
    """ordered_xyz (B,n0,3): a cloud gathered in the order of the FPS run whose per-round winning distances
    are `dmax` (B,n0); m_levels: samples of the (up to 3) following pyramid levels.  -> first_rounds (B,3)
    int32: level l's run on cloud b selects 0, 1, ..., R-1 in its first R = first_rounds[b][l] rounds
    (R >= m_l: the whole run is the identity; R = the first round in which a tie is broken differently under
    that level's block shape, or a degenerate round; 1 = the level samples a different cloud, run it all)."""
    import ctypes
    _chk(ordered_xyz, "ordered_xyz", torch.float32)
    _chk(dmax, "dmax", torch.int32)
    _same_dev(ordered_xyz, dmax, "dmax")
    B, n0 = ordered_xyz.size(0), ordered_xyz.size(1)
    if tuple(dmax.shape) != (B, n0):
        raise RuntimeError("dmax must be (B, n0)")
    ms = [int(m) for m in m_levels]
    if not 1 <= len(ms) <= 3:
        raise RuntimeError("1 to 3 following levels")
    flags = torch.empty((B, 3), dtype=torch.int32, device=ordered_xyz.device)
    arr = (ctypes.c_int * len(ms))(*ms)
    with on_device(ordered_xyz.device):
        check(lib.pvn3d_fps_nest_verify(B, n0, len(ms), arr, ordered_xyz.data_ptr(), dmax.data_ptr(),
                                        flags.data_ptr(), _stream(ordered_xyz)), "fps_nest_verify")
    return flags


def gather_points(points, idx):
    """points (B,C,N), idx (B,npoint) -> (B,C,npoint).  sampling.cpp:15-39"""
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_dev(points, idx, "idx")
    B, C, N = points.shape
    m = idx.size(1)
    out = torch.empty((B, C, m), dtype=torch.float32, device=points.device)
    with on_device(points.device):
        check(lib.pvn3d_gather_points(B, C, N, m, points.data_ptr(), idx.data_ptr(), out.data_ptr(),
                                      _stream(points)), "gather_points")
    return out


# Backward scatters: False = float atomics like the reference (result depends on the arrival order
# in the last bits); True = 64-bit fixed-point accumulation, bit-reproducible (csrc/scatter_det.hip).
DETERMINISTIC_GRADS = False


def _det_ws(B, C, n, dev):
    nbytes = int(lib.pvn3d_scatter_det_workspace_bytes(B, C, int(n)))
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


def gather_points_grad(grad_out, idx, n):
    """grad_out (B,C,npoint), idx (B,npoint) -> (B,C,n).  sampling.cpp:40-64"""
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_dev(grad_out, idx, "idx")
    B, C, m = grad_out.shape
    out = torch.empty((B, C, int(n)), dtype=torch.float32, device=grad_out.device)
    if DETERMINISTIC_GRADS:
        ws, nbytes = _det_ws(B, C, n, grad_out.device)
        with on_device(grad_out.device):
            check(lib.pvn3d_group_points_grad_det(B, C, int(n), m, 1, grad_out.data_ptr(), idx.data_ptr(),
                                                  out.data_ptr(), ws.data_ptr(), nbytes, _stream(grad_out)),
                  "gather_points_grad_det")
        return out
    with on_device(grad_out.device):
        check(lib.pvn3d_gather_points_grad(B, C, int(n), m, grad_out.data_ptr(), idx.data_ptr(),
                                           out.data_ptr(), _stream(grad_out)), "gather_points_grad")
    return out


# Clouds at least this large go through the uniform-grid ball query (identical output; the
# brute-force scan is faster for small clouds).  32768 is the grid kernel's bitmap capacity.
import os as _os
GRID_MIN_N = int(_os.environ.get("PVN3D_GRID_MIN_N", "1024"))
GRID_MAX_N = 32768


def _grid_ws(B, N, dev):
    nbytes = int(lib.pvn3d_ball_query_grid_workspace_bytes(B, N))
    return torch.empty((nbytes,), dtype=torch.uint8, device=dev), nbytes


def ball_query(new_xyz, xyz, radius, nsample):
    """new_xyz (B,npoint,3), xyz (B,N,3) -> (B,npoint,nsample) int32.  ball_query.cpp:8-32"""
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    _same_dev(new_xyz, xyz, "xyz")
    B, m = new_xyz.size(0), new_xyz.size(1)
    N = xyz.size(1)
    idx = torch.empty((B, m, int(nsample)), dtype=torch.int32, device=new_xyz.device)
    with on_device(new_xyz.device):
        if GRID_MIN_N <= N <= GRID_MAX_N and radius > 0 and m > 0:
            ws, nbytes = _grid_ws(B, N, new_xyz.device)
            check(lib.pvn3d_ball_query_pair_grid(B, N, m, float(radius), int(nsample), 0.0, 0,
                                                 new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr(),
                                                 None, ws.data_ptr(), nbytes, _stream(new_xyz)),
                  "ball_query")
        else:
            check(lib.pvn3d_ball_query(B, N, m, float(radius), int(nsample), new_xyz.data_ptr(),
                                       xyz.data_ptr(), idx.data_ptr(), _stream(new_xyz)), "ball_query")
    return idx


def group_points(points, idx):
    """points (B,C,N), idx (B,npoint,nsample) -> (B,C,npoint,nsample).  group_points.cpp:12-35"""
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_dev(points, idx, "idx")
    B, C, N = points.shape
    npoint, nsample = idx.size(1), idx.size(2)
    out = torch.empty((B, C, npoint, nsample), dtype=torch.float32, device=points.device)
    with on_device(points.device):
        check(lib.pvn3d_group_points(B, C, N, npoint, nsample, points.data_ptr(), idx.data_ptr(),
                                     out.data_ptr(), _stream(points)), "group_points")
    return out


def group_points_grad(grad_out, idx, n):
    """grad_out (B,C,npoint,nsample) -> (B,C,n).  group_points.cpp:37-60"""
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_dev(grad_out, idx, "idx")
    B, C, npoint, nsample = grad_out.shape
    out = torch.empty((B, C, int(n)), dtype=torch.float32, device=grad_out.device)
    if DETERMINISTIC_GRADS:
        ws, nbytes = _det_ws(B, C, n, grad_out.device)
        with on_device(grad_out.device):
            check(lib.pvn3d_group_points_grad_det(B, C, int(n), npoint, nsample, grad_out.data_ptr(),
                                                  idx.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes,
                                                  _stream(grad_out)), "group_points_grad_det")
        return out
    with on_device(grad_out.device):
        check(lib.pvn3d_group_points_grad(B, C, int(n), npoint, nsample, grad_out.data_ptr(),
                                          idx.data_ptr(), out.data_ptr(), _stream(grad_out)),
              "group_points_grad")
    return out


# three_nn through a uniform grid over the known points (csrc/three_nn_grid.hip), same output
NN_GRID = _os.environ.get("PVN3D_NN_GRID", "1") != "0"
NN_GRID_MIN_M, NN_GRID_MAX_M, NN_GRID_MIN_N = 64, 2048, 512


def three_nn(unknowns, knows):
    """unknowns (B,n,3), knows (B,m,3) -> [dist2 (B,n,3), idx (B,n,3)].  interpolate.cpp:14-40"""
    _chk(unknowns, "unknowns", torch.float32)
    _chk(knows, "knows", torch.float32)
    _same_dev(unknowns, knows, "knows")
    B, n = unknowns.size(0), unknowns.size(1)
    m = knows.size(1)
    idx = torch.empty((B, n, 3), dtype=torch.int32, device=unknowns.device)
    dist2 = torch.empty((B, n, 3), dtype=torch.float32, device=unknowns.device)
    with on_device(unknowns.device):
        if NN_GRID and NN_GRID_MIN_M <= m <= NN_GRID_MAX_M and n >= NN_GRID_MIN_N:
            # identical output; ~30x fewer distance evaluations for evenly sampled surfaces
            nbytes = int(lib.pvn3d_three_nn_grid_workspace_bytes(B, m))
            ws = torch.empty((nbytes,), dtype=torch.uint8, device=unknowns.device)
            check(lib.pvn3d_three_nn_grid(B, n, m, unknowns.data_ptr(), knows.data_ptr(), dist2.data_ptr(),
                                          idx.data_ptr(), ws.data_ptr(), nbytes, _stream(unknowns)),
                  "three_nn_grid")
        else:
            check(lib.pvn3d_three_nn(B, n, m, unknowns.data_ptr(), knows.data_ptr(), dist2.data_ptr(),
                                     idx.data_ptr(), _stream(unknowns)), "three_nn")
    return [dist2, idx]


def three_nn_weights(dist2):
    """dist2 (B,n,3) of three_nn -> the inverse-distance weights of PointnetFPModule.forward
    (pointnet2_modules.py:184-186), same fp32 operation order, one launch."""
    _chk(dist2, "dist2", torch.float32)
    w = torch.empty_like(dist2)
    with on_device(dist2.device):
        check(lib.pvn3d_three_nn_weights(dist2.numel() // 3, dist2.data_ptr(), w.data_ptr(), _stream(dist2)),
              "three_nn_weights")
    return w


def three_interpolate(points, idx, weight):
    """points (B,C,m), idx/weight (B,n,3) -> (B,C,n).  interpolate.cpp:42-68"""
    _chk(points, "points", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_dev(points, idx, "idx")
    _same_dev(points, weight, "weight")
    B, C, m = points.shape
    n = idx.size(1)
    out = torch.empty((B, C, n), dtype=torch.float32, device=points.device)
    with on_device(points.device):
        check(lib.pvn3d_three_interpolate(B, C, m, n, points.data_ptr(), idx.data_ptr(),
                                          weight.data_ptr(), out.data_ptr(), _stream(points)),
              "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, m):
    """grad_out (B,C,n), idx/weight (B,n,3) -> (B,C,m).  interpolate.cpp:70-99"""
    _chk(grad_out, "grad_out", torch.float32)
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_dev(grad_out, idx, "idx")
    _same_dev(grad_out, weight, "weight")
    B, C, n = grad_out.shape
    out = torch.empty((B, C, int(m)), dtype=torch.float32, device=grad_out.device)
    if DETERMINISTIC_GRADS and not REFERENCE_BUG_COMPAT:
        ws, nbytes = _det_ws(B, C, m, grad_out.device)
        with on_device(grad_out.device):
            check(lib.pvn3d_three_interpolate_grad_det(B, C, n, int(m), grad_out.data_ptr(), idx.data_ptr(),
                                                       weight.data_ptr(), out.data_ptr(), ws.data_ptr(), nbytes,
                                                       _stream(grad_out)), "three_interpolate_grad_det")
        return out
    with on_device(grad_out.device):
        check(lib.pvn3d_three_interpolate_grad(B, C, n, int(m), grad_out.data_ptr(), idx.data_ptr(),
                                               weight.data_ptr(), out.data_ptr(),
                                               1 if REFERENCE_BUG_COMPAT else 0, _stream(grad_out)),
              "three_interpolate_grad")
    return out


# ------------------------------------------------------------------------------------------
# fused entry points (not in the reference's _ext; used by QueryAndGroup / SA-MSG modules)
# ------------------------------------------------------------------------------------------

def ball_query_pair(new_xyz, xyz, radius0, nsample0, radius1, nsample1):
    """Two ball queries over the same (new_xyz, xyz) in one scan -> (idx0, idx1)."""
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(xyz, "xyz", torch.float32)
    _same_dev(new_xyz, xyz, "xyz")
    B, m = new_xyz.size(0), new_xyz.size(1)
    N = xyz.size(1)
    idx0 = torch.empty((B, m, int(nsample0)), dtype=torch.int32, device=new_xyz.device)
    idx1 = torch.empty((B, m, int(nsample1)), dtype=torch.int32, device=new_xyz.device)
    with on_device(new_xyz.device):
        if GRID_MIN_N <= N <= GRID_MAX_N and radius0 > 0 and radius1 > 0 and m > 0:
            ws, nbytes = _grid_ws(B, N, new_xyz.device)
            check(lib.pvn3d_ball_query_pair_grid(B, N, m, float(radius0), int(nsample0),
                                                 float(radius1), int(nsample1), new_xyz.data_ptr(),
                                                 xyz.data_ptr(), idx0.data_ptr(), idx1.data_ptr(),
                                                 ws.data_ptr(), nbytes, _stream(new_xyz)),
                  "ball_query_pair")
        else:
            check(lib.pvn3d_ball_query_pair(B, N, m, float(radius0), int(nsample0), float(radius1),
                                            int(nsample1), new_xyz.data_ptr(), xyz.data_ptr(),
                                            idx0.data_ptr(), idx1.data_ptr(), _stream(new_xyz)),
                  "ball_query_pair")
    return idx0, idx1


def group_xyz_features(xyz, new_xyz, features, idx, use_xyz=True):
    """QueryAndGroup's gather + subtract + cat in one pass -> (B, 3*use_xyz + C, npoint, nsample)."""
    _chk(xyz, "xyz", torch.float32)
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_dev(xyz, new_xyz, "new_xyz")
    _same_dev(xyz, idx, "idx")
    C = 0
    if features is not None:
        _chk(features, "features", torch.float32)
        _same_dev(xyz, features, "features")
        C = features.size(1)
    B, N = xyz.size(0), xyz.size(1)
    m, nsample = idx.size(1), idx.size(2)
    c_out = (3 if use_xyz else 0) + C
    out = torch.empty((B, c_out, m, nsample), dtype=torch.float32, device=xyz.device)
    with on_device(xyz.device):
        check(lib.pvn3d_group_xyz_features(B, N, m, C, nsample, 1 if use_xyz else 0, xyz.data_ptr(),
                                           new_xyz.data_ptr(),
                                           features.data_ptr() if features is not None else None,
                                           idx.data_ptr(), out.data_ptr(), _stream(xyz)),
              "group_xyz_features")
    return out


def group_xyz_features_pair(xyz, new_xyz, features, idx0, idx1):
    """QueryAndGroup (use_xyz=True) for both radii of a multi-scale level in one launch ->
    ((B, 3 + C, npoint, ns0), (B, 3 + C, npoint, ns1)); same values as two group_xyz_features calls."""
    for t, name, dt in ((xyz, "xyz", torch.float32), (new_xyz, "new_xyz", torch.float32), (features, "features", torch.float32),
                        (idx0, "idx0", torch.int32), (idx1, "idx1", torch.int32)):
        _chk(t, name, dt)
        _same_dev(xyz, t, name)
    B, N = xyz.size(0), xyz.size(1)
    C = features.size(1)
    m = idx0.size(1)
    assert idx1.size(1) == m and idx0.size(0) == B and idx1.size(0) == B
    ns0, ns1 = idx0.size(2), idx1.size(2)
    out0 = torch.empty((B, 3 + C, m, ns0), dtype=torch.float32, device=xyz.device)
    out1 = torch.empty((B, 3 + C, m, ns1), dtype=torch.float32, device=xyz.device)
    with on_device(xyz.device):
        check(lib.pvn3d_group_xyz_features_pair(B, N, m, C, ns0, ns1, xyz.data_ptr(), new_xyz.data_ptr(),
                                                features.data_ptr(), idx0.data_ptr(), idx1.data_ptr(), out0.data_ptr(),
                                                out1.data_ptr(), _stream(xyz)), "group_xyz_features_pair")
    return out0, out1


_NARROW_MISS_SEEN = set()


def _note_narrow_miss(is_sa, c_a, c_b, nsample, packed, out_point_major=0):
    """Once per chain shape: a chain whose layers are all <= 128 wide (what the narrow-chain kernels are for) but whose
    widths are not among the instantiated ones runs on the 4 + 4-wave kernel -- 2-4 x slower for such chains.  The
    library answers which family it would pick (pvn3d_mlp_split2_kernel); this is the message a host with a different
    backbone gets instead of a silent slowdown."""
    dims = tuple(int(d) for d in packed.dims)
    if not NARROW_KERNELS or max(dims[1:]) > 128 or (is_sa and nsample not in (16, 32)):
        return
    key = (bool(is_sa), int(c_a), int(c_b), int(nsample), dims, int(out_point_major))
    if key in _NARROW_MISS_SEEN:
        return
    _NARROW_MISS_SEEN.add(key)
    fam = lib.pvn3d_mlp_split2_kernel(int(is_sa), int(c_a), int(c_b), int(nsample), packed.n_layers, packed.dims_c,
                                      int(out_point_major), _mlp_flags())
    if fam == 1:
        import warnings
        warnings.warn("pvn3d_amd: the %s chain %s (%s) has no narrow-chain kernel instance and runs on the 4 + 4-wave "
                      "kernel; the narrow kernels (csrc/sa_mlp_split.hip: nw_signature / nwfp_ok) are instantiated for the "
                      "PVN3D backbone's widths" % ("set-abstraction" if is_sa else "feature-propagation", list(dims),
                                                   "nsample %d" % nsample if is_sa else "skip %d" % c_b), stacklevel=3)


def _point_major(t):
    """(B, C, n) tensor -> (base tensor, ld) of a point-major (B, n, ld) table holding it.
    Zero-copy when `t` already is a transposed view of a point-major buffer (what the fused
    modules hand to each other, and what `pc[..., 3:].transpose(1, 2)` is); otherwise one
    tiled transpose (csrc/sa_mlp.hip)."""
    B, C, n = t.shape
    if t.stride(1) == 1 and t.stride(2) >= C and t.stride(0) == n * t.stride(2):
        return t, t.stride(2)
    src = t if t.is_contiguous() else t.contiguous()
    ld = (C + 3) // 4 * 4
    out = torch.empty((B, n, ld), dtype=torch.float32, device=t.device)
    with on_device(t.device):
        check(lib.pvn3d_transpose_bcn_to_bnc(B, C, n, src.data_ptr(), out.data_ptr(), ld, _stream(t)),
              "transpose_bcn_to_bnc")
    return out, ld


# fp16 x 2 also for chains whose first hidden layer is narrower than 128 channels (SA level 1 of the backbone); False keeps
# those on the fp32-MFMA kernels (A/B switch)
SPLIT2_NARROW = True
# The narrow-chain kernels (csrc/sa_mlp_split.hip: SA levels 0-1, the pre-contracted FP level 0) behind the fp16 x 2 entry
# points; False passes PVN3D_MLP_NO_NARROW with every call (A/B measurements -- a per-call flag of the C ABI, read here per
# call; the library itself has no switch)
NARROW_KERNELS = True


IDENTITY_SKIP = True      # pass PVN3D_MLP_IDENTITY_A for pre-contracted chains (False: A/B -- the results are the same bits)


def _mlp_flags(packed=None):
    f = 0 if NARROW_KERNELS else 1             # PVN3D_MLP_NO_NARROW
    if IDENTITY_SKIP and packed is not None and getattr(packed, "identity_a", False):
        f |= 2                                 # PVN3D_MLP_IDENTITY_A
    return f


def invalidate_table_caches(t):
    """Drop the abs-max bound and the h16 copy cached on tensor object `t` (table_absmax / table_h16 / seed_absmax).
    The caches are keyed on (data_ptr, shape, t._version); _version only moves on torch in-place ops, so a table that is
    REWRITTEN through a raw pointer -- a kernel of this library writing into a persistent buffer, a HIP-graph replay into a
    static tensor -- while the same view object is reused keeps its old bound and its old h16 copy.  A bound that is too
    small is not harmless: the scaled operand then exceeds fp16's range.  Pointnet2MSG builds fresh views every forward and
    is not affected; callers of sa_mlp_maxpool / fp_interp_mlp that keep feeding one tensor object whose contents they
    rewrite out of torch's sight call this after every rewrite."""
    for name in ("_pvn3d_absmax", "_pvn3d_h16", "_pvn3d_bound"):
        if hasattr(t, name):
            try:
                delattr(t, name)
            except AttributeError:
                pass


# Device-side scalars that a kernel accumulates into (abs-max words, bounds) have to start at zero: a dozen one-word
# torch.zeros per 64-frame forward, each a fill kernel of its own on the feature path's stream.  Inside `zero_arena` they
# are 16-byte slices of ONE zeroed buffer (one fill).  Thread-local: the evaluator's worker threads run their own calls.
_ZERO_ARENA = threading.local()


@contextlib.contextmanager
def zero_arena(device, floats=512):
    prev = getattr(_ZERO_ARENA, "state", None)
    _ZERO_ARENA.state = [torch.zeros(floats, dtype=torch.float32, device=device), 0]
    try:
        yield
    finally:
        _ZERO_ARENA.state = prev


def zeros_f32(n, device):
    """n zeroed float32 words on `device` (a 16-byte-aligned slice of the current zero_arena, or a tensor of their own)."""
    st = getattr(_ZERO_ARENA, "state", None)
    if st is not None and st[0].device == torch.device(device):
        off, need = st[1], (int(n) + 3) // 4 * 4
        if off + need <= st[0].numel():
            st[1] = off + need
            return st[0][off:off + int(n)]
    return torch.zeros(int(n), dtype=torch.float32, device=device)


def table_absmax(base, rows, c, ld):
    """Device float32[1] holding max|x| over the point-major table (rows, c) at `base` (row stride ld): the input bound
    the fp16 x 2 kernels scale by (include/pvn3d_hip.h).  One reduction per table: the result is cached on the tensor
    object (the fused levels hand the SAME view object to every consumer of a level's output)."""
    inherited = getattr(base, "_pvn3d_bound", None)       # a bound that holds for every table inside this tensor's data
    if inherited is not None:
        return inherited
    key = (base.data_ptr(), int(rows), int(c), int(ld), base._version)
    cache = getattr(base, "_pvn3d_absmax", None)
    if cache is not None and cache[0] == key:
        return cache[1]
    out = zeros_f32(1, base.device)
    with on_device(base.device):
        check(lib.pvn3d_absmax(int(rows), int(c), base.data_ptr(), int(ld), out.data_ptr(), _stream(base)), "absmax")
    try:
        base._pvn3d_absmax = (key, out)
    except AttributeError:
        pass
    return out


def table_h16(base, rows, c, ld, slabs):
    """-> (h16 buffer, bound): the (rows, c) point-major table at `base` as an h16 matrix of `slabs` 16-k slabs
    (pvn3d_split_rows2) together with the device-side bound it was written with.  Cached on the tensor object like the
    bound: a level's output is split once although two consumers contract over it (SA level l + 1's pre-contraction and
    the skip half of an FP level read the same table)."""
    bound = table_absmax(base, rows, c, ld)
    key = (base.data_ptr(), int(rows), int(c), int(ld), int(slabs), base._version, bound.data_ptr())
    cache = getattr(base, "_pvn3d_h16", None)
    if cache is not None and cache[0] == key:
        return cache[1], bound
    xs = torch.empty((int(rows) * int(slabs) * 64,), dtype=torch.uint8, device=base.device)
    with on_device(base.device):
        check(lib.pvn3d_split_rows2(int(rows), int(c), base.data_ptr(), int(ld), bound.data_ptr(), xs.data_ptr(), int(slabs),
                                    _stream(base)), "split_rows2")
    try:
        base._pvn3d_h16 = (key, xs)
    except AttributeError:
        pass
    return xs, bound


def seed_absmax(view, rows, c, ld, bound):
    """Attach `bound` (device float32[1], already known -- e.g. a GEMM's out_absmax) to `view` as the cached result of
    table_absmax(view, rows, c, ld)."""
    view._pvn3d_absmax = ((view.data_ptr(), int(rows), int(c), int(ld), view._version), bound)
    return view


def sa_mlp_maxpool(xyz, new_xyz, features, idx, use_xyz, packed, out_pm=None, out_coff=0, out_absmax=None):
    """(out_absmax: optional device float32[1] that the fp16 x 2 kernel raises to max|output| -- the caller zeroes it once
    per output table and seeds the table's bound with it; ignored by the other arithmetics.)
    Fused group -> SharedMLP (BN folded, fp32 MFMA) -> max over nsample, inference only.
    packed: _fused_mlp.PackedMLP built with n_xyz_first=3 when use_xyz.  features: (B, C, n) in
    any layout (a transposed view of a point-major buffer is used in place).  Writes the
    packed.dims[-1] pooled channels into the point-major buffer out_pm (B, npoint, ld) at channel
    offset out_coff (allocated if None) and returns the (B, packed.dims[-1], npoint) VIEW of it."""
    _chk(xyz, "xyz", torch.float32)
    _chk(new_xyz, "new_xyz", torch.float32)
    _chk(idx, "idx", torch.int32)
    _same_dev(xyz, new_xyz, "new_xyz")
    _same_dev(xyz, idx, "idx")
    C, feat, ld_feat = 0, None, 0
    if features is not None:
        if not features.is_cuda:
            raise RuntimeError("CPU not supported")
        if features.dtype != torch.float32:
            raise RuntimeError("features must be a float tensor")
        _same_dev(xyz, features, "features")
        C = features.size(1)
        feat, ld_feat = _point_major(features)
    B, N = xyz.size(0), xyz.size(1)
    m, nsample = idx.size(1), idx.size(2)
    M = packed.dims[-1]
    if out_pm is None:
        out_pm = torch.empty((B, m, (M + 3) // 4 * 4), dtype=torch.float32, device=xyz.device)
        out_coff = 0
    ld_out = out_pm.size(2)
    vec = use_xyz and feat is not None and ld_feat % 4 == 0 and feat.data_ptr() % 16 == 0
    # (the narrow-chain kernel reads a six-feature table in place, whatever its row stride: the view pc[..., 3:] of SA
    # level 0)
    few = use_xyz and feat is not None and C == 6
    # (fp16x2_safe: the host-side probe of the rescaled chain -- _fused_mlp.fp16x2_probe; a chain whose weights two fp16
    # pieces cannot hold falls through to the bf16 x 3 / fp32 kernels)
    if ((vec or few) and _fused_mlp.MLP_ARITH == "fp16x2" and (SPLIT2_NARROW or packed.dims[1] >= 128)
            and lib.pvn3d_mlp_split2_ok(1, C, 0, nsample, packed.n_layers, packed.dims_c, _mlp_flags())
            and packed.fp16x2_safe()):
        w2, meta, b2, rinv = packed.split2()
        _note_narrow_miss(1, C, 0, nsample, packed)
        fa, xa = table_absmax(feat, B * N, C, ld_feat), table_absmax(xyz, B * N, 3, 3)
        with on_device(xyz.device):
            check(lib.pvn3d_sa_mlp_maxpool_split2(B, N, m, C, nsample, xyz.data_ptr(), new_xyz.data_ptr(), feat.data_ptr(),
                                                  ld_feat, idx.data_ptr(), packed.n_layers, packed.dims_c, w2, b2,
                                                  meta, fa.data_ptr(), xa.data_ptr(), out_pm.data_ptr(), ld_out, out_coff,
                                                  out_absmax.data_ptr() if out_absmax is not None else None,
                                                  rinv.data_ptr(), _mlp_flags(packed), _stream(xyz)), "sa_mlp_maxpool_split2")
        if out_absmax is not None:
            out_absmax._pvn3d_written = True
        return out_pm[:, :, out_coff:out_coff + M].transpose(1, 2)
    if (vec and _fused_mlp.split_arith()
            and lib.pvn3d_mlp_split_ok(1, C, 0, nsample, packed.n_layers, packed.dims_c)):
        with on_device(xyz.device):
            check(lib.pvn3d_sa_mlp_maxpool_split(B, N, m, C, nsample, xyz.data_ptr(), new_xyz.data_ptr(), feat.data_ptr(),
                                                 ld_feat, idx.data_ptr(), packed.n_layers, packed.dims_c, packed.split(),
                                                 packed.b_c, out_pm.data_ptr(), ld_out, out_coff, _stream(xyz)),
                  "sa_mlp_maxpool_split")
        return out_pm[:, :, out_coff:out_coff + M].transpose(1, 2)
    if packed._equil_given is not None:
        raise RuntimeError("a rescaled (fp16 x 2) chain on a kernel that does not undo the rescaling")
    with on_device(xyz.device):
        check(lib.pvn3d_sa_mlp_maxpool(B, N, m, C, nsample, 1 if use_xyz else 0, xyz.data_ptr(),
                                       new_xyz.data_ptr(), feat.data_ptr() if feat is not None else None,
                                       ld_feat, idx.data_ptr(), packed.n_layers, packed.dims_c, packed.w_c,
                                       packed.b_c, out_pm.data_ptr(), ld_out, out_coff, _stream(xyz)),
              "sa_mlp_maxpool")
    return out_pm[:, :, out_coff:out_coff + M].transpose(1, 2)


# Set-abstraction levels whose first layer is narrower than their input (SA levels 2-3 of the backbone: 256 -> 128,
# 512 -> 256): the feature half of the first conv is applied per SOURCE point before the gather -- one split-bf16 GEMM
# for all scales of the level (csrc/split_gemm.hip) -- and the fused chain gathers M0 instead of C channels per sample
# and contracts [I | Wx] over them: grouping is linear, so conv([xyz_rel; f[idx]]) = Wx.xyz_rel + (Wf.f)[idx].
SA_PRECONTRACT = True


def sa_precontract(features, packs, nsamples):
    """-> [(features' (B, M0, n) view, PackedMLP') per scale] or None when the level does not qualify.
    pointnet2_modules.py:57-69 / pointnet2_utils.py:293-330 regrouped; the fp32 rounding sequence of layer 0 changes
    (the identity part of the new layer 0 is exact), accuracy against fp64 does not (tests/test_gpu_ops.py)."""
    if not (SA_PRECONTRACT and _fused_mlp.split_arith()) or features is None or not features.is_cuda:
        return None
    B, C, n = features.shape
    if C < 128 or B * n < 4096 or features.dtype != torch.float32:
        return None
    pres = []
    h2 = _fused_mlp.MLP_ARITH == "fp16x2" and all(p.fp16x2_safe() for p in packs)
    for p, ns in zip(packs, nsamples):
        if p.n_layers != 3 or p.dims[0] != C + 3 or p.dims[1] % 32 != 0 or 2 * p.dims[1] > C:
            return None
        pre, wf = p.precontracted(C, equil=h2)
        if h2:
            ok = lib.pvn3d_mlp_split2_ok(1, p.dims[1], 0, ns, pre.n_layers, pre.dims_c, _mlp_flags())
        else:
            ok = lib.pvn3d_mlp_split_ok(1, p.dims[1], 0, ns, pre.n_layers, pre.dims_c)
        if not ok:
            return None
        pres.append((pre, wf))
    feat, ld = _point_major(features)
    if ld % 4 != 0 or feat.data_ptr() % 16 != 0:
        return None
    key = (tuple(id(p) for p in packs), h2)
    cache = getattr(packs[0], "_pre_cat", None)
    if cache is None or cache[0] != key:
        wcat = torch.cat([wf for _, wf in pres], 0)
        # (fp16 x 2: every row with its own power-of-two scale, undone by the GEMM's w_row_mul)
        wp, rm = _fused_mlp._pack_weight_h16_rows(wcat, _fused_mlp._slabs(C)) if h2 else \
            (_fused_mlp._pack_weight_s16(wcat, _fused_mlp._slabs(C)), None)
        cache = (key, wp, list(packs), rm)
        packs[0]._pre_cat = cache
    ws, rm = cache[1], cache[3]
    S, n_out = _fused_mlp._slabs(C), ws.size(0)
    dev = features.device
    st = _stream(features)
    y = torch.empty((B, n, n_out), dtype=torch.float32, device=dev)
    amax = None
    if h2:
        xs, fa = table_h16(feat, B * n, C, ld, S)
        amax = zeros_f32(1, dev)
        with on_device(dev):
            check(lib.pvn3d_split_gemm2(B * n, n_out, S, xs.data_ptr(), fa.data_ptr(), ws.data_ptr(), 1.0, rm.data_ptr(), None, 0,
                                        None, 0, 0, 0, None, None, y.data_ptr(), n_out, amax.data_ptr(), None, 0, None, st),
                  "split_gemm2")
    else:
        xs = torch.empty((B * n * S * 96,), dtype=torch.uint8, device=dev)
        with on_device(dev):
            check(lib.pvn3d_split_rows(B * n, C, feat.data_ptr(), ld, xs.data_ptr(), S, st), "split_rows")
            check(lib.pvn3d_split_gemm(B * n, n_out, S, xs.data_ptr(), ws.data_ptr(), None, 0, None, 0, 0, 0, None, None,
                                       y.data_ptr(), n_out, None, 0, st), "split_gemm")
    out, off = [], 0
    for pre, _ in pres:
        m0 = pre.dims[1]
        view = y[:, :, off:off + m0].transpose(1, 2)
        if amax is not None:
            seed_absmax(view, B * n, m0, n_out, amax)           # max|y| bounds every channel slice of y
        out.append((view, pre))
        off += m0
    return out


# Wide two-layer FP chains (hidden layer >= 256 channels: 64 columns of it do not fit one CU's LDS as three bf16 pieces)
# run layer by layer on the split-bf16 GEMM of csrc/split_gemm.hip instead of the fused fp32-MFMA chain.
FP_LAYERWISE_SPLIT = True
FP_PRECONTRACT = True


def fp_layerwise_shape_ok(n_points, c1, dims):
    """The shape half of the dispatch test of fp_interp_mlp (bench.py prices a chain by the pipe it runs on): a
    two-layer chain with skip features whose layers are at least 256 wide, on at least 4096 points, that the fused
    split kernel does not take."""
    return (_fused_mlp.split_arith() and FP_LAYERWISE_SPLIT and len(dims) == 3 and c1 > 0
            and min(dims[1], dims[2]) >= 256 and n_points >= 4096)


def _fp_layerwise_split(B, n, m, C2, C1, kf, ld_k, uf, ld_u, idx, weight, packed, out, ld_out):
    """H = relu(Wb.skip + interp(Wa.known) + b1); out = relu(W2.H + b2) -- the FP module's
    conv([interp(known); skip]) -> bn -> relu -> conv -> bn -> relu (pointnet2_modules.py:188-206) with the first
    conv pulled through the (linear) interpolation, so that it runs over the m known points.  Three launches of
    pvn3d_split_gemm + two row splits; every intermediate is a torch allocation (caching allocator)."""
    dev = kf.device
    st = _stream(kf)
    P, Pk = B * n, B * m
    if _fused_mlp.MLP_ARITH == "fp16x2" and packed.h16_safe(C2):
        # the same three launches in the two-piece fp16 arithmetic (pvn3d_split_gemm2): operand scales from device-side
        # bounds -- abs-max of the two inputs, the rigorous bound of H from them -- and the output's abs-max for its consumer
        w = packed.h16(C2)
        n1p = w["b1"].numel()
        xk, ka = table_h16(kf, Pk, C2, ld_k, w["s_a"])
        xu, ua = table_h16(uf, P, C1, ld_u, w["s_b"])
        z = torch.empty((Pk, n1p), dtype=torch.float32, device=dev)
        h = torch.empty((P * w["s_h"] * 64,), dtype=torch.uint8, device=dev)
        bnd = zeros_f32(2, dev)            # [0] bound of H, [1] abs-max of the output
        with on_device(dev):
            # |H| <= ||Wb||_inf max|skip| + ||Wa||_inf max|known| + max|b1|  (interpolation weights are >= 0 and sum to 1)
            check(lib.pvn3d_bound_affine(bnd.data_ptr(), ua.data_ptr(), w["nb"], ka.data_ptr(), w["na"], w["b1max"], st),
                  "bound_affine")
            check(lib.pvn3d_split_gemm2(Pk, n1p, w["s_a"], xk.data_ptr(), ka.data_ptr(), w["wa"].data_ptr(), 1.0,
                                        w["rm_a"].data_ptr(), None, 0, None, 0, 0, 0, None, None, z.data_ptr(), n1p, None, None,
                                        0, None, st), "split_gemm2")
            check(lib.pvn3d_split_gemm2(P, w["n1"], w["s_b"], xu.data_ptr(), ua.data_ptr(), w["wb"].data_ptr(), 1.0,
                                        w["rm_b"].data_ptr(), w["b1"].data_ptr(), 1, z.data_ptr(), n1p, n, m, idx.data_ptr(),
                                        weight.data_ptr(), None, 0, None, h.data_ptr(), w["s_h"], bnd.data_ptr(), st),
                  "split_gemm2")
            check(lib.pvn3d_split_gemm2(P, w["n2"], w["s_h"], h.data_ptr(), bnd.data_ptr(), w["w2"].data_ptr(), 1.0,
                                        w["rm_2"].data_ptr(), w["b2"].data_ptr(), 1, None, 0, 0, 0, None, None, out.data_ptr(),
                                        ld_out, bnd.data_ptr() + 4, None, 0, None, st), "split_gemm2")
        return bnd[1:2]
    w = packed.s16(C2)
    n1p = w["b1"].numel()
    xk = torch.empty((Pk * w["s_a"] * 96,), dtype=torch.uint8, device=dev)
    xu = torch.empty((P * w["s_b"] * 96,), dtype=torch.uint8, device=dev)
    z = torch.empty((Pk, n1p), dtype=torch.float32, device=dev)
    h = torch.empty((P * w["s_h"] * 96,), dtype=torch.uint8, device=dev)
    with on_device(dev):
        check(lib.pvn3d_split_rows(Pk, C2, kf.data_ptr(), ld_k, xk.data_ptr(), w["s_a"], st), "split_rows")
        check(lib.pvn3d_split_rows(P, C1, uf.data_ptr(), ld_u, xu.data_ptr(), w["s_b"], st), "split_rows")
        # Z over all n1p columns: the pad columns are computed zeros (zero weight rows), the gather below reads them
        check(lib.pvn3d_split_gemm(Pk, n1p, w["s_a"], xk.data_ptr(), w["wa"].data_ptr(), None, 0, None, 0, 0, 0, None,
                                   None, z.data_ptr(), n1p, None, 0, st), "split_gemm")
        check(lib.pvn3d_split_gemm(P, w["n1"], w["s_b"], xu.data_ptr(), w["wb"].data_ptr(), w["b1"].data_ptr(), 1,
                                   z.data_ptr(), n1p, n, m, idx.data_ptr(), weight.data_ptr(), None, 0, h.data_ptr(),
                                   w["s_h"], st), "split_gemm")
        check(lib.pvn3d_split_gemm(P, w["n2"], w["s_h"], h.data_ptr(), w["w2"].data_ptr(), w["b2"].data_ptr(), 1,
                                   None, 0, 0, 0, None, None, out.data_ptr(), ld_out, None, 0, st), "split_gemm")
    return None


def fp_interp_mlp(known_feats, unknow_feats, idx, weight, packed, point_major_out=False):
    """Fused three_interpolate ++ unknow_feats -> SharedMLP (BN folded, fp32 MFMA), inference
    only.  known_feats (B, C2, m), unknow_feats (B, C1, n) in any layout (see _point_major).
    Returns (B, packed.dims[-1], n): contiguous, or with point_major_out a transposed view of a
    (B, n, M) buffer (what the next fused level gathers from)."""
    if not known_feats.is_cuda:
        raise RuntimeError("CPU not supported")
    if known_feats.dtype != torch.float32:
        raise RuntimeError("known_feats must be a float tensor")
    _chk(idx, "idx", torch.int32)
    _chk(weight, "weight", torch.float32)
    _same_dev(known_feats, idx, "idx")
    _same_dev(known_feats, weight, "weight")
    B, C2, m = known_feats.shape
    n = idx.size(1)
    kf, ld_k = _point_major(known_feats)
    C1, uf, ld_u = 0, None, 0
    if unknow_feats is not None:
        if unknow_feats.dtype != torch.float32:
            raise RuntimeError("unknow_feats must be a float tensor")
        _same_dev(known_feats, unknow_feats, "unknow_feats")
        C1 = unknow_feats.size(1)
        uf, ld_u = _point_major(unknow_feats)
    M = packed.dims[-1]
    if point_major_out:
        ld_out = (M + 3) // 4 * 4
        out = torch.empty((B, n, ld_out), dtype=torch.float32, device=known_feats.device)
    else:
        ld_out = 0
        out = torch.empty((B, M, n), dtype=torch.float32, device=known_feats.device)
    # FP level 0 (12288 <- 2048 points, 256 -> 128): the interpolated half of the first conv per KNOWN point ahead of the
    # interpolation (six times fewer points, and the chain then gathers 128 instead of 256 channels per neighbour) --
    # the regrouping of _fp_layerwise_split with the fused kernel behind it
    h2 = _fused_mlp.MLP_ARITH == "fp16x2" and packed.fp16x2_safe()

    def split_ok(*a):
        return lib.pvn3d_mlp_split2_ok(*a, _mlp_flags()) if h2 else lib.pvn3d_mlp_split_ok(*a)
    precontracted = False          # layer 0 of `packed` starts with an identity block over kf's channels
    if (FP_PRECONTRACT and _fused_mlp.split_arith() and packed.n_layers == 2 and C1 > 0 and C2 >= 256
            and packed.dims[1] % 32 == 0 and 2 * packed.dims[1] <= C2 and n >= 4 * m and B * m >= 4096
            and ld_k % 4 == 0 and kf.data_ptr() % 16 == 0
            and (C1 < 32 or (ld_u % 4 == 0 and uf.data_ptr() % 16 == 0))):
        pre, wa = packed.precontracted(C2, equil=h2)
        if split_ok(0, pre.dims[1], C1, 0, pre.n_layers, pre.dims_c):
            cache = getattr(packed, "_pre_s16", None)
            if cache is None or cache[0] != (C2, h2):       # keyed on the split point like PackedMLP.precontracted()
                wp, rm = _fused_mlp._pack_weight_h16_rows(wa, _fused_mlp._slabs(C2)) if h2 else \
                    (_fused_mlp._pack_weight_s16(wa, _fused_mlp._slabs(C2)), None)
                cache = packed._pre_s16 = ((C2, h2), wp, rm)
            wp, rm = cache[1], cache[2]
            S, n_out = _fused_mlp._slabs(C2), wp.size(0)
            dev, st = known_feats.device, _stream(known_feats)
            z = torch.empty((B, m, n_out), dtype=torch.float32, device=dev)
            if h2:
                xs, ka = table_h16(kf, B * m, C2, ld_k, S)
                amax = zeros_f32(1, dev)
                with on_device(dev):
                    check(lib.pvn3d_split_gemm2(B * m, n_out, S, xs.data_ptr(), ka.data_ptr(), wp.data_ptr(), 1.0, rm.data_ptr(),
                                                None, 0, None, 0, 0, 0, None, None, z.data_ptr(), n_out, amax.data_ptr(), None,
                                                0, None, st), "split_gemm2")
                seed_absmax(z, B * m, pre.dims[1], n_out, amax)
            else:
                xs = torch.empty((B * m * S * 96,), dtype=torch.uint8, device=dev)
                with on_device(dev):
                    check(lib.pvn3d_split_rows(B * m, C2, kf.data_ptr(), ld_k, xs.data_ptr(), S, st), "split_rows")
                    check(lib.pvn3d_split_gemm(B * m, n_out, S, xs.data_ptr(), wp.data_ptr(), None, 0, None, 0, 0, 0, None,
                                               None, z.data_ptr(), n_out, None, 0, st), "split_gemm")
            kf, ld_k, C2, packed = z, n_out, pre.dims[1], pre
            precontracted = True
    vec = ld_k % 4 == 0 and kf.data_ptr() % 16 == 0 and (C1 < 32 or (ld_u % 4 == 0 and uf.data_ptr() % 16 == 0))
    if (vec and h2
            and lib.pvn3d_mlp_split2_ok(0, C2, C1, 0, packed.n_layers, packed.dims_c, _mlp_flags())):
        w2, meta, b2, rinv = packed.split2()
        # a point-major output feeds another fused level: leave its abs-max for that level's operand scale
        amax = zeros_f32(1, known_feats.device) if point_major_out else None
        ka = table_absmax(kf, B * m, C2, ld_k)
        ua = table_absmax(uf, B * n, C1, ld_u) if uf is not None else None
        # (the pre-contracted form has an entry point of its own: the shapes its kernel takes add the interpolated rows
        # to the accumulators instead of multiplying them with the identity block)
        entry = lib.pvn3d_fp_interp_add_mlp_split2 if precontracted else lib.pvn3d_fp_interp_mlp_split2
        with on_device(known_feats.device):
            check(entry(B, n, m, C2, C1, kf.data_ptr(), ld_k,
                                                 uf.data_ptr() if uf is not None else None, ld_u, idx.data_ptr(),
                                                 weight.data_ptr(), packed.n_layers, packed.dims_c, w2, b2, meta,
                                                 ka.data_ptr(), ua.data_ptr() if ua is not None else None, out.data_ptr(),
                                                 1 if point_major_out else 0, ld_out,
                                                 amax.data_ptr() if amax is not None else None, rinv.data_ptr(),
                                                 _mlp_flags(packed), _stream(known_feats)),
                  "fp_interp_mlp_split2")
        if point_major_out:
            return seed_absmax(out[:, :, :M].transpose(1, 2), B * n, M, ld_out, amax)
        return out
    if (vec and _fused_mlp.split_arith()
            and lib.pvn3d_mlp_split_ok(0, C2, C1, 0, packed.n_layers, packed.dims_c)):
        with on_device(known_feats.device):
            check(lib.pvn3d_fp_interp_mlp_split(B, n, m, C2, C1, kf.data_ptr(), ld_k,
                                                uf.data_ptr() if uf is not None else None, ld_u, idx.data_ptr(),
                                                weight.data_ptr(), packed.n_layers, packed.dims_c, packed.split(),
                                                packed.b_c, out.data_ptr(), 1 if point_major_out else 0, ld_out,
                                                _stream(known_feats)), "fp_interp_mlp_split")
        return out[:, :, :M].transpose(1, 2) if point_major_out else out
    if (point_major_out and fp_layerwise_shape_ok(B * n, C1, packed.dims)
            and ld_k % 4 == 0 and kf.data_ptr() % 16 == 0 and ld_u % 4 == 0 and uf.data_ptr() % 16 == 0):
        amax = _fp_layerwise_split(B, n, m, C2, C1, kf, ld_k, uf, ld_u, idx, weight, packed, out, ld_out)
        view = out[:, :, :M].transpose(1, 2)
        return seed_absmax(view, B * n, M, ld_out, amax) if amax is not None else view
    if packed._equil_given is not None:
        raise RuntimeError("a rescaled (fp16 x 2) chain on a kernel that does not undo the rescaling")
    with on_device(known_feats.device):
        check(lib.pvn3d_fp_interp_mlp(B, n, m, C2, C1, kf.data_ptr(), ld_k,
                                      uf.data_ptr() if uf is not None else None, ld_u,
                                      idx.data_ptr(), weight.data_ptr(), packed.n_layers,
                                      packed.dims_c, packed.w_c, packed.b_c, out.data_ptr(),
                                      1 if point_major_out else 0, ld_out,
                                      _stream(known_feats)), "fp_interp_mlp")
    return out[:, :, :M].transpose(1, 2) if point_major_out else out
