"""ctypes loader for libpvn3d_hip.so -- the C-ABI boundary (include/pvn3d_hip.h).

Fails loudly: if the shared library is missing or does not export a declared symbol, importing
this module raises.  There is deliberately no fallback path.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PVN3D_HIP_LIB: alternative build of the same library (kernel A/B experiments)
LIB_PATH = os.environ.get("PVN3D_HIP_LIB") or os.path.join(_HERE, "libpvn3d_hip.so")

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t
_ll = ctypes.c_longlong
_d = ctypes.c_double

# name -> (restype, argtypes); one entry per declaration in include/pvn3d_hip.h
SIGNATURES = {
    "pvn3d_abi_version": (_i, []),
    "pvn3d_opt_n_threads": (_i, [_i]),
    "pvn3d_furthest_point_sampling": (_i, [_i, _i, _i, _p, _p, _p, _p]),
    "pvn3d_furthest_point_sampling_nested": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "pvn3d_fps_nest_verify": (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    "pvn3d_fps_ws_words": (_i, [_i]),
    "pvn3d_furthest_point_sampling_ws": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _p]),
    "pvn3d_furthest_point_sampling_ws_waves": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p]),
    "pvn3d_gather_points": (_i, [_i, _i, _i, _i, _p, _p, _p, _p]),
    "pvn3d_gather_points_grad": (_i, [_i, _i, _i, _i, _p, _p, _p, _p]),
    "pvn3d_ball_query": (_i, [_i, _i, _i, _f, _i, _p, _p, _p, _p]),
    "pvn3d_group_points": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "pvn3d_group_points_grad": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "pvn3d_three_nn": (_i, [_i, _i, _i, _p, _p, _p, _p, _p]),
    "pvn3d_three_interpolate": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "pvn3d_three_interpolate_grad": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _i, _p]),
    "pvn3d_ball_query_pair": (_i, [_i, _i, _i, _f, _i, _f, _i, _p, _p, _p, _p, _p]),
    "pvn3d_ball_query_grid_workspace_bytes": (_sz, [_i, _i]),
    "pvn3d_ball_query_pair_grid": (_i, [_i, _i, _i, _f, _i, _f, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "pvn3d_group_xyz_features": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "pvn3d_group_xyz_features_pair": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_sa_mlp_maxpool": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "pvn3d_fp_interp_mlp": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "pvn3d_mlp_split_ok": (_i, [_i, _i, _i, _i, _i, _p]),
    "pvn3d_sa_mlp_maxpool_split": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "pvn3d_fp_interp_mlp_split": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _i, _p]),
    "pvn3d_mlp_split2_ok": (_i, [_i, _i, _i, _i, _i, _p, _i]),
    "pvn3d_mlp_split2_kernel": (_i, [_i, _i, _i, _i, _i, _p, _i, _i]),
    "pvn3d_sa_mlp_maxpool_split2": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "pvn3d_fp_interp_mlp_split2": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "pvn3d_fp_interp_add_mlp_split2": (_i, [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _i, _p]),
    "pvn3d_absmax": (_i, [ctypes.c_longlong, _i, _p, _i, _p, _p]),
    "pvn3d_split_rows2": (_i, [ctypes.c_longlong, _i, _p, _i, _p, _p, _i, _p]),
    "pvn3d_split_gemm2": (_i, [_i, _i, _i, _p, _p, _p, _f, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _p]),
    "pvn3d_split_gemm2_tile128": (_i, [_i, _i, _i, _p, _p, _p, _f, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _i, _p, _p]),
    "pvn3d_bound_affine": (_i, [_p, _p, _f, _p, _f, _f, _p]),
    "pvn3d_split_rows": (_i, [ctypes.c_longlong, _i, _p, _i, _p, _i, _p]),
    "pvn3d_split_gemm": (_i, [_i, _i, _i, _p, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p]),
    "pvn3d_transpose_bcn_to_bnc": (_i, [_i, _i, _i, _p, _p, _i, _p]),
    "pvn3d_meanshift_workspace_bytes": (_sz, [_i, _i, _i]),
    "pvn3d_meanshift_fit_batch": (_i, [_p, _p, _p, _i, _i, _i, _f, _i, _p, _p, _p, _p, _sz, _p, _i, _i, _p]),
    "pvn3d_vote_compact": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p]),
    "pvn3d_vote_compact_strided": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p]),
    "pvn3d_best_fit_transform": (_i, [_i, _i, _p, _p, _p, _p, _p]),
    "pvn3d_three_nn_weights": (_i, [ctypes.c_longlong, _p, _p, _p]),
    "pvn3d_three_nn_grid_workspace_bytes": (_sz, [_i, _i]),
    "pvn3d_three_nn_grid": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "pvn3d_scatter_det_workspace_bytes": (_sz, [_i, _i, _i]),
    "pvn3d_group_points_grad_det": (_i, [_i, _i, _i, _i, _i, _p, _p, _p, _p, _sz, _p]),
    "pvn3d_three_interpolate_grad_det": (_i, [_i, _i, _i, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "pvn3d_of_l1_loss": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "pvn3d_of_l1_loss_grad": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_relabel_by_centre": (_i, [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_add_adds_workspace_bytes": (_sz, [_i, _i]),
    "pvn3d_add_adds_batch": (_i, [_i, _i, _p, _p, _p, _p, _p, _sz, _p, _p, _p]),
    # layer-by-layer fp32-MFMA SharedMLP for small launches (csrc/small_batch.hip)
    "pvn3d_sb_linear": (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p, _i, _p]),
    "pvn3d_sb_linear_splits": (_i, [_i, _i, _i]),
    "pvn3d_sb_gather_sa": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _ll, _ll, _ll, _p, _p, _i, _p]),
    "pvn3d_sb_gather_fp": (_i, [_i, _i, _i, _i, _i, _p, _ll, _ll, _ll, _p, _ll, _ll, _ll, _p, _p, _p, _i, _p]),
    "pvn3d_sb_pool_max": (_i, [_ll, _i, _i, _i, _p, _p, _ll, _p]),
    # training-mode SharedMLP on bf16 MFMA (csrc/mlp_train.hip)
    "pvn3d_mt_gemm_nt": (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _p, _i, _p]),
    "pvn3d_mt_gemm_nt_stat_rows": (_i, [_i]),
    "pvn3d_mt_gemm_nt_splitk": (_i, [_i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p]),
    "pvn3d_mt_transpose": (_i, [_ll, _i, _p, _p, _ll, _p]),
    "pvn3d_mt_wgrad_tn_ok": (_i, [_i, _i]),
    "pvn3d_mt_wgrad_tn": (_i, [_ll, _i, _i, _p, _i, _p, _i, _p, _i, _p]),
    "pvn3d_mt_pack_weight": (_i, [_i, _i, _p, _i, _i, _p, _i, _i, _p]),
    "pvn3d_mt_gather_sa": (_i, [_i, _i, _i, _i, _i, _i, _p, _p, _p, _ll, _ll, _ll, _p, _p, _i, _p]),
    "pvn3d_mt_unpack_cm": (_i, [_i, _i, _i, _i, _i, _p, _p, _p]),
    "pvn3d_mt_csr_build": (_i, [_i, _i, _i, _p, _p, _p, _p]),
    "pvn3d_mt_inv_gather": (_i, [_i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _p]),
    "pvn3d_mt_pack_cm": (_i, [_i, _i, _i, _i, _p, _p, _p]),
    "pvn3d_mt_gather_fp": (_i, [_i, _i, _i, _i, _i, _p, _ll, _ll, _ll, _p, _ll, _ll, _ll, _p, _p, _p, _i, _p]),
    "pvn3d_mt_bn_finalize": (_i, [_i, _i, _i, _d, _p, _p, _p, _p, _f, _f, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_mt_bn_relu_apply": (_i, [_ll, _i, _p, _p, _p, _p, _p]),
    "pvn3d_mt_pool_max": (_i, [_ll, _i, _i, _i, _p, _p, _ll, _p, _p]),
    "pvn3d_mt_pool_bwd": (_i, [_ll, _i, _i, _i, _p, _ll, _p, _p, _p]),
    "pvn3d_mt_pack_grad": (_i, [_ll, _i, _i, _p, _ll, _p, _p]),
    "pvn3d_mt_unpack_out": (_i, [_ll, _i, _i, _p, _p, _ll, _p]),
    "pvn3d_mt_bn_bwd_partials": (_i, [_ll]),
    "pvn3d_mt_bn_bwd_reduce": (_i, [_ll, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_mt_bn_relu_pool": (_i, [_ll, _i, _i, _i, _p, _p, _p, _p, _ll, _p, _p]),
    "pvn3d_mt_bn_bwd_reduce_pooled": (_i, [_ll, _i, _i, _i, _p, _ll, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_mt_bn_bwd_apply_pooled": (_i, [_ll, _i, _i, _i, _p, _ll, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_mt_bn_bwd_finalize": (_i, [_i, _i, _i, _d, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "pvn3d_mt_bn_bwd_apply": (_i, [_ll, _i, _p, _p, _p, _p, _p, _p, _p, _p]),
}

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "pvn3d_amd: %s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'`"
        " or `make -C pvn3d_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == ABI mismatch, by design
    _fn.restype = _res
    _fn.argtypes = _args

if lib.pvn3d_abi_version() != 1:
    raise ImportError("pvn3d_amd: libpvn3d_hip.so ABI version mismatch")


def check(rc, what):
    """Non-zero return (a hipError_t) -> RuntimeError, like AT_CHECK in the reference glue."""
    if rc != 0:
        raise RuntimeError("%s failed: hipError %d" % (what, rc))


import contextlib as _contextlib

_NO_SWITCH = _contextlib.nullcontext()


def on_device(dev):
    """Context for a library call on `dev`: torch.cuda.device(dev) only when `dev` is not already the
    current device (entering / leaving that context costs two driver calls, ~10 us of host time per
    operator -- more than the short kernels of the small pyramid levels take)."""
    import torch
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(dev)
