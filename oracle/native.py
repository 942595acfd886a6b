"""ctypes binding of oracle/libpvn3d_oracle.so (see pvn3d_oracle.c for reference citations).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  All arrays are numpy, C-contiguous,
float32 / int32, shaped exactly like the reference's tensors.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpvn3d_oracle.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)
_u8 = ctypes.POINTER(ctypes.c_uint8)
_d = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile the C oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "pvn3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _fa(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return x, x.ctypes.data_as(_f)


def _ia(x):
    x = np.ascontiguousarray(x, dtype=np.int32)
    return x, x.ctypes.data_as(_i)


def opt_n_threads(w):
    return int(lib().orc_opt_n_threads(int(w)))


def furthest_point_sampling(xyz, npoint):
    xyz, p = _fa(xyz)
    b, n, _ = xyz.shape
    temp = np.empty((b, n), np.float32)
    out = np.zeros((b, npoint), np.int32)
    lib().orc_furthest_point_sampling(b, n, npoint, p, temp.ctypes.data_as(_f), out.ctypes.data_as(_i))
    return out


def gather_points(points, idx):
    points, pp = _fa(points)
    idx, pi = _ia(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib().orc_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(_f))
    return out


def gather_points_grad(grad_out, idx, n):
    grad_out, pg = _fa(grad_out)
    idx, pi = _ia(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().orc_gather_points_grad(b, c, n, m, pg, pi, out.ctypes.data_as(_f))
    return out


def ball_query(new_xyz, xyz, radius, nsample):
    new_xyz, pn = _fa(new_xyz)
    xyz, px = _fa(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    out = np.zeros((b, m, nsample), np.int32)
    lib().orc_ball_query(b, n, m, ctypes.c_float(radius), nsample, pn, px, out.ctypes.data_as(_i))
    return out


def group_points(points, idx):
    points, pp = _fa(points)
    idx, pi = _ia(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib().orc_group_points(b, c, n, npoints, nsample, pp, pi, out.ctypes.data_as(_f))
    return out


def group_points_grad(grad_out, idx, n):
    grad_out, pg = _fa(grad_out)
    idx, pi = _ia(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib().orc_group_points_grad(b, c, n, npoints, nsample, pg, pi, out.ctypes.data_as(_f))
    return out


def three_nn(unknown, known):
    unknown, pu = _fa(unknown)
    known, pk = _fa(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib().orc_three_nn(b, n, m, pu, pk, dist2.ctypes.data_as(_f), idx.ctypes.data_as(_i))
    return dist2, idx


def three_interpolate(points, idx, weight):
    points, pp = _fa(points)
    idx, pi = _ia(idx)
    weight, pw = _fa(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib().orc_three_interpolate(b, c, m, n, pp, pi, pw, out.ctypes.data_as(_f))
    return out


def three_interpolate_grad(grad_out, idx, weight, m, refbug=False):
    grad_out, pg = _fa(grad_out)
    idx, pi = _ia(idx)
    weight, pw = _fa(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    fn = lib().orc_three_interpolate_grad_refbug if refbug else lib().orc_three_interpolate_grad
    fn(b, c, n, m, pg, pi, pw, out.ctypes.data_as(_f))
    return out


def meanshift_fit(A, bandwidth, max_iter=300, return_all=False):
    """MeanShiftTorch(bandwidth, max_iter).fit(A) -> (ctr[3], labels[n] bool, iters).
    return_all adds the final seed positions and the last iteration's max shift."""
    A, pa = _fa(A)
    n = A.shape[0]
    ctr = np.zeros(3, np.float32)
    labels = np.zeros(n, np.uint8)
    iters = ctypes.c_int(0)
    cfin = np.zeros((n, 3), np.float32)
    last = ctypes.c_float(0)
    lib().orc_meanshift_fit(pa, n, ctypes.c_float(bandwidth), int(max_iter), ctr.ctypes.data_as(_f),
                            labels.ctypes.data_as(_u8), ctypes.byref(iters), cfin.ctypes.data_as(_f),
                            ctypes.byref(last))
    if return_all:
        return ctr, labels.astype(bool), iters.value, cfin, float(last.value)
    return ctr, labels.astype(bool), iters.value


def best_fit_transform(A, B):
    """basic_utils.best_fit_transform(A,B) -> (3,4) float64."""
    A, pa = _fa(A)
    B, pb = _fa(B)
    assert A.shape == B.shape
    T = np.zeros((3, 4), np.float64)
    lib().orc_best_fit_transform(pa, pb, A.shape[0], T.ctypes.data_as(_d))
    return T


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(t):
    lib().orc_set_num_threads(int(t))
