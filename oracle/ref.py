"""ctypes binding of oracle/_ref/libpvn3d_ref_{nofma,fma}.so = the REFERENCE'S OWN native-op kernels
(`pvn3d/_ext-src/src/*_gpu.cu`) compiled for the CPU by oracle/ref_shim/build_ref.py.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Same call interface as oracle/native.py, so a
test can run `native.f(...)` and `ref.f(...)` on the same arrays.

What is the reference here and what is restated:
  * the kernels and their `*_kernel_wrapper` launch code (grid/block shapes through the reference's own
    opt_n_threads / opt_block_config) are the reference's sources, unmodified except for the launch
    syntax (build_ref.py);
  * the ATen host functions around them cannot be compiled (removed ATen API), so the three facts they
    add are restated below with their citation: outputs start as zeros (`torch::zeros`, e.g.
    ball_query.cpp:19-21), the FPS scratch starts at 1e10 (sampling.cpp:73-75), and
    three_interpolate_grad calls the FORWARD wrapper with swapped sizes (interpolate.cpp:89-93).

`variant` = "nofma" (g++ -ffp-contract=off) or "fma" (g++ -ffp-contract=fast -mfma).
The libraries are built in the development container (where /root/reference exists) and travel to
the GPU box as prebuilt files; `available()` says whether they are there.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
_libs = {}

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)


def _path(variant):
    return os.path.join(_DIR, "libpvn3d_ref_%s.so" % variant)


def build(force=False):
    """Run the recipe if the reference tree is present (no-op otherwise)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_ref", os.path.join(_HERE, "ref_shim", "build_ref.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force)


def available(variant="nofma"):
    if not os.path.exists(_path(variant)):
        try:
            build()
        except Exception:  # no compiler / no reference: simply not available
            return False
    return os.path.exists(_path(variant))


def lib(variant="nofma"):
    if variant not in _libs:
        if not available(variant):
            raise RuntimeError("oracle/_ref is not built (needs /root/reference + g++): %s" % _path(variant))
        _libs[variant] = ctypes.CDLL(_path(variant))
    return _libs[variant]


def _fa(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return x, x.ctypes.data_as(_f)


def _ia(x):
    x = np.ascontiguousarray(x, dtype=np.int32)
    return x, x.ctypes.data_as(_i)


def opt_n_threads(w, variant="nofma"):
    return int(lib(variant).ref_opt_n_threads(int(w)))


def opt_block_config(x, y, variant="nofma"):
    out = (ctypes.c_int * 2)()
    lib(variant).ref_opt_block_config(int(x), int(y), out)
    return out[0], out[1]


def furthest_point_sampling(xyz, npoint, variant="nofma"):
    xyz, p = _fa(xyz)
    b, n, _ = xyz.shape
    temp = np.full((b, n), 1e10, np.float32)        # sampling.cpp:73-75
    out = np.zeros((b, npoint), np.int32)           # sampling.cpp:69-71
    lib(variant).ref_furthest_point_sampling(b, n, npoint, p, temp.ctypes.data_as(_f), out.ctypes.data_as(_i))
    return out


def gather_points(points, idx, variant="nofma"):
    points, pp = _fa(points)
    idx, pi = _ia(idx)
    b, c, n = points.shape
    m = idx.shape[1]
    out = np.zeros((b, c, m), np.float32)
    lib(variant).ref_gather_points(b, c, n, m, pp, pi, out.ctypes.data_as(_f))
    return out


def gather_points_grad(grad_out, idx, n, variant="nofma"):
    grad_out, pg = _fa(grad_out)
    idx, pi = _ia(idx)
    b, c, m = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib(variant).ref_gather_points_grad(b, c, n, m, pg, pi, out.ctypes.data_as(_f))
    return out


def ball_query(new_xyz, xyz, radius, nsample, variant="nofma"):
    new_xyz, pn = _fa(new_xyz)
    xyz, px = _fa(xyz)
    b, m, _ = new_xyz.shape
    n = xyz.shape[1]
    out = np.zeros((b, m, nsample), np.int32)       # ball_query.cpp:19-21
    lib(variant).ref_query_ball_point(b, n, m, ctypes.c_float(radius), nsample, pn, px, out.ctypes.data_as(_i))
    return out


def group_points(points, idx, variant="nofma"):
    points, pp = _fa(points)
    idx, pi = _ia(idx)
    b, c, n = points.shape
    _, npoints, nsample = idx.shape
    out = np.zeros((b, c, npoints, nsample), np.float32)
    lib(variant).ref_group_points(b, c, n, npoints, nsample, pp, pi, out.ctypes.data_as(_f))
    return out


def group_points_grad(grad_out, idx, n, variant="nofma"):
    grad_out, pg = _fa(grad_out)
    idx, pi = _ia(idx)
    b, c, npoints, nsample = grad_out.shape
    out = np.zeros((b, c, n), np.float32)
    lib(variant).ref_group_points_grad(b, c, n, npoints, nsample, pg, pi, out.ctypes.data_as(_f))
    return out


def three_nn(unknown, known, variant="nofma"):
    unknown, pu = _fa(unknown)
    known, pk = _fa(known)
    b, n, _ = unknown.shape
    m = known.shape[1]
    dist2 = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    lib(variant).ref_three_nn(b, n, m, pu, pk, dist2.ctypes.data_as(_f), idx.ctypes.data_as(_i))
    return dist2, idx


def three_interpolate(points, idx, weight, variant="nofma"):
    points, pp = _fa(points)
    idx, pi = _ia(idx)
    weight, pw = _fa(weight)
    b, c, m = points.shape
    n = idx.shape[1]
    out = np.zeros((b, c, n), np.float32)
    lib(variant).ref_three_interpolate(b, c, m, n, pp, pi, pw, out.ctypes.data_as(_f))
    return out


def three_interpolate_grad(grad_out, idx, weight, m, refbug=False, variant="nofma"):
    """refbug=False: the gradient kernel the reference defines (interpolate_gpu.cu:116-154);
    refbug=True: what the reference's binding actually calls (interpolate.cpp:89-93): the forward
    wrapper with (m := n, n := m)."""
    grad_out, pg = _fa(grad_out)
    idx, pi = _ia(idx)
    weight, pw = _fa(weight)
    b, c, n = grad_out.shape
    out = np.zeros((b, c, m), np.float32)
    if refbug:
        lib(variant).ref_three_interpolate(b, c, n, m, pg, pi, pw, out.ctypes.data_as(_f))
    else:
        lib(variant).ref_three_interpolate_grad(b, c, n, m, pg, pi, pw, out.ctypes.data_as(_f))
    return out
