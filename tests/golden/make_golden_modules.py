#!/usr/bin/env python3
"""Generate tests/golden/pointnet2msg_ref.npz by running the REFERENCE'S OWN module code.

What runs here (CPU, build container only -- needs /root/reference + g++):
  * pvn3d/lib/pointnet2_utils/pointnet2_modules.py   (_PointnetSAModuleBase.forward :27-71,
    PointnetSAModuleMSG :74-112, PointnetFPModule :143-206)                     file-level import, unmodified
  * pvn3d/lib/pointnet2_utils/pointnet2_utils.py      (the autograd Functions, QueryAndGroup)   file-level, unmodified
  * pvn3d/lib/utils/etw_pytorch_utils/pytorch_utils.py (SharedMLP / Conv2d / BatchNorm2d :25-134) file-level, unmodified
  * pvn3d/lib/pvn3d.py: class Pointnet2MSG (:46-154)  file-level import with `lib.pspnet` stubbed (the CNN is
    not on the path and needs torchvision)
  * `lib.pointnet2_utils._ext` is bound to a CPU adapter over oracle/_ref = the reference's own
    *_gpu.cu kernels compiled for the CPU (oracle/ref_shim/build_ref.py), with the three host-side facts of the
    ATen wrappers restated in oracle/ref.py.
So every number in the fixture is produced by reference code: its kernels, its Python ops, its modules.

Cases
  full : the reference's `Pointnet2MSG(input_channels=6)` (hyper-parameters lib/pvn3d.py:65-118), eval mode,
         one seeded N = 12 288 synthetic frame.  The state_dict is NOT stored (3.3 M floats); it is a pure
         function `weights(keys, shapes, seed)` of numpy's frozen legacy RandomState stream (module_weights.py,
         shared with the test); the fixture stores the key list, shapes and the SHA-256 of the bytes, and the
         generator loads it into the reference model with strict=True.  Stored per level (4 SA, 4 FP):
         the FPS indices, the two ball-query index tensors, three_nn idx, 32 seeded point columns of the
         output (all channels), and two fp64 checksum projections that cover every element (sum over points
         per channel, sum over channels per point) plus the matching sums of |x| (the scale of the tolerance).
         Also the same projections from a float64 run of the same module code (truth estimate: how far fp32
         Conv2d on the CPU itself is from exact arithmetic).
  small: one PointnetSAModuleMSG + one PointnetFPModule on B = 2 ragged-size clouds (N = 777, npoint = 100),
         eval AND train mode; everything stored in full, state_dict included; the train-mode case also stores
         the reference's autograd gradients (with the binding's three_interpolate_grad bug,
         interpolate.cpp:89-93, as the reference really runs, and with the gradient kernel the reference
         defines but never calls).

Usage: python tests/golden/make_golden_modules.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/pvn3d"
sys.path.insert(0, ROOT)

from oracle import ref as kref  # noqa: E402
from pvn3d_amd import synth  # noqa: E402
sys.path.insert(0, HERE)
from module_weights import weights, weights_sha  # noqa: E402

N_COLS = 32


# ------------------------------------------------------------------------------------------ _ext over oracle/_ref
class RefExt(object):
    """`lib.pointnet2_utils._ext` (bindings.cpp:6-19) on CPU tensors: the reference's kernels through oracle/ref.py.
    Index ops always run the fp32 kernels on the fp32 values of their inputs; in a float64 run the three
    value ops (gather / group / interpolate, which only copy or form 3-term sums) are evaluated in float64."""

    refbug = True     # three_interpolate_grad as the reference's binding really behaves (interpolate.cpp:89-93)

    @staticmethod
    def _np(t):
        return t.detach().contiguous().numpy()

    def furthest_point_sampling(self, xyz, npoint):
        return torch.from_numpy(kref.furthest_point_sampling(self._np(xyz.float()), npoint))

    def ball_query(self, new_xyz, xyz, radius, nsample):
        return torch.from_numpy(kref.ball_query(self._np(new_xyz.float()), self._np(xyz.float()), radius, nsample))

    def three_nn(self, unknown, known):
        d2, idx = kref.three_nn(self._np(unknown.float()), self._np(known.float()))
        return torch.from_numpy(d2).to(unknown.dtype), torch.from_numpy(idx)

    def gather_points(self, points, idx):
        if points.dtype == torch.float64:
            return torch.gather(points, 2, idx.long()[:, None, :].expand(-1, points.shape[1], -1)).contiguous()
        return torch.from_numpy(kref.gather_points(self._np(points), self._np(idx)))

    def gather_points_grad(self, grad_out, idx, n):
        return torch.from_numpy(kref.gather_points_grad(self._np(grad_out), self._np(idx), n))

    def group_points(self, points, idx):
        if points.dtype == torch.float64:
            b, c, n = points.shape
            _, m, s = idx.shape
            flat = idx.long().reshape(b, 1, m * s).expand(-1, c, -1)
            return torch.gather(points, 2, flat).reshape(b, c, m, s).contiguous()
        return torch.from_numpy(kref.group_points(self._np(points), self._np(idx)))

    def group_points_grad(self, grad_out, idx, n):
        return torch.from_numpy(kref.group_points_grad(self._np(grad_out), self._np(idx), n))

    def three_interpolate(self, points, idx, weight):
        if points.dtype == torch.float64:
            b, c, m = points.shape
            n = idx.shape[1]
            g = torch.gather(points, 2, idx.long().reshape(b, 1, n * 3).expand(-1, c, -1)).reshape(b, c, n, 3)
            return (g * weight[:, None]).sum(-1).contiguous()
        return torch.from_numpy(kref.three_interpolate(self._np(points), self._np(idx), self._np(weight)))

    def three_interpolate_grad(self, grad_out, idx, weight, m):
        return torch.from_numpy(kref.three_interpolate_grad(self._np(grad_out), self._np(idx), self._np(weight), m,
                                                            refbug=self.refbug))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_modules(ext):
    """-> (pointnet2_modules, pointnet2_utils, pvn3d) of the reference, wired to `ext`."""
    lib = types.ModuleType("lib"); lib.__path__ = []
    lu = types.ModuleType("lib.utils"); lu.__path__ = []
    lp = types.ModuleType("lib.pointnet2_utils"); lp.__path__ = []
    lp._ext = ext
    psp = types.ModuleType("lib.pspnet"); psp.PSPNet = None; psp.Modified_PSPNet = None
    sys.modules.update({"lib": lib, "lib.utils": lu, "lib.pointnet2_utils": lp, "lib.pointnet2_utils._ext": ext,
                        "lib.pspnet": psp})
    etw = _load("lib.utils.etw_pytorch_utils", os.path.join(REF, "lib/utils/etw_pytorch_utils/pytorch_utils.py"))
    lu.etw_pytorch_utils = etw
    pu = _load("lib.pointnet2_utils.pointnet2_utils", os.path.join(REF, "lib/pointnet2_utils/pointnet2_utils.py"))
    lp.pointnet2_utils = pu
    pm = _load("lib.pointnet2_utils.pointnet2_modules", os.path.join(REF, "lib/pointnet2_utils/pointnet2_modules.py"))
    lp.pointnet2_modules = pm
    net = _load("lib.pvn3d", os.path.join(REF, "lib/pvn3d.py"))
    return pm, pu, net


# deterministic weights: tests/golden/module_weights.py (shared with the tests that rebuild the state_dict)
def load_weights(model, seed):
    sd = model.state_dict()
    keys = list(sd.keys())
    shapes = [tuple(sd[k].shape) for k in keys]
    w = weights(keys, shapes, seed)
    model.load_state_dict({k: torch.from_numpy(np.asarray(w[k])) for k in keys}, strict=True)
    return keys, shapes, w


# ------------------------------------------------------------------------------------------ capture
class Capture(object):
    """Wraps ext functions to record the index tensors the reference modules computed, in call order."""

    def __init__(self, ext):
        self.ext = ext
        self.log = []
        for name in ("furthest_point_sampling", "ball_query", "three_nn"):
            setattr(self, name, self._wrap(name))
        for name in ("gather_points", "gather_points_grad", "group_points", "group_points_grad",
                     "three_interpolate", "three_interpolate_grad"):
            setattr(self, name, getattr(ext, name))

    def _wrap(self, name):
        f = getattr(self.ext, name)

        def g(*a):
            r = f(*a)
            self.log.append((name, r))
            return r
        return g


def projections(t):
    """t (B, C, n) -> checksum projections in float64."""
    t = t.detach().double()
    return dict(chan_sum=t.sum(2).numpy(), chan_abs=t.abs().sum(2).numpy(),
                pt_sum=t.sum(1).numpy(), pt_abs=t.abs().sum(1).numpy())


def run_full(pm, pu, net_mod, cap, out):
    torch.manual_seed(0)
    f = synth.synth_frame(frame=0, n_pts=12288, n_obj=3072)
    pc = np.concatenate([f["pcld"], f["feats"].T], 1).astype(np.float32)[None]      # (1, N, 9): xyz ++ 6 features
    model = net_mod.Pointnet2MSG(input_channels=6).eval()
    keys, shapes, w = load_weights(model, seed=4242)
    out["full_pc"] = pc
    out["full_keys"] = np.array(keys)
    out["full_shapes"] = np.array([",".join(map(str, s)) for s in shapes])
    out["full_seed"] = np.int64(4242)
    out["full_sha256"] = np.array(weights_sha(keys, w))

    rs = np.random.RandomState(7)
    for dtype, tag in ((torch.float32, ""), (torch.float64, "_f64")):
        feats = {}
        hooks = []
        for i, m in enumerate(model.SA_modules):
            hooks.append(m.register_forward_hook(lambda mod, a, r, i=i: feats.__setitem__("sa%d" % i, r[1])))
        for i, m in enumerate(model.FP_modules):
            hooks.append(m.register_forward_hook(lambda mod, a, r, i=i: feats.__setitem__("fp%d" % i, r)))
        cap.log.clear()
        mdl = model.double() if dtype == torch.float64 else model.float()
        with torch.no_grad():
            y = mdl(torch.from_numpy(pc).to(dtype))
        for h in hooks:
            h.remove()
        assert y.shape == (1, 128, 12288)
        if tag == "":
            names = iter(cap.log)
            lvl = {"furthest_point_sampling": 0, "ball_query": 0, "three_nn": 0}
            for name, r in names:
                k = lvl[name]
                lvl[name] += 1
                if name == "furthest_point_sampling":
                    out["full_fps%d" % k] = r[0].numpy().astype(np.int16)
                elif name == "ball_query":
                    out["full_bq%d_%d" % (k // 2, k % 2)] = r[0].numpy().astype(np.int16)
                else:       # FP modules run in reverse: call k=0 is FP_modules[3]
                    out["full_nn%d_idx" % (3 - k)] = r[1][0].numpy().astype(np.int16)
                    out["full_nn%d_d2" % (3 - k)] = r[0][0].numpy()
            assert lvl == {"furthest_point_sampling": 4, "ball_query": 8, "three_nn": 4}
        for name, t in sorted(feats.items()):
            pr = projections(t)
            for k, v in pr.items():
                out["full_%s_%s%s" % (name, k, tag)] = v[0]
            if tag == "":
                cols = np.sort(rs.choice(t.shape[2], size=min(N_COLS, t.shape[2]), replace=False))
                out["full_%s_cols" % name] = cols.astype(np.int32)
                out["full_%s_vals" % name] = t[0][:, cols].numpy()
            else:
                cols = out["full_%s_cols" % name]
                out["full_%s_vals_f64" % name] = t[0][:, cols].numpy().astype(np.float32)
        print("full%s done" % tag, flush=True)
    model.float()


def run_small(pm, pu, ext, out):
    rs = np.random.RandomState(99)
    B, N = 2, 777
    xyz = np.stack([synth.synth_frame(frame=10 + b, n_pts=N, n_obj=100)["pcld"] for b in range(B)]).astype(np.float32)
    feats = rs.standard_normal(size=(B, 5, N)).astype(np.float32)
    sa = pm.PointnetSAModuleMSG(npoint=100, radii=[0.03, 0.07], nsamples=[8, 32], mlps=[[5, 16, 24], [5, 8, 40]])
    fp = pm.PointnetFPModule(mlp=[64 + 5, 48, 32])
    for name, mod, seed in (("sa", sa, 11), ("fp", fp, 12)):
        keys, shapes, w = load_weights(mod, seed)
        out["small_%s_keys" % name] = np.array(keys)
        for k in keys:
            out["small_%s_w/%s" % (name, k)] = np.asarray(w[k])
    out["small_xyz"] = xyz
    out["small_feats"] = feats
    G = rs.standard_normal(size=(B, 32, N)).astype(np.float32)
    out["small_G"] = G

    def forward(xyz_t, feats_t):
        new_xyz, f1 = sa(xyz_t, feats_t)
        y = fp(xyz_t, new_xyz, feats_t, f1)
        return new_xyz, f1, y

    # eval
    sa.eval(); fp.eval()
    with torch.no_grad():
        new_xyz, f1, y = forward(torch.from_numpy(xyz), torch.from_numpy(feats))
    out["small_eval_new_xyz"] = new_xyz.numpy()
    out["small_eval_sa"] = f1.numpy()
    out["small_eval_fp"] = y.numpy()
    # train (batch statistics, running-stat update, autograd through the reference's Functions)
    for refbug in (True, False):
        tag = "refbug" if refbug else "fixed"
        for name, mod, seed in (("sa", sa, 11), ("fp", fp, 12)):
            load_weights(mod, seed)
            mod.train()
            mod.zero_grad()
        ext.refbug = refbug
        ft = torch.from_numpy(feats).clone().requires_grad_(True)
        new_xyz, f1, y = forward(torch.from_numpy(xyz), ft)
        (y * torch.from_numpy(G)).sum().backward()
        out["small_train_sa"] = f1.detach().numpy()
        out["small_train_fp"] = y.detach().numpy()
        out["small_train_%s_dfeats" % tag] = ft.grad.numpy()
        for name, mod in (("sa", sa), ("fp", fp)):
            for k, p in mod.named_parameters():
                out["small_train_%s_grad_%s/%s" % (tag, name, k)] = p.grad.numpy()
            if refbug:
                for k, b in mod.named_buffers():
                    out["small_train_buf_%s/%s" % (name, k)] = b.numpy().copy()
    ext.refbug = True
    print("small done", flush=True)


def main():
    assert kref.build(), "needs /root/reference"
    torch.set_num_threads(8)
    ext = RefExt()
    cap = Capture(ext)
    pm, pu, net_mod = load_reference_modules(cap)
    assert pu._ext is cap
    out = {}
    run_small(pm, pu, ext, out)
    run_full(pm, pu, net_mod, cap, out)
    path = os.path.join(HERE, "pointnet2msg_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote %s, %d KiB, %d arrays" % (path, os.path.getsize(path) // 1024, len(out)))


if __name__ == "__main__":
    main()
