"""Hot-path subset of the reference's pvn3d/lib/utils/basic_utils.py.

  best_fit_transform(A, B)      Kabsch fit, (3,4) float64 [R|t]           (reference :47-80)
  Basic_Utils.get_kps / get_ctr per-object keypoint fixtures               (:541-595)
  Basic_Utils.best_fit_transform (method alias)                           (:671-672)
  VOCap, Basic_Utils.cal_auc / cal_add_cuda / cal_adds_cuda  pose metrics  (:32-44, :597-635)
  Basic_Utils.get_pointxyz(_cuda)  mesh points of an object (dataset files) (:497-539)

best_fit_transform keeps the reference's numpy-in / numpy-out signature but runs the fit on the
GPU (pvn3d_amd/csrc/pose.hip); the batched, sync-free form used by cal_frame_poses* is
``_vote_engine.best_fit_transform_batch``; ADD / ADD-S run on the GPU (csrc/metrics.hip, batched form
``_vote_engine.add_adds_batch``).  Depth/cloud helpers and drawing are out of scope (SURVEY.md
section 2, row 11).
"""
import numpy as np
import torch

from . import _vote_engine as _eng
from ... import synth as _synth

LM_OBJ_DICT = {'ape': 1, 'benchvise': 2, 'cam': 4, 'can': 5, 'cat': 6, 'driller': 8, 'duck': 9,
               'eggbox': 10, 'glue': 11, 'holepuncher': 12, 'iron': 13, 'lamp': 14, 'phone': 15}
LM_ID2OBJ = dict((v, k) for k, v in LM_OBJ_DICT.items())


def _accuracy_curve_area(dists, n_total, max_dis=0.1):
    """Area under the accuracy-vs-threshold step curve on [0, 0.1] m, scaled to 1 (what the reference's
    VOCap returns for the sorted in-range distances and the running fraction k / n, basic_utils.py:30-42):
    the curve steps to k / n_total at the k-th smallest distance and is held to the end of the range."""
    d = np.sort(np.asarray(dists, dtype=np.float64))
    d = d[d <= max_dis]
    k = d.size
    if k == 0:
        return 0.0
    edges = np.concatenate(([0.0], d, [0.1]))
    frac = (np.arange(1, k + 1, dtype=np.float32) / np.float32(n_total)).astype(np.float64)   # fp32 like the reference
    height = np.concatenate((frac, frac[-1:]))
    return float(np.sum(np.diff(edges) * height) * 10.0)


def VOCap(rec, prec):
    """Reference signature kept (basic_utils.py:30): rec = sorted distances with out-of-range ones set to inf,
    prec = (1..n) / n."""
    rec = np.asarray(rec, dtype=np.float64)
    return _accuracy_curve_area(rec[np.isfinite(rec)], len(prec))


def read_ply_vertices(path):
    """(n, 3) float64 vertex coordinates of a PLY file (ascii or binary little/big endian); what the
    reference's `ply_vtx` (plyfile-based, basic_utils.py:483-495) returns."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s is not a PLY file" % path)
        fmt, n_vtx, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: unterminated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok:
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    n_vtx = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError("%s: list property inside the vertex element" % path)
                props.append((tok[2], tok[1]))
            elif tok[0] == "end_header":
                break
        names = [n for n, _ in props]
        if not all(c in names for c in "xyz"):
            raise ValueError("%s: vertex element without x / y / z" % path)
        if fmt == "ascii":
            rows = np.loadtxt(f, max_rows=n_vtx, ndmin=2)
            return np.stack([rows[:, names.index(c)] for c in "xyz"], 1).astype(np.float64)
        code = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
                "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
                "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}
        end = "<" if fmt == "binary_little_endian" else ">"
        rec = np.dtype([(n, end + code[t]) for n, t in props])
        data = np.frombuffer(f.read(rec.itemsize * n_vtx), dtype=rec, count=n_vtx)
        return np.stack([data[c].astype(np.float64) for c in "xyz"], 1)


def best_fit_transform(A, B):
    """Least-squares rigid transform mapping A (N,3) onto B (N,3): (3,4) float64 [R|t]."""
    assert A.shape == B.shape
    dev = torch.device("cuda", torch.cuda.current_device())
    a = torch.as_tensor(np.ascontiguousarray(A, dtype=np.float32), device=dev).unsqueeze(0)
    b = torch.as_tensor(np.ascontiguousarray(B, dtype=np.float32), device=dev).unsqueeze(0)
    return _eng.best_fit_transform_batch(a, b)[0].cpu().numpy()


class Basic_Utils(object):
    """Fixture loaders + pose fit of the reference's Basic_Utils (constructed there with a
    Config; here the object tables are bundled, see tools/import_obj_kps.py)."""

    def __init__(self, config=None):
        self.config = config
        z = _synth.obj_kps()
        self.ycb_cls_lst = [str(c) for c in z["ycb_classes"]]
        self.ycb_r_lst = list(z["ycb_radius"])

    def _name(self, cls, ds_type):
        if isinstance(cls, (int, np.integer)):
            return self.ycb_cls_lst[cls - 1] if ds_type == "ycb" else LM_ID2OBJ[int(cls)]
        return cls

    def get_kps(self, cls, kp_type='farthest', ds_type='ycb', kp_pth=None):
        if kp_pth:
            return np.loadtxt(kp_pth)
        if kp_type != 'farthest':
            raise NotImplementedError("only the 8 'farthest' keypoints are bundled")
        key = "%s/%s/farthest" % ("ycb" if ds_type == "ycb" else "lm", self._name(cls, ds_type))
        return _synth.obj_kps()[key].astype(np.float32).copy()

    def get_ctr(self, cls, ds_type='ycb', ctr_pth=None):
        if ctr_pth:
            return np.loadtxt(ctr_pth)
        key = "%s/%s/corners" % ("ycb" if ds_type == "ycb" else "lm", self._name(cls, ds_type))
        return _synth.obj_kps()[key].astype(np.float32).mean(0)

    def cal_auc(self, add_dis, max_dis=0.1):
        """AUC of the ADD(-S) accuracy curve, in percent (reference :597-605)."""
        return _accuracy_curve_area(add_dis, len(add_dis), max_dis) * 100

    def cal_add_cuda(self, pred_RT, gt_RT, p3ds):
        """mean_k |pred(x_k) - gt(x_k)| as a 0-dim CUDA tensor (reference :617-623)."""
        add, _ = _eng.add_adds_batch(p3ds, pred_RT.reshape(1, 3, 4), gt_RT.reshape(1, 3, 4))
        return add[0]

    def cal_adds_cuda(self, pred_RT, gt_RT, p3ds):
        """mean_k min_j |pred(x_j) - gt(x_k)| as a 0-dim CUDA tensor (reference :625-635)."""
        _, adds = _eng.add_adds_batch(p3ds, pred_RT.reshape(1, 3, 4), gt_RT.reshape(1, 3, 4))
        return adds[0]

    def get_pointxyz(self, cls, ds_type='ycb'):
        """Mesh points of an object from the dataset tree, as the reference reads them (:497-521):
        YCB `<config.ycb_root>/models/<cls>/points.xyz`; LineMOD
        `datasets/linemod/Linemod_preprocessed/models/obj_%02d.ply` (relative to the working directory, like
        the reference; `config.lm_root` overrides the directory), millimetres -> metres, sub-sampled to 2000
        vertices with `random.sample` as at :514-516.  ``set_pointxyz`` supplies points directly (the
        datasets are not part of this repository).  Raises FileNotFoundError when the file is missing."""
        import os
        key = (ds_type, self._name(cls, ds_type) if ds_type == "ycb" else int(cls))
        cache = self.__dict__.setdefault("_ptsxyz", {})
        if key in cache:
            return cache[key]
        if ds_type == "ycb":
            root = getattr(self.config, "ycb_root", None) if self.config is not None else None
            if root is None:
                raise FileNotFoundError("YCB mesh points of %s: config.ycb_root is not set (or call set_pointxyz())" % (key[1],))
            pts = np.loadtxt(os.path.join(root, "models", "%s/points.xyz" % key[1]), dtype=np.float32)
        else:
            import random
            root = getattr(self.config, "lm_root", None) if self.config is not None else None
            pth = os.path.join(root or "datasets/linemod/Linemod_preprocessed", "models", "obj_%02d.ply" % int(cls))
            if not os.path.isfile(pth):
                raise FileNotFoundError("LineMOD mesh %s not found (or call set_pointxyz())" % pth)
            pts = read_ply_vertices(pth) / 1000.0
            if len(pts) > 2000:
                drop = random.sample(range(len(pts)), len(pts) - 2000)
                pts = np.delete(pts, drop, axis=0)
            pts = pts.astype(np.float32)
        cache[key] = pts
        return pts

    def set_pointxyz(self, cls, pts, ds_type='ycb'):
        key = (ds_type, self._name(cls, ds_type) if ds_type == "ycb" else int(cls))
        self.__dict__.setdefault("_ptsxyz", {})[key] = np.asarray(pts, dtype=np.float32)
        self.__dict__.setdefault("_ptsxyz_cuda", {}).pop(key, None)

    def get_pointxyz_cuda(self, cls, ds_type='ycb'):
        key = (ds_type, self._name(cls, ds_type) if ds_type == "ycb" else int(cls))
        cache = self.__dict__.setdefault("_ptsxyz_cuda", {})
        if key not in cache:
            cache[key] = torch.from_numpy(self.get_pointxyz(cls, ds_type).astype(np.float32)).cuda()
        return cache[key].clone()

    def best_fit_transform(self, A, B):
        return best_fit_transform(A, B)
