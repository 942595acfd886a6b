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


def VOCap(rec, prec):
    """Area under the accuracy-threshold curve up to 0.1 m (reference :32-44, verbatim semantics)."""
    idx = np.where(rec != np.inf)
    if len(idx[0]) == 0:
        return 0
    rec = rec[idx]
    prec = prec[idx]
    mrec = np.array([0.0] + list(rec) + [0.1])
    mpre = np.array([0.0] + list(prec) + [prec[-1]])
    for i in range(1, prec.shape[0]):
        mpre[i] = max(mpre[i], mpre[i - 1])
    i = np.where(mrec[1:] != mrec[0:-1])[0] + 1
    ap = np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10
    return ap


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
        D = np.array(add_dis)
        D[np.where(D > max_dis)] = np.inf
        D = np.sort(D)
        n = len(add_dis)
        acc = np.cumsum(np.ones((1, n)), dtype=np.float32) / n
        return VOCap(D, acc) * 100

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
        YCB `<ycb_root>/models/<cls>/points.xyz`; LineMOD `obj_%02d.ply` sub-sampled to 2000 points
        is NOT reproduced (it needs plyfile and Python's `random`): pass LineMOD points through
        ``set_pointxyz``.  The datasets are not part of this repository."""
        key = (ds_type, self._name(cls, ds_type) if ds_type == "ycb" else int(cls))
        cache = self.__dict__.setdefault("_ptsxyz", {})
        if key in cache:
            return cache[key]
        if ds_type != "ycb":
            raise FileNotFoundError("LineMOD mesh points for object %s: call set_pointxyz() first" % (cls,))
        root = getattr(self.config, "ycb_root", None) if self.config is not None else None
        if root is None:
            raise FileNotFoundError("config.ycb_root is not set; call set_pointxyz() or pass a config")
        import os
        pts = np.loadtxt(os.path.join(root, "models", "%s/points.xyz" % key[1]), dtype=np.float32)
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
