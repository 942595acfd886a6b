"""Hot-path subset of the reference's pvn3d/lib/utils/basic_utils.py.

  best_fit_transform(A, B)      Kabsch fit, (3,4) float64 [R|t]           (reference :47-80)
  Basic_Utils.get_kps / get_ctr per-object keypoint fixtures               (:541-595)
  Basic_Utils.best_fit_transform (method alias)                           (:671-672)

best_fit_transform keeps the reference's numpy-in / numpy-out signature but runs the fit on the
GPU (pvn3d_amd/csrc/pose.hip); the batched, sync-free form used by cal_frame_poses* is
``_vote_engine.best_fit_transform_batch``.  Depth/cloud helpers, metrics and drawing are out of
scope.
"""
import numpy as np
import torch

from . import _vote_engine as _eng
from ... import synth as _synth

LM_OBJ_DICT = {'ape': 1, 'benchvise': 2, 'cam': 4, 'can': 5, 'cat': 6, 'driller': 8, 'duck': 9,
               'eggbox': 10, 'glue': 11, 'holepuncher': 12, 'iron': 13, 'lamp': 14, 'phone': 15}
LM_ID2OBJ = dict((v, k) for k, v in LM_OBJ_DICT.items())


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

    def best_fit_transform(self, A, B):
        return best_fit_transform(A, B)
