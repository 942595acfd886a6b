"""Pose post-processing with the reference's call signatures
(pvn3d/lib/utils/pvn3d_eval_utils.py): per-point keypoint votes -> MeanShift clustering ->
least-squares pose.

  cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, obj_id)
      -> [pred_RT]                                   (reference :156-201)
  cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter)
      -> (pred_cls_ids, pred_pose_lst)               (reference :37-110)
  cal_batch_poses_lm / cal_batch_poses               batched, sync-free forms used by bench.py
  TorchEval.eval_pose_parallel                       pose part of reference :345-387

The reference launches ~12 torch kernels per mean-shift iteration per fit and synchronises
with the host every iteration; here each call is a handful of launches covering every fit of
every object (see _vote_engine.py).  ADD/ADD-S accumulation is out of scope.
"""
import numpy as np
import torch

from . import _vote_engine as _eng
from .basic_utils import Basic_Utils

_bs_utils = Basic_Utils()
RADIUS = 0.08  # hard-coded in the reference (:44, :163)


def _mesh_kps(cls, ds_type, use_ctr):
    kps = _bs_utils.get_kps(cls, ds_type=ds_type)
    if use_ctr:
        kps = np.concatenate((kps, _bs_utils.get_ctr(cls, ds_type=ds_type).reshape(1, 3)), axis=0)
    return torch.from_numpy(kps.astype(np.float32))


def cal_batch_poses_lm(pclds, masks, ctr_ofs, pred_kp_ofs, use_ctr, n_cls, use_ctr_clus_flter,
                       obj_id, poll_every=8):
    """Batched cal_frame_poses_lm: pclds (F,N,3), masks (F,N), ctr_ofs (F,1,N,3),
    pred_kp_ofs (F,K,N,3).  Returns the engine dict (poses (F,3,4) float64 on the device, ...)."""
    mesh = _mesh_kps(obj_id, "linemod", True)
    return _eng.frames_pose_single_class(pclds, masks, ctr_ofs, pred_kp_ofs, mesh, cls_id=1,
                                         use_ctr=use_ctr, use_ctr_clus_flter=use_ctr_clus_flter,
                                         radius=RADIUS, poll_every=poll_every)


def cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter, obj_id):
    """One LineMOD frame: pcld (N,3), mask (N), ctr_of (1,N,3), pred_kp_of (K,N,3) -> [3x4 ndarray]."""
    res = cal_batch_poses_lm(pcld.unsqueeze(0), mask.unsqueeze(0), ctr_of.unsqueeze(0),
                             pred_kp_of.unsqueeze(0), use_ctr, n_cls, use_ctr_clus_flter, obj_id)
    return [res["poses"][0].cpu().numpy()]


_YCB_MESH = {}


def _ycb_mesh_all(n_cls, use_ctr=True):
    key = (n_cls, use_ctr)
    if key not in _YCB_MESH:
        _YCB_MESH[key] = torch.stack([_mesh_kps(c, "ycb", True) for c in range(1, n_cls)], 0)
    return _YCB_MESH[key]


def cal_batch_poses(pclds, masks, ctr_ofs, pred_kp_ofs, use_ctr, n_cls, use_ctr_clus_flter,
                    poll_every=8):
    """Batched cal_frame_poses over F frames and class ids 1..n_cls-1 (engine dict)."""
    return _eng.frames_pose_multi_class(pclds, masks, ctr_ofs, pred_kp_ofs, _ycb_mesh_all(n_cls),
                                        n_cls, _bs_utils.ycb_r_lst, use_ctr=use_ctr,
                                        use_ctr_clus_flter=use_ctr_clus_flter, radius=RADIUS,
                                        poll_every=poll_every)


def cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter):
    """One YCB frame -> (pred_cls_ids ndarray, [3x4 ndarray per predicted class])."""
    res = cal_batch_poses(pcld.unsqueeze(0), mask.unsqueeze(0), ctr_of.unsqueeze(0),
                          pred_kp_of.unsqueeze(0), use_ctr, n_cls, use_ctr_clus_flter)
    present = res["present"][0].cpu().numpy()
    poses = res["poses"][0].cpu().numpy()
    pred_cls_ids = (np.nonzero(present)[0] + 1).astype(np.int64)
    return pred_cls_ids, [poses[c - 1] for c in pred_cls_ids]


class TorchEval(object):
    """Pose half of the reference's TorchEval (metrics accumulation is out of scope)."""

    def __init__(self, n_cls=22):
        self.n_cls = n_cls

    def eval_pose_parallel(self, pclds, rgbs, masks, pred_ctr_ofs, gt_ctr_ofs, labels, cnt,
                           cls_ids, RTs, pred_kp_ofs, min_cnt=20, merge_clus=False, bbox=False,
                           ds='YCB', cls_type=None, use_p2d=False, vote_type=None,
                           use_ctr_clus_flter=True, use_ctr=True, ds_type="ycb", obj_id=0):
        """Same arguments as the reference (:345-351).  The reference fans the frames of a batch
        out to a thread pool (:373-380); here the whole batch is one device-side pipeline.
        Returns the per-frame pose lists (and stores them in ``self.last_poses``)."""
        masks = masks.long()
        if ds_type == "ycb":
            res = cal_batch_poses(pclds, masks, pred_ctr_ofs, pred_kp_ofs, use_ctr, self.n_cls,
                                  use_ctr_clus_flter)
            present = res["present"].cpu().numpy()
            poses = res["poses"].cpu().numpy()
            out = []
            for f in range(poses.shape[0]):
                ids = np.nonzero(present[f])[0] + 1
                out.append((ids, [poses[f, c - 1] for c in ids]))
        else:
            res = cal_batch_poses_lm(pclds, masks, pred_ctr_ofs, pred_kp_ofs, use_ctr, self.n_cls,
                                     use_ctr_clus_flter, obj_id)
            poses = res["poses"].cpu().numpy()
            out = [[poses[f]] for f in range(poses.shape[0])]
        self.last_poses = out
        return out
