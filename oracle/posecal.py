"""CPU restatement of the vote -> cluster -> pose drivers (TEST INFRASTRUCTURE ONLY).

  cal_frame_poses_lm : pvn3d/lib/utils/pvn3d_eval_utils.py:156-201
  cal_frame_poses    : pvn3d/lib/utils/pvn3d_eval_utils.py:37-110
restated on numpy arrays with the `.cuda()` calls dropped (the original cannot be imported in
this container: common.Config -> yaml.load TypeError, lib/__init__ -> torch._six).  ``fit`` and
``bft`` default to the C oracle (oracle/native.py); tests/golden/make_golden.py passes the
reference's own MeanShiftTorch.fit / best_fit_transform instead to produce pinned outputs.
"""
import numpy as np

from . import native


def _fit_native(A, bw):
    ctr, labels, iters = native.meanshift_fit(A, bw)
    return ctr, labels, iters


def cal_frame_poses_lm(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter,
                       mesh_kps_ctr, fit=_fit_native, bft=native.best_fit_transform,
                       return_debug=False):
    """mesh_kps_ctr (K+1,3): object keypoints with the centre appended (get_kps + get_ctr)."""
    n_kps, n_pts, _ = pred_kp_of.shape
    pred_ctr = pcld - ctr_of[0]                                  # :160
    pred_kp = pcld.reshape(1, n_pts, 3).repeat(n_kps, 0) - pred_kp_of   # :161
    radius = 0.08                                                # :163
    cls_kps = np.zeros((n_cls, n_kps + 1, 3), np.float32)
    iters = np.zeros(n_kps + 1, np.int64)
    pred_pose_lst = []
    cls_id = 1
    cls_msk = mask == cls_id
    if cls_msk.sum() < 1:
        pred_pose_lst.append(np.identity(4)[:3, :])              # :172-173
    else:
        cls_voted_kps = pred_kp[:, cls_msk, :]
        ctr, ctr_labels, iters[n_kps] = fit(pred_ctr[cls_msk, :], radius)
        ctr_labels = np.array(ctr_labels, dtype=bool)
        if ctr_labels.sum() < 1:
            ctr_labels[0] = True
        cls_kps[cls_id, n_kps, :] = ctr
        in_pred_kp = cls_voted_kps[:, ctr_labels, :] if use_ctr_clus_flter else cls_voted_kps
        for ikp in range(n_kps):
            cls_kps[cls_id, ikp, :], _, iters[ikp] = fit(in_pred_kp[ikp], radius)
        npts = n_kps + 1 if use_ctr else n_kps
        pred_RT = bft(np.ascontiguousarray(mesh_kps_ctr[:npts], np.float32),
                      np.ascontiguousarray(cls_kps[cls_id, :npts], np.float32))
        pred_pose_lst.append(pred_RT)
    if return_debug:
        return pred_pose_lst, cls_kps[cls_id], iters
    return pred_pose_lst


def cal_frame_poses(pcld, mask, ctr_of, pred_kp_of, use_ctr, n_cls, use_ctr_clus_flter,
                    mesh_kps_ctr_of_cls, ycb_r_lst, fit=_fit_native,
                    bft=native.best_fit_transform, return_debug=False):
    """mesh_kps_ctr_of_cls: callable cls_id -> (K+1,3) keypoints+centre."""
    n_kps, n_pts, _ = pred_kp_of.shape
    pred_ctr = pcld - ctr_of[0]
    pred_kp = pcld.reshape(1, n_pts, 3).repeat(n_kps, 0) - pred_kp_of
    radius = 0.08
    cls_kps = np.zeros((n_cls, n_kps + 1, 3), np.float32)
    pred_cls_ids = np.unique(mask[mask > 0])                     # :49
    mask = mask.copy()
    if use_ctr_clus_flter:                                       # :50-72
        ctrs = []
        for cls_id in pred_cls_ids:
            ctr, _, _ = fit(pred_ctr[mask == cls_id, :], radius)
            ctrs.append(ctr)
        ctrs = np.array(ctrs).astype(np.float32)
        ctr_dis = np.linalg.norm(pred_ctr[:, None, :] - ctrs[None, :, :], axis=2).astype(np.float32)
        min_idx = np.argmin(ctr_dis, axis=1)                     # first min, like torch.min
        min_dis = ctr_dis[np.arange(n_pts), min_idx]
        msk_closest_ctr = pred_cls_ids[min_idx]
        new_msk = mask.copy()
        for cls_id in pred_cls_ids:
            if cls_id == 0:
                break
            min_msk = min_dis < np.float32(ycb_r_lst[cls_id - 1] * 0.8)
            update_msk = (mask > 0) & (msk_closest_ctr == cls_id) & min_msk
            new_msk[update_msk] = msk_closest_ctr[update_msk]
        mask = new_msk
    pred_pose_lst = []
    for cls_id in pred_cls_ids:
        if cls_id == 0:
            break
        cls_msk = mask == cls_id
        if cls_msk.sum() < 1:
            pred_pose_lst.append(np.identity(4)[:3, :])
            continue
        cls_voted_kps = pred_kp[:, cls_msk, :]
        ctr, ctr_labels, _ = fit(pred_ctr[cls_msk, :], radius)
        ctr_labels = np.array(ctr_labels, dtype=bool)
        if ctr_labels.sum() < 1:
            ctr_labels[0] = True
        cls_kps[cls_id, n_kps, :] = ctr
        in_pred_kp = cls_voted_kps[:, ctr_labels, :] if use_ctr_clus_flter else cls_voted_kps
        for ikp in range(n_kps):
            cls_kps[cls_id, ikp, :], _, _ = fit(in_pred_kp[ikp], radius)
        npts = n_kps + 1 if use_ctr else n_kps
        mesh = mesh_kps_ctr_of_cls(int(cls_id))
        pred_RT = bft(np.ascontiguousarray(mesh[:npts], np.float32),
                      np.ascontiguousarray(cls_kps[cls_id, :npts], np.float32))
        pred_pose_lst.append(pred_RT)
    if return_debug:
        return pred_cls_ids, pred_pose_lst, cls_kps, mask
    return pred_cls_ids, pred_pose_lst
