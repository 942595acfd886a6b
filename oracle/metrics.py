"""CPU restatement (test infrastructure) of the reference's pose metrics:
  cal_add / cal_adds   Basic_Utils.cal_add_cuda / cal_adds_cuda  (pvn3d/lib/utils/basic_utils.py:617-635)
  voc_ap / cal_auc     VOCap (:32-44), Basic_Utils.cal_auc (:597-605)
Pinned by tests/golden/metrics_ref.npz = outputs of the reference's own functions
(tests/golden/make_golden.py)."""
import numpy as np


def _xform(RT, pts):
    RT = np.asarray(RT, dtype=np.float32)
    return (pts.astype(np.float32) @ RT[:, :3].T + RT[:, 3]).astype(np.float32)


def cal_add(pred_RT, gt_RT, p3ds):
    pd, gt = _xform(pred_RT, p3ds), _xform(gt_RT, p3ds)
    return float(np.mean(np.linalg.norm(pd - gt, axis=1).astype(np.float32)))


def cal_adds(pred_RT, gt_RT, p3ds, chunk=512):
    pd, gt = _xform(pred_RT, p3ds), _xform(gt_RT, p3ds)
    mins = np.empty(len(gt), dtype=np.float32)
    for i0 in range(0, len(gt), chunk):          # dis[i][j] = |pd_j - gt_i|, min over j
        d = np.linalg.norm(pd[None, :, :] - gt[i0:i0 + chunk, None, :], axis=2)
        mins[i0:i0 + chunk] = d.min(axis=1)
    return float(np.mean(mins))


def voc_ap(rec, prec):
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
    return np.sum((mrec[i] - mrec[i - 1]) * mpre[i]) * 10


def cal_auc(add_dis, max_dis=0.1):
    D = np.array(add_dis)
    D[np.where(D > max_dis)] = np.inf
    D = np.sort(D)
    n = len(add_dis)
    acc = np.cumsum(np.ones((1, n)), dtype=np.float32) / n
    return voc_ap(D, acc) * 100
