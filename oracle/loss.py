"""CPU restatement (test infrastructure) of of_l1_loss (pvn3d/lib/loss.py:45-73), forward and the
analytic gradient w.r.t. pred_ofsts.  Pinned by tests/golden/loss_ref.npz (outputs + autograd
gradients of the reference's own function, tests/golden/make_golden.py)."""
import numpy as np


def of_l1_loss(pred_ofsts, kp_targ_ofst, labels):
    bs, n_kpts, n_pts, c = pred_ofsts.shape
    w = (labels.reshape(bs, 1, n_pts, 1) > 1e-8).astype(np.float32)
    targ = kp_targ_ofst.reshape(bs, n_pts, n_kpts, 3).transpose(0, 2, 1, 3)
    in_loss = w * np.abs(pred_ofsts - targ)
    num = in_loss.reshape(bs, n_kpts, -1).astype(np.float64).sum(2)
    den = np.repeat(w, n_kpts, 1).reshape(bs, n_kpts, -1).astype(np.float64).sum(2) + 1e-3
    return (num / den).astype(np.float32)


def of_l1_loss_grad(pred_ofsts, kp_targ_ofst, labels, grad_loss):
    bs, n_kpts, n_pts, c = pred_ofsts.shape
    w = (labels.reshape(bs, 1, n_pts, 1) > 1e-8).astype(np.float32)
    targ = kp_targ_ofst.reshape(bs, n_pts, n_kpts, 3).transpose(0, 2, 1, 3)
    den = w.reshape(bs, 1, -1).sum(2) + np.float32(1e-3)                     # (bs,1)
    return (grad_loss.reshape(bs, n_kpts, 1, 1) / den.reshape(bs, 1, 1, 1) * w * np.sign(pred_ofsts - targ)).astype(np.float32)
