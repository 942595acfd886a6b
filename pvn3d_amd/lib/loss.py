"""The vote loss of the reference's pvn3d/lib/loss.py: ``of_l1_loss`` (:45-73) and ``OFLoss``
(:76-90) with the same signatures.  Forward and backward are one HIP launch each
(csrc/vote_loss.hip) instead of ~8 elementwise torch kernels over (bs, K, N, 3) temporaries;
summation order is fixed, so the loss is bit-reproducible.  ``FocalLoss`` (:13-42, the
segmentation loss, not on the hot path) is provided in plain torch so that
``from lib.loss import OFLoss, FocalLoss`` keeps working when this module is substituted."""
import torch
from torch.amp import custom_bwd, custom_fwd
from torch.nn.modules.loss import _Loss

from .._lib import lib, check, on_device


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class _OfL1Loss(torch.autograd.Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)     # fp32 loss inside a bf16 autocast region
    def forward(ctx, pred_ofsts, kp_targ_ofst, w_labels):
        bs, n_kpts, n_pts, c = pred_ofsts.shape
        pred = pred_ofsts.contiguous()
        targ = kp_targ_ofst.contiguous()
        loss = torch.empty((bs, n_kpts), dtype=torch.float32, device=pred.device)
        wsum = torch.empty((bs, n_kpts), dtype=torch.float32, device=pred.device)
        with on_device(pred.device):
            check(lib.pvn3d_of_l1_loss(bs, n_kpts, n_pts, pred.data_ptr(), targ.data_ptr(), w_labels.data_ptr(),
                                       loss.data_ptr(), wsum.data_ptr(), _stream(pred)), "of_l1_loss")
        ctx.save_for_backward(pred, targ, w_labels, wsum)
        return loss

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_loss):
        pred, targ, w_labels, wsum = ctx.saved_tensors
        bs, n_kpts, n_pts, c = pred.shape
        grad_pred = torch.empty_like(pred)
        g = grad_loss.contiguous().float()
        with on_device(pred.device):
            check(lib.pvn3d_of_l1_loss_grad(bs, n_kpts, n_pts, pred.data_ptr(), targ.data_ptr(),
                                            w_labels.data_ptr(), wsum.data_ptr(), g.data_ptr(),
                                            grad_pred.data_ptr(), _stream(pred)), "of_l1_loss_grad")
        return grad_pred, None, None


def of_l1_loss(pred_ofsts, kp_targ_ofst, labels, sigma=1.0, normalize=True, reduce=False):
    """
    :param pred_ofsts:      [bs, n_kpts, n_pts, c]
    :param kp_targ_ofst:    [bs, n_pts, n_kpts, c]
    :param labels:          [bs, n_pts, 1]
    Returns [bs, n_kpts] (normalize=True) or the weighted |diff| [bs, n_kpts, n_pts, c].
    `sigma` and `reduce` are accepted and ignored exactly like the reference (its
    ``torch.mean(in_loss)`` result is discarded, :70-71).
    """
    if not pred_ofsts.is_cuda:
        raise RuntimeError("CPU not supported")
    bs, n_kpts, n_pts, c = pred_ofsts.size()
    if pred_ofsts.dtype in (torch.bfloat16, torch.float16):
        pred_ofsts = pred_ofsts.float()        # mixed-precision heads: the loss itself is evaluated in fp32
    if c != 3 or pred_ofsts.dtype != torch.float32:
        raise RuntimeError("pred_ofsts must be a float tensor of shape (bs, n_kpts, n_pts, 3)")
    w_labels = (labels.reshape(bs, n_pts) > 1e-8).float().contiguous()
    targ = kp_targ_ofst.reshape(bs, n_pts, n_kpts, 3).float()
    if not normalize:
        w = w_labels.view(bs, 1, n_pts, 1)
        return w * torch.abs(pred_ofsts - targ.permute(0, 2, 1, 3))
    return _OfL1Loss.apply(pred_ofsts, targ, w_labels)


class FocalLoss(_Loss):
    """Focal loss -(1 - p_t)^gamma * alpha_t * log p_t over class logits (reference :13-42):
    `input` (N, C) or (N, C, ...) logits, `target` integer class ids; `alpha` a float (binary:
    [alpha, 1 - alpha]) or a per-class list; p_t is treated as a constant in the backward pass,
    as the reference does."""

    def __init__(self, gamma=0, alpha=None, size_average=True):
        super(FocalLoss, self).__init__()
        self.gamma = gamma
        if isinstance(alpha, (float, int)):
            alpha = torch.tensor([alpha, 1 - alpha], dtype=torch.float32)
        elif isinstance(alpha, list):
            alpha = torch.tensor(alpha, dtype=torch.float32)
        self.alpha = alpha
        self.size_average = size_average

    def forward(self, input, target):
        if input.dim() > 2:
            input = input.reshape(input.size(0), input.size(1), -1).transpose(1, 2)
            input = input.reshape(-1, input.size(2))
        target = target.reshape(-1, 1)
        logpt = torch.log_softmax(input, dim=1).gather(1, target).reshape(-1)
        pt = logpt.detach().exp()
        if self.alpha is not None:
            self.alpha = self.alpha.to(device=input.device, dtype=input.dtype)
            logpt = logpt * self.alpha.gather(0, target.reshape(-1))
        loss = -1 * (1 - pt) ** self.gamma * logpt
        return loss.mean() if self.size_average else loss.sum()


class OFLoss(_Loss):
    def __init__(self):
        super(OFLoss, self).__init__(True)

    def forward(self, pred_ofsts, kp_targ_ofst, labels, normalize=True, reduce=False):
        return of_l1_loss(pred_ofsts, kp_targ_ofst, labels, sigma=1.0, normalize=True, reduce=False)
