"""MeanShiftTorch with the reference's interface (pvn3d/lib/utils/meanshift_pytorch.py:18-51):

    ms = MeanShiftTorch(bandwidth=0.05, max_iter=300)
    ctr, labels = ms.fit(A)          # A (n,3) cuda tensor -> ctr (3,), labels (n,) bool

The dense O(n^2)-memory torch implementation of the reference is replaced by the fused gfx950
kernels (pvn3d_amd/csrc/meanshift.hip): same Gaussian kernel, same stop rule
(max seed shift < bandwidth*1e-3 or it > max_iter), same cluster pick (converged position of
the first seed whose ORIGINAL position has the most original neighbours within bandwidth).
``fit_batch`` runs many independent fits in one batched launch sequence; ``last_iters`` holds
the number of iterations the most recent call ran: by default a fit stops as soon as the winning seed has
landed on a bitwise fixed point (its centre is known then; csrc/meanshift.hip "winner stop"), which is at
most the reference's ``it``; with ``full_iterations = True`` the remaining iterations of the reference's
stop rule are run as well (same centre and labels, bit for bit) and ``last_iters`` equals that ``it``.  The reference's ``fit_batch_npts`` is broken
upstream (undefined name, never called) and is not reproduced.
"""
import torch

from . import _vote_engine as _eng


class MeanShiftTorch(object):
    def __init__(self, bandwidth=0.05, max_iter=300):
        self.bandwidth = bandwidth
        self.stop_thresh = bandwidth * 1e-3
        self.max_iter = max_iter
        self.last_iters = None
        self.full_iterations = False

    @staticmethod
    def _pack(A):
        if not A.is_cuda:
            raise RuntimeError("CPU not supported")  # same contract as the native ops
        n = A.size(0)
        pts4 = torch.zeros((max((n + 31) // 32 * 32, 32), 4), dtype=torch.float32, device=A.device)
        if n:
            pts4[:n, :3] = A
        return pts4

    def fit(self, A):
        """A (N,3) -> (centre (3,), labels (N,) bool)."""
        N, c = A.size()
        assert c == 3
        pts4 = self._pack(A.detach().float())
        seg_off = torch.zeros(1, dtype=torch.int32, device=A.device)
        seg_cnt = torch.full((1,), N, dtype=torch.int32, device=A.device)
        ctr, labels, iters = _eng.meanshift_fit_batch(pts4, seg_off, seg_cnt, max(N, 1),
                                                      self.bandwidth, self.max_iter,
                                                      aligned32=True,
                                                      kernel="nowin" if self.full_iterations else None)
        self.last_iters = iters
        return ctr[0].to(A.dtype), labels[:N].bool()

    def fit_batch(self, A_list):
        """List of (n_i,3) tensors -> (centres (S,3), list of bool label tensors)."""
        dev = A_list[0].device
        cnts = [int(a.size(0)) for a in A_list]
        offs = [0]                      # every segment starts on a multiple of 32 rows
        for n in cnts[:-1]:
            offs.append(offs[-1] + (n + 31) // 32 * 32)
        total = max(offs[-1] + (cnts[-1] + 31) // 32 * 32, 32)
        pts4 = torch.zeros((total, 4), dtype=torch.float32, device=dev)
        for o, a in zip(offs, A_list):
            if a.size(0):
                pts4[o:o + a.size(0), :3] = a.detach().float()
        seg_off = torch.tensor(offs, dtype=torch.int32, device=dev)
        seg_cnt = torch.tensor(cnts, dtype=torch.int32, device=dev)
        ctr, labels, iters = _eng.meanshift_fit_batch(pts4, seg_off, seg_cnt, max(max(cnts), 1),
                                                      self.bandwidth, self.max_iter,
                                                      aligned32=True,
                                                      kernel="nowin" if self.full_iterations else None)
        self.last_iters = iters
        return ctr, [labels[o:o + n].bool() for o, n in zip(offs, cnts)]
