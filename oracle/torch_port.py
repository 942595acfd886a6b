"""Dense torch-CPU restatement of MeanShiftTorch.fit (TEST INFRASTRUCTURE / CPU BASELINE ONLY).

Same tensor algebra as pvn3d/lib/utils/meanshift_pytorch.py:13-51 -- every iteration
materialises the (n,n,3) difference tensor, the (n,n) Gaussian weights and the weighted sums --
so its wall-clock on the host cores is what "the reference's CPU MeanShift path" costs
(BASELINE.md section 3).  It exists because the reference module itself cannot travel to the
GPU box (/root/reference is absent there); tests/test_oracle_golden.py checks it against
outputs of the real reference recorded in tests/golden/.
"""
import numpy as np
import torch


def gaussian_kernel(distance, bandwidth):
    return (1 / (bandwidth * torch.sqrt(2 * torch.tensor(np.pi)))) * torch.exp(-0.5 * ((distance / bandwidth)) ** 2)


def meanshift_fit_dense(A, bandwidth, max_iter=300):
    """A (n,3) CPU float tensor -> (ctr (3,), labels (n,) bool, iters)."""
    stop_thresh = bandwidth * 1e-3
    n, c = A.shape
    it = 0
    C = A.clone()
    pts = A.view(1, n, c)
    while True:
        it += 1
        dis = torch.norm(C.view(n, 1, c) - pts, dim=2)            # (n,n)
        w = gaussian_kernel(dis, bandwidth).view(n, n, 1)
        new_C = torch.sum(w * pts, dim=1) / torch.sum(w, dim=1)
        adis = torch.norm(new_C - C, dim=1)
        C = new_C
        if torch.max(adis) < stop_thresh or it > max_iter:
            break
    dis0 = torch.norm(pts - A.view(n, 1, c), dim=2)
    num_in = torch.sum(dis0 < bandwidth, dim=1)
    _, max_idx = torch.max(num_in, 0)
    labels = dis0[max_idx] < bandwidth
    return C[max_idx, :], labels, it


def best_fit_transform_np(A, B):
    """numpy restatement of pvn3d/lib/utils/basic_utils.py:47-80 (same LAPACK call)."""
    assert A.shape == B.shape
    m = A.shape[1]
    ca, cb = np.mean(A, axis=0), np.mean(B, axis=0)
    H = np.dot((A - ca).T, B - cb)
    U, S, Vt = np.linalg.svd(H)
    R = np.dot(Vt.T, U.T)
    if np.linalg.det(R) < 0:
        Vt[m - 1, :] *= -1
        R = np.dot(Vt.T, U.T)
    t = cb.T - np.dot(R, ca.T)
    T = np.zeros((3, 4))
    T[:, :3] = R
    T[:, 3] = t
    return T
