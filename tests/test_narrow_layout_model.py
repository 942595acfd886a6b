"""CPU: lane-level model of the narrow-chain kernels' register data flow (csrc/sa_mlp_split.hip: sa_chain_narrow_kernel).

The kernel keeps activations in registers from layer to layer: the D layout of v_mfma_f32_32x32x16_f16 (lane = column,
registers = rows 8 j + 4 half + i) is used as the B fragment of the next layer, which requires the next layer's K index
to be permuted inside every 16-k slab -- applied to the host-packed weights (PackedMLP.split2: _pack_weight_split2)
while they are copied into LDS -- and the last layer is evaluated transposed so that the max over nsample happens inside
a lane.  This test restates those index maps in numpy (the same formulas as the kernel: staging permutation, fragment
order s * T + t, D-register -> next-fragment position, transposed pool) on top of the REAL host packing, and compares
the result with a direct evaluation of the chain.  It pins the contract between the packer and the kernel without a GPU
(operand layouts of the MFMA as documented in include/pvn3d_hip.h and used by every kernel of the file)."""
import numpy as np
import pytest
import torch

from pvn3d_amd.lib.pointnet2_utils import _fused_mlp as fm


def _stage(packed, S, T, permute):
    """packed int16 [S][T][2 pieces][64 lanes][8] -> the kernel's LDS image of the layer (bytes)"""
    src = packed.contiguous().view(torch.uint8).numpy().reshape(-1)
    dst = np.zeros(S * T * 2048, np.uint8)
    if not permute:
        dst[:] = src[:S * T * 2048]
        return dst
    for u in range(S * T * 256):                                  # 8-byte units, as the staging loop of the kernel
        f, pc, ln, q = u >> 8, (u >> 7) & 1, (u >> 1) & 63, u & 1
        m, h = ln & 31, ln >> 5
        so = f * 2048 + pc * 1024 + (m + 32 * q) * 16 + 8 * h
        do = f * 2048 + pc * 1024 + ln * 16 + 8 * q
        dst[do:do + 8] = src[so:so + 8]
    return dst


def _frag(lds, i):
    """fragment i (hi + lo pieces) -> (64 lanes, 8 positions) float64"""
    out = np.zeros((64, 8))
    for piece in range(2):
        b = lds[i * 2048 + piece * 1024: i * 2048 + piece * 1024 + 1024]
        out += np.frombuffer(b.tobytes(), np.float16).reshape(64, 8).astype(np.float64)
    return out


def _mfma(A, B, acc):
    """v_mfma_f32_32x32x16: A lane (row m, half h) and B lane (column c, half h) hold 8 k positions each; the result
    D[m][c] lands in lane (c, h'), register 4 j + i = row 8 j + 4 h' + i"""
    D = A[:32] @ B[:32].T + A[32:] @ B[32:].T
    out = acc.copy()
    for lane in range(64):
        c, hh = lane & 31, lane >> 5
        for j in range(4):
            out[lane, 4 * j:4 * j + 4] += D[8 * j + 4 * hh: 8 * j + 4 * hh + 4, c]
    return out


def _bias_acc(bias_padded, t):
    a = np.zeros((64, 16))
    for lane in range(64):
        hh = lane >> 5
        for g in range(4):
            a[lane, 4 * g:4 * g + 4] = bias_padded[32 * t + 8 * g + 4 * hh: 32 * t + 8 * g + 4 * hh + 4]
    return a


@pytest.mark.parametrize("dims,ns", [([9, 16, 16, 32], 16), ([9, 32, 32, 64], 32), ([99, 64, 64, 128], 16),
                                     ([99, 64, 96, 128], 32)])
def test_register_chaining_and_transposed_pool_reproduce_the_chain(dims, ns):
    rs = np.random.RandomState(len(dims) + ns)
    K0, M0, M1, M2 = dims
    Ws = [rs.randn(M0, K0) * 0.3, rs.randn(M1, M0) * 0.3, rs.randn(M2, M1) * 0.3]
    bs = [rs.randn(M0) * 0.1, rs.randn(M1) * 0.1, rs.randn(M2) * 0.1]
    S0, S1, S2 = (K0 + 15) // 16, (M0 + 15) // 16, (M1 + 15) // 16
    T0, T1, T2 = (S1 + 1) // 2, (S2 + 1) // 2, (M2 + 31) // 32
    packed = [fm._pack_weight_split2(torch.tensor(W, dtype=torch.float32)) for W in Ws]
    lds = [_stage(packed[0], S0, T0, False), _stage(packed[1], S1, T1, True), _stage(packed[2], S2, T2, True)]
    X = rs.randn(K0, 32)                                           # one tile: 32 columns, k = features then xyz

    def pad(b, T):
        o = np.zeros(T * 32)
        o[:len(b)] = b
        return o
    bp = [pad(bs[0], T0), pad(bs[1], T1), pad(bs[2], T2)]
    # layer 0: standard K order, fragment of lane (c, h) = channels 16 s + 8 h + 0..7 of column c
    acc0 = [_bias_acc(bp[0], t) for t in range(T0)]
    for s in range(S0):
        B = np.zeros((64, 8))
        for lane in range(64):
            c, hh = lane & 31, lane >> 5
            for p in range(8):
                k = 16 * s + 8 * hh + p
                B[lane, p] = X[k, c] if k < K0 else 0.0
        for t in range(T0):
            acc0[t] = _mfma(_frag(lds[0], s * T0 + t), B, acc0[t])

    def next_frags(acc):                                           # registers 0..7 -> slab 2 t, 8..15 -> slab 2 t + 1
        x = np.maximum(acc, 0)
        return [x[:, 0:8].copy(), x[:, 8:16].copy()]
    b1 = sum((next_frags(a) for a in acc0), [])
    acc1 = [_bias_acc(bp[1], t) for t in range(T1)]
    for s in range(S1):
        for t in range(T1):
            acc1[t] = _mfma(_frag(lds[1], s * T1 + t), b1[s], acc1[t])
    b2 = sum((next_frags(a) for a in acc1), [])
    out = np.zeros((M2, 32 // ns))
    for t in range(T2):
        a = np.zeros((64, 16))
        for s in range(S2):
            a = _mfma(b2[s], _frag(lds[2], s * T2 + t), a)         # transposed: activations as A, weights as B
        a = a + bp[2][32 * t + (np.arange(64) & 31)][:, None]      # bias after the product, one value per lane
        lo, hi = np.maximum(a[:, 0:8].max(1), 0), np.maximum(a[:, 8:16].max(1), 0)      # columns 0..15 / 16..31
        for lane in range(64):
            col, hh = lane & 31, lane >> 5
            row = 32 * t + col
            if row >= M2:
                continue
            if ns == 32:
                if hh == 0:
                    out[row, 0] = max(lo[lane], hi[lane], lo[lane ^ 32], hi[lane ^ 32])
            else:
                out[row, hh] = max(hi[lane], hi[lane ^ 32]) if hh else max(lo[lane], lo[lane ^ 32])

    def wq(W):                                                     # what the two fp16 pieces of the packer represent
        Wt = torch.tensor(W, dtype=torch.float32)
        h = Wt.half()
        return (h.double() + (Wt - h.float()).half().double()).numpy()
    y = np.maximum(wq(Ws[0]) @ X + bs[0][:, None], 0)
    y = np.maximum(wq(Ws[1]) @ y + bs[1][:, None], 0)
    y = np.maximum(wq(Ws[2]) @ y + bs[2][:, None], 0)
    ref = y.reshape(M2, 32 // ns, ns).max(2)
    assert np.abs(out - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
