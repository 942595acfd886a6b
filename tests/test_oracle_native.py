"""The C oracle for the pointnet2 ops, cross-checked against INDEPENDENT brute-force numpy
formulations (the reference ships no golden vectors for these CUDA-only ops), plus the
oracle-generated regression vectors in tests/golden/native_oracle.npz."""
import numpy as np
import pytest

from pvn3d_amd import synth


def d2_f32(a, b):
    """((dx*dx + dy*dy) + dz*dz) in fp32 with one rounding per op; a (m,3) vs b (n,3) -> (m,n)."""
    d = a[:, None, :].astype(np.float32) - b[None, :, :].astype(np.float32)
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def cloud(seed, n, wrap=0.0):
    return synth.synth_cloud(np.random.default_rng(seed), n, wrap_pad=wrap)[0]


def brute_ball_query(new_xyz, xyz, radius, nsample):
    r2 = np.float32(radius) * np.float32(radius)
    d2 = d2_f32(new_xyz, xyz)
    out = np.zeros((len(new_xyz), nsample), np.int32)
    for j in range(len(new_xyz)):
        hits = np.nonzero(d2[j] < r2)[0]
        if len(hits):
            out[j, :] = hits[0]
            k = min(nsample, len(hits))
            out[j, :k] = hits[:k]
    return out


def brute_fps(xyz, m, bs):
    """FPS with the reference's documented tie order: max d2, then smallest bit-reversed
    (k mod bs), then smallest k."""
    n = len(xyz)
    L = int(np.log2(bs))
    k = np.arange(n)
    rev = np.array([int(format(v, "0%db" % L)[::-1], 2) if L else 0 for v in (k % bs)])
    prio = rev * ((n + bs - 1) // bs) + k // bs
    x = xyz.astype(np.float32)
    mag = (x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]) + x[:, 2] * x[:, 2]
    valid = ~(mag.astype(np.float64) <= 1e-3)
    temp = np.full(n, 1e10, np.float32)
    out = [0]
    old = 0
    for _ in range(1, m):
        d = d2_f32(x[old:old + 1], x)[0]
        temp = np.where(valid, np.minimum(d, temp), temp)
        cand = np.where(valid, temp, -np.inf)
        if not valid.any():
            old = 0
        else:
            best = cand.max()
            ties = np.nonzero(cand == best)[0]
            old = int(ties[np.argmin(prio[ties])])
        out.append(old)
    return np.array(out, np.int32)


@pytest.mark.parametrize("n,m,wrap", [(512, 64, 0.0), (1000, 100, 0.2), (2048, 256, 0.1), (96, 96, 0.5)])
def test_fps_matches_tie_order_model(orc, n, m, wrap):
    xyz = cloud(n + m, n, wrap)
    got = orc.furthest_point_sampling(xyz[None], m)[0]
    want = brute_fps(xyz, m, orc.opt_n_threads(n))
    assert np.array_equal(got, want)


def test_fps_skip_rule_and_all_skipped(orc):
    xyz = cloud(5, 300)
    xyz[10:40] = 0.0                      # |p|^2 <= 1e-3: never selected, never updated
    xyz[0] = [0.01, 0.0, 0.0]             # the seed index 0 itself is skipped too
    got = orc.furthest_point_sampling(xyz[None], 50)[0]
    assert np.array_equal(got, brute_fps(xyz, 50, orc.opt_n_threads(300)))
    assert not set(got[1:].tolist()) & set(range(10, 40))
    z = np.zeros((1, 128, 3), np.float32)
    assert np.array_equal(orc.furthest_point_sampling(z, 8), np.zeros((1, 8), np.int32))


@pytest.mark.parametrize("n,m,r,ns", [(1024, 128, 0.03, 16), (2048, 64, 0.1, 32), (300, 300, 0.02, 8)])
def test_ball_query_matches_bruteforce(orc, n, m, r, ns):
    xyz = cloud(n, n, 0.1)
    new_xyz = xyz[np.random.default_rng(1).permutation(n)[:m]]
    assert np.array_equal(orc.ball_query(new_xyz[None], xyz[None], r, ns)[0],
                          brute_ball_query(new_xyz, xyz, r, ns))


def test_ball_query_no_hit_rows_are_zero(orc):
    xyz = cloud(3, 256)
    far = xyz[:4] + 100.0
    assert np.array_equal(orc.ball_query(far[None], xyz[None], 0.05, 8), np.zeros((1, 4, 8), np.int32))


@pytest.mark.parametrize("n,m", [(700, 64), (2048, 512), (33, 3)])
def test_three_nn_matches_stable_sort(orc, n, m):
    unk = cloud(n, n, 0.1)
    kn = unk[np.random.default_rng(2).permutation(n)[:m]]
    d2, idx = orc.three_nn(unk[None], kn[None])
    full = d2_f32(unk, kn)
    order = np.argsort(full, axis=1, kind="stable")[:, :3]     # ties -> earlier k, like strict '<'
    assert np.array_equal(idx[0], order.astype(np.int32))
    assert np.array_equal(d2[0], np.take_along_axis(full, order, 1))


def test_three_nn_fewer_than_three_known(orc):
    unk = cloud(9, 16)
    d2, idx = orc.three_nn(unk[None], unk[None, :2])
    assert np.all(np.isinf(d2[0, :, 2])) and np.all(idx[0, :, 2] == 0)


def test_group_gather_interpolate_match_numpy(orc):
    g = np.random.default_rng(4)
    B, C, n, m, ns = 2, 5, 200, 40, 6
    pts = g.normal(size=(B, C, n)).astype(np.float32)
    idx = g.integers(0, n, size=(B, m, ns)).astype(np.int32)
    out = orc.group_points(pts, idx)
    for b in range(B):
        assert np.array_equal(out[b], pts[b][:, idx[b]])
    i1 = g.integers(0, n, size=(B, m)).astype(np.int32)
    ga = orc.gather_points(pts, i1)
    for b in range(B):
        assert np.array_equal(ga[b], pts[b][:, i1[b]])
    i3 = g.integers(0, n, size=(B, m, 3)).astype(np.int32)
    w = g.random((B, m, 3)).astype(np.float32)
    it = orc.three_interpolate(pts, i3, w)
    for b in range(B):
        p = pts[b][:, i3[b]]                                   # (C,m,3)
        want = (p[..., 0] * w[b, :, 0] + p[..., 1] * w[b, :, 1]) + p[..., 2] * w[b, :, 2]
        assert np.array_equal(it[b], want.astype(np.float32))
    # gradients are the adjoints of the forward gathers
    go = g.normal(size=out.shape).astype(np.float32)
    gg = orc.group_points_grad(go, idx, n)
    assert np.allclose((gg * pts).sum(), (go * out).sum(), rtol=1e-4)
    gi = g.normal(size=it.shape).astype(np.float32)
    gt = orc.three_interpolate_grad(gi, i3, w, n)
    assert np.allclose((gt * pts).sum(), (gi * it).sum(), rtol=1e-4)
    g1 = g.normal(size=ga.shape).astype(np.float32)
    assert np.allclose((orc.gather_points_grad(g1, i1, n) * pts).sum(), (g1 * ga).sum(), rtol=1e-4)


def test_three_interpolate_grad_refbug_is_forward_with_swapped_sizes(orc):
    """interpolate.cpp:89-93: the reference calls the forward wrapper with (m:=n, n:=m)."""
    g = np.random.default_rng(6)
    B, C, n, m = 1, 3, 50, 20
    go = g.normal(size=(B, C, n)).astype(np.float32)
    i3 = g.integers(0, m, size=(B, n, 3)).astype(np.int32)    # valid indices into m known points
    w = g.random((B, n, 3)).astype(np.float32)
    bug = orc.three_interpolate_grad(go, i3, w, m, refbug=True)
    want = orc.three_interpolate(go, i3[:, :m], w[:, :m])      # B == 1: strides coincide
    assert np.array_equal(bug, want)


def test_native_regression_vectors(orc, golden):
    z = golden("native_oracle.npz")
    xyz = z["xyz"]
    fps = orc.furthest_point_sampling(xyz, 512)
    assert np.array_equal(fps, z["fps"])
    new_xyz = np.take_along_axis(xyz, fps[..., None].astype(np.int64).repeat(3, -1), 1)
    assert np.array_equal(orc.ball_query(new_xyz, xyz, 0.025, 16), z["bq_r0025_16"])
    assert np.array_equal(orc.ball_query(new_xyz, xyz, 0.05, 32), z["bq_r005_32"])
    d2, idx = orc.three_nn(xyz, new_xyz)
    assert np.array_equal(idx, z["nn_idx"]) and np.array_equal(d2, z["nn_d2"])
