"""Row (c) pin for the nine native ops: the C oracle (oracle/pvn3d_oracle.c) against the REFERENCE'S OWN
kernels compiled for the CPU (oracle/_ref, recipe oracle/ref_shim/build_ref.py) and against the fixture
those kernels produced at the BASELINE shapes (tests/golden/native_ref.npz, make_golden_native.py).

The fixture tests run anywhere; the live `_ref` tests need the prebuilt oracle/_ref/*.so (built in the
development container from /root/reference, shipped as files to the GPU box) and skip otherwise.
"""
import json

import numpy as np
import pytest

from oracle import ref

NPOINT = [2048, 1024, 512, 128]
RADII = [(0.0175, 0.025), (0.025, 0.05), (0.05, 0.1), (0.1, 0.2)]
NSAMPLE = (16, 32)

needs_ref = pytest.mark.skipif(not ref.available("nofma"), reason="oracle/_ref not built (needs /root/reference)")


def _clouds(rng, b, n, kind):
    if kind == "normal":
        return rng.normal(size=(b, n, 3)).astype(np.float32)
    if kind == "lattice":  # exact ties everywhere
        g = rng.integers(0, 6, size=(b, n, 3)).astype(np.float32) * 0.25 + 0.5
        return g
    if kind == "dups":  # 'wrap'-padded duplicates
        x = rng.normal(size=(b, n, 3)).astype(np.float32) + 2.0
        x[:, n - n // 8:] = x[:, :n // 8]
        return x
    if kind == "origin":  # points the FPS skip rule drops (|p|^2 <= 1e-3)
        x = rng.normal(size=(b, n, 3)).astype(np.float32)
        x[:, ::7] *= 0.01
        return x
    raise ValueError(kind)


# ---------------------------------------------------------------- fixture (reference-generated) vs oracle
@pytest.mark.parametrize("c", [0, 1])
def test_oracle_reproduces_reference_fixture(orc, golden, c):
    """Every FPS / ball_query / three_nn output the reference's kernels gave on the 12 288-point clouds."""
    z = golden("native_ref.npz")
    levels = [z["c%d_xyz" % c][None]]
    for l in range(4):
        cur = levels[-1]
        fps = orc.furthest_point_sampling(cur, NPOINT[l])
        assert np.array_equal(fps[0], z["c%d_fps%d" % (c, l)].astype(np.int32)), "fps level %d" % l
        new = np.ascontiguousarray(cur[:, fps[0]])
        for s in range(2):
            bq = orc.ball_query(new, cur, RADII[l][s], NSAMPLE[s])
            assert np.array_equal(bq[0], z["c%d_bq%d_%d" % (c, l, s)].astype(np.int32)), "ball_query %d/%d" % (l, s)
        levels.append(new)
    for l in range(4):
        d2, idx = orc.three_nn(levels[l], levels[l + 1])
        assert np.array_equal(idx[0], z["c%d_nn%d_idx" % (c, l)].astype(np.int32)), "three_nn idx %d" % l
        assert np.array_equal(d2[0], z["c%d_nn%d_d2" % (c, l)]), "three_nn dist2 %d" % l


def test_fma_sensitivity_recorded(golden):
    """The reference's kernels built with and without floating-point contraction pick the same indices
    on the BASELINE clouds; only three_nn's returned distances move (last-ulp)."""
    flips = json.loads(bytes(golden("native_ref.npz")["fma_flips"]).decode())
    assert flips["fps"][0] == 0 and flips["ball_query"][0] == 0 and flips["three_nn_idx"][0] == 0
    assert flips["fps"][1] == 2 * sum(NPOINT) and flips["three_nn_d2"][0] > 0


# ---------------------------------------------------------------- live: oracle == reference kernels
@needs_ref
@pytest.mark.parametrize("kind", ["normal", "lattice", "dups", "origin"])
@pytest.mark.parametrize("n,m", [(1, 1), (7, 3), (64, 64), (100, 37), (513, 128), (1024, 200), (2048, 512)])
def test_fps_equals_reference(orc, kind, n, m):
    rng = np.random.default_rng(n * 131 + m)
    xyz = _clouds(rng, 2, n, kind)
    m = min(m, n)
    assert np.array_equal(orc.furthest_point_sampling(xyz, m), ref.furthest_point_sampling(xyz, m))


@needs_ref
def test_opt_n_threads_equals_reference(orc):
    for w in list(range(1, 1100)) + [2047, 2048, 4095, 4096, 12288, 1 << 20]:
        assert orc.opt_n_threads(w) == ref.opt_n_threads(w), w


@needs_ref
@pytest.mark.parametrize("kind", ["normal", "lattice", "dups"])
@pytest.mark.parametrize("n,m,r,ns", [(50, 9, 0.4, 4), (300, 64, 0.3, 16), (1000, 256, 0.5, 32), (2048, 700, 0.2, 64),
                                      (33, 33, 10.0, 64), (128, 16, 1e-4, 8)])
def test_ball_query_equals_reference(orc, kind, n, m, r, ns):
    rng = np.random.default_rng(n + m + ns)
    xyz = _clouds(rng, 2, n, kind)
    new = np.ascontiguousarray(xyz[:, rng.permutation(n)[:m]])
    if kind == "normal":
        new[:, ::5] += 100.0  # empty balls: rows stay zero
    assert np.array_equal(orc.ball_query(new, xyz, r, ns), ref.ball_query(new, xyz, r, ns))


@needs_ref
@pytest.mark.parametrize("kind", ["normal", "lattice", "dups"])
@pytest.mark.parametrize("n,m", [(5, 1), (9, 2), (64, 3), (700, 100), (2048, 512)])
def test_three_nn_equals_reference(orc, kind, n, m):
    rng = np.random.default_rng(n * 7 + m)
    unk = _clouds(rng, 2, n, kind)
    kn = _clouds(rng, 2, m, kind)
    d2a, ia = orc.three_nn(unk, kn)
    d2b, ib = ref.three_nn(unk, kn)
    assert np.array_equal(ia, ib)
    assert np.array_equal(d2a, d2b)


@needs_ref
@pytest.mark.parametrize("b,c,n,npoint,ns", [(1, 1, 5, 3, 2), (2, 9, 100, 40, 16), (2, 131, 300, 64, 32), (1, 515, 128, 16, 8)])
def test_group_gather_interpolate_equal_reference(orc, b, c, n, npoint, ns):
    rng = np.random.default_rng(c + n)
    pts = rng.normal(size=(b, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, npoint, ns)).astype(np.int32)
    assert np.array_equal(orc.group_points(pts, idx), ref.group_points(pts, idx))
    g = rng.normal(size=(b, c, npoint, ns)).astype(np.float32)
    np.testing.assert_allclose(orc.group_points_grad(g, idx, n), ref.group_points_grad(g, idx, n), rtol=1e-5, atol=1e-6)
    i1 = idx[:, :, 0].copy()
    assert np.array_equal(orc.gather_points(pts, i1), ref.gather_points(pts, i1))
    g1 = rng.normal(size=(b, c, npoint)).astype(np.float32)
    np.testing.assert_allclose(orc.gather_points_grad(g1, i1, n), ref.gather_points_grad(g1, i1, n), rtol=1e-5, atol=1e-6)
    # three_interpolate: known = the n points, unknown = q queries
    q = npoint * 3
    i3 = rng.integers(0, n, size=(b, q, 3)).astype(np.int32)
    w = rng.random(size=(b, q, 3)).astype(np.float32)
    w /= w.sum(-1, keepdims=True)
    assert np.array_equal(orc.three_interpolate(pts, i3, w), ref.three_interpolate(pts, i3, w))
    gq = rng.normal(size=(b, c, q)).astype(np.float32)
    np.testing.assert_allclose(orc.three_interpolate_grad(gq, i3, w, n), ref.three_interpolate_grad(gq, i3, w, n),
                               rtol=1e-5, atol=1e-6)
    if q >= n:  # the binding's forward-with-swapped-sizes call (interpolate.cpp:89-93) reads rows j < m only
        assert np.array_equal(orc.three_interpolate_grad(gq, i3 % q, w, n, refbug=True),
                              ref.three_interpolate_grad(gq, i3 % q, w, n, refbug=True))


@needs_ref
def test_fma_build_differs_only_in_distances():
    """Sensitivity to contraction on adversarial inputs (lattice ties): index outputs of the two builds."""
    rng = np.random.default_rng(5)
    tot = dict(fps=0, bq=0, nn=0)
    for kind in ("normal", "dups", "lattice"):
        xyz = _clouds(rng, 1, 1024, kind)
        a, b = (ref.furthest_point_sampling(xyz, 256, variant=v) for v in ("nofma", "fma"))
        tot["fps"] += int((a != b).sum())
        new = np.ascontiguousarray(xyz[:, a[0]])
        a, b = (ref.ball_query(new, xyz, 0.3, 32, variant=v) for v in ("nofma", "fma"))
        tot["bq"] += int((a != b).sum())
        a, b = (ref.three_nn(xyz, new, variant=v)[1] for v in ("nofma", "fma"))
        tot["nn"] += int((a != b).sum())
    print("index flips nofma vs fma:", tot)
    # lattice coordinates are exact in fp32 (multiples of 0.25), so contraction cannot change anything there;
    # on continuous clouds a flip needs two candidates within one ulp, i.e. it is rare but legal -- bounded, not zero
    assert tot["fps"] <= 4 and tot["bq"] <= 8 and tot["nn"] <= 8


@needs_ref
@pytest.mark.parametrize("kind", ["normal", "lattice", "dups"])
def test_reference_fps_kernel_on_its_own_picks_is_the_identity_up_to_the_predicted_round(kind):
    """The property the nested sampling of csrc/sampling.hip relies on, shown on the REFERENCE'S OWN
    kernel (sampling_gpu.cu:69-173 compiled for the CPU): run on a cloud given in the order an earlier run picked
    it, furthest_point_sampling selects 0, 1, 2, ... up to the first round in which a tie is broken differently
    under the new block shape -- exactly the round the verification pass computes."""
    from test_oracle_properties import _first_differing_round
    from oracle import native
    rng = np.random.default_rng(3)
    cloud = _clouds(rng, 1, 1536, kind)[0]
    if kind == "lattice":
        cloud = cloud + rng.integers(0, 2, size=cloud.shape).astype(np.float32) * np.float32(0.0625)
    sel = ref.furthest_point_sampling(cloud[None], 600)[0]
    S = np.ascontiguousarray(cloud[sel])
    for m in (256, 100):
        want = ref.furthest_point_sampling(S[None], m)[0]
        r = _first_differing_round(S, m, native)
        assert np.array_equal(want[:r], np.arange(r))
        assert r >= m or want[r] != r
        S = np.ascontiguousarray(S[want])
