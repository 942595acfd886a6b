"""Property tests (hypothesis) of the C oracle for the pointnet2 ops against brute-force numpy:
random sizes, radii and clouds with many exactly duplicated points -- the situations in which the
reference's 'first nsample in ascending k' / 'strict <, earlier k wins' rules actually matter."""
import numpy as np
from hypothesis import given, settings, strategies as st

from test_oracle_native import brute_ball_query, d2_f32


def _cloud(rng, n, dup_frac):
    x = (rng.random((n, 3)) * 0.3).astype(np.float32)
    n_dup = int(n * dup_frac)
    if n_dup:
        x[rng.integers(0, n, n_dup)] = x[rng.integers(0, n, n_dup)]      # exact duplicates -> exact ties
    return x


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 300), st.integers(1, 40), st.integers(1, 24),
       st.floats(0.005, 0.3), st.sampled_from([0.0, 0.3, 0.9]))
def test_ball_query_property(orc, seed, n, m, nsample, radius, dup):
    rng = np.random.default_rng(seed)
    xyz = _cloud(rng, n, dup)
    new_xyz = np.ascontiguousarray(xyz[rng.integers(0, n, m)] + (rng.random((m, 3)) < 0.5) * np.float32(radius * 0.5))
    new_xyz = new_xyz.astype(np.float32)
    got = orc.ball_query(new_xyz[None], xyz[None], radius, nsample)[0]
    assert np.array_equal(got, brute_ball_query(new_xyz, xyz, radius, nsample))


@settings(max_examples=30, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 200), st.integers(3, 120), st.sampled_from([0.0, 0.5, 0.95]))
def test_three_nn_property(orc, seed, n, m, dup):
    rng = np.random.default_rng(seed)
    known = _cloud(rng, m, dup)
    unknown = np.concatenate([known[rng.integers(0, m, n // 2)], _cloud(rng, n - n // 2, 0.0)], 0).astype(np.float32)
    d2, idx = orc.three_nn(unknown[None], known[None])
    dm = d2_f32(unknown, known)
    order = np.argsort(dm, axis=1, kind="stable")[:, :3]           # ties -> smaller index, like strict '<'
    assert np.array_equal(idx[0], order.astype(np.int32))
    assert np.array_equal(d2[0], np.take_along_axis(dm, order, 1))


@settings(max_examples=20, deadline=None)
@given(st.integers(0, 2 ** 31 - 1), st.integers(2, 60), st.integers(1, 9), st.integers(1, 30), st.integers(1, 6))
def test_group_grad_is_the_transpose_of_group(orc, seed, n, c, m, ns):
    """<group(points), g> == <points, group_grad(g)> (adjoint identity, float64 accumulation)."""
    rng = np.random.default_rng(seed)
    pts = rng.normal(size=(1, c, n)).astype(np.float32)
    idx = rng.integers(0, n, size=(1, m, ns)).astype(np.int32)
    g = rng.normal(size=(1, c, m, ns)).astype(np.float32)
    lhs = float((orc.group_points(pts, idx).astype(np.float64) * g).sum())
    rhs = float((pts.astype(np.float64) * orc.group_points_grad(g, idx, n)).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


# ---- the property behind the nested sampling of csrc/sampling.hip, on the CPU oracle ----------
def _fps_prio(k, n, orc):
    """Tie-break priority of point k in the reference's block of opt_n_threads(n) threads (smaller wins)."""
    bs = orc.opt_n_threads(n)
    L = bs.bit_length() - 1
    Q = (n + bs - 1) // bs
    r = int(format(k & (bs - 1), "0%db" % L)[::-1], 2) if L else 0
    return r * Q + (k >> L)


def _first_differing_round(S, m, orc):
    """numpy restatement of fps_nest_verify_kernel for one following level: S = a cloud in the order an FPS run
    picked it; the first round t < m of a run ON S that does not select t (m if there is none)."""
    n = len(S)
    skipped = ((S[:, 0] * S[:, 0] + S[:, 1] * S[:, 1]) + S[:, 2] * S[:, 2]).astype(np.float64) <= 1e-3
    run = np.full(n, 1e10, np.float32)
    for t in range(1, m):
        d = S - S[t - 1]
        run = np.minimum(((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]).astype(np.float32), run)
        dt = run[t]
        if not dt > 0 or skipped[t]:
            return t
        for k in np.nonzero((run[t + 1:] >= dt) & ~skipped[t + 1:])[0] + t + 1:
            if run[k] > dt or _fps_prio(int(k), n, orc) < _fps_prio(t, n, orc):
                return t
    return m


def test_fps_of_an_fps_ordered_cloud_is_the_identity_up_to_the_first_broken_tie(orc):
    """FPS is greedy: on the picks of a run, in pick order, the next pyramid level's run selects 0, 1, 2, ...
    until a tie is broken differently under its own block shape -- the round the verification predicts."""
    g = np.random.default_rng(7)
    side = 14
    lattice = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    lattice = lattice[g.permutation(len(lattice))[:2048]] * np.float32(0.03125) + np.float32(0.25)
    generic = (g.normal(size=(2048, 3)) * 0.2).astype(np.float32) + np.float32(0.7)
    dups = generic.copy()
    dups[1024:1536] = dups[:512]
    seen_identity = seen_partial = False
    for cloud in (generic, dups, lattice):
        sel = orc.furthest_point_sampling(cloud[None], 700)[0]
        S = np.ascontiguousarray(cloud[sel])
        for m in (300, 128):
            want = orc.furthest_point_sampling(S[None], m)[0]
            r = _first_differing_round(S, m, orc)
            assert np.array_equal(want[:r], np.arange(r))
            if r < m:
                assert want[r] != r
                seen_partial = True
            else:
                seen_identity = True
            S = np.ascontiguousarray(S[want])
    assert seen_identity and seen_partial
