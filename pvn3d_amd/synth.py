"""Seeded synthetic frames for parity tests and bench.py.

No dataset or checkpoint ships with the reference, so frames are synthesised with the shapes
and statistics of the reference's input pipeline:
  * cloud: a 480x640 depth map (tilted plane at ~1 m + an object blob at 0.85-0.95 m + 1 mm noise)
    back-projected with the LineMOD intrinsics (pvn3d/common.py:138-140) the way ``dpt_2_cld``
    does (pvn3d/lib/utils/basic_utils.py:381-399), then ``n_pts`` pixels drawn by a shuffled
    choice as in pvn3d/datasets/linemod/linemod_dataset.py:258-262; ``wrap_pad`` reproduces
    the ``np.pad(..., 'wrap')`` duplicate padding (:264) that makes FPS ties common.
  * votes: ``pred_kp_of[k,i] = pcld[i] - (R kp_k + t) + eps`` with eps ~ N(0,(5 mm)^2) for 90 %
    of the points and N(0,(50 mm)^2) for 10 % outliers; same for the centre offset.
numpy only; every array is float32/int64 like the tensors ``cal_frame_poses`` receives.
"""
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "obj_kps.npz")
_KPS = None

LM_K = np.array([[572.4114, 0., 325.2611], [0., 573.57043, 242.04899], [0., 0., 1.]])


def obj_kps():
    """dict of the per-object keypoint fixtures (tools/import_obj_kps.py)."""
    global _KPS
    if _KPS is None:
        with np.load(_DATA) as z:
            _KPS = {k: z[k] for k in z.files}
    return _KPS


def mesh_kps(name, ds="lm", use_ctr=True):
    """(K,3) farthest keypoints, plus centre = mean(corners) appended when use_ctr
    (Basic_Utils.get_kps / get_ctr, pvn3d/lib/utils/basic_utils.py:541-595)."""
    z = obj_kps()
    kps = z["%s/%s/farthest" % (ds, name)].astype(np.float32)
    if use_ctr:
        ctr = z["%s/%s/corners" % (ds, name)].astype(np.float32).mean(0)
        kps = np.concatenate([kps, ctr.reshape(1, 3)], 0)
    return kps


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    a, b, c, d = q
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])


def synth_cloud(rng, n_pts=12288, wrap_pad=0.0, obj_uv=(320.0, 240.0)):
    """(n_pts,3) float32 camera-frame cloud + the pixel coords it came from."""
    H, W = 480, 640
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    tilt = np.deg2rad(rng.uniform(-15, 15, size=2))
    z = 1.0 + np.tan(tilt[0]) * (u - W / 2) / LM_K[0, 0] + np.tan(tilt[1]) * (v - H / 2) / LM_K[1, 1]
    r2 = ((u - obj_uv[0]) / 60.0) ** 2 + ((v - obj_uv[1]) / 60.0) ** 2
    blob = r2 < 1.0
    z = np.where(blob, 0.95 - 0.10 * np.sqrt(np.clip(1.0 - r2, 0, 1)), z)
    z = z + rng.normal(scale=1e-3, size=z.shape)
    x = (u - LM_K[0, 2]) * z / LM_K[0, 0]
    y = (v - LM_K[1, 2]) * z / LM_K[1, 1]
    cld = np.stack([x, y, z], -1).reshape(-1, 3)
    n_uniq = int(round(n_pts * (1.0 - wrap_pad)))
    choose = rng.permutation(H * W)[:n_uniq]
    if n_uniq < n_pts:  # np.pad(choose, (0, n_pts-len), 'wrap')
        choose = np.pad(choose, (0, n_pts - n_uniq), "wrap")
    return cld[choose].astype(np.float32), choose


def synth_frame(frame=0, n_pts=12288, n_obj=3072, obj="ape", ds="lm", n_kps=8,
                outlier_frac=0.10, sig_in=0.005, sig_out=0.05, wrap_pad=0.0, seed=1234):
    """One LineMOD-style frame: dict(pcld, mask, ctr_of, pred_kp_of, R, t, mesh_kps, feats)."""
    rng = np.random.default_rng(seed + frame)
    pcld, _ = synth_cloud(rng, n_pts, wrap_pad)
    kps_ctr = mesh_kps(obj, ds, use_ctr=True)
    R = random_rotation(rng)
    t = np.array([0.05, -0.02, 0.9])
    tgt = (kps_ctr.astype(np.float64) @ R.T + t)  # (K+1,3) camera-frame keypoints + centre
    # object points: the n_obj points closest to the projected object centre
    d = np.linalg.norm(pcld[:, :2] / pcld[:, 2:3] - (tgt[-1, :2] / tgt[-1, 2]), axis=1)
    order = np.argsort(d, kind="stable")
    mask = np.zeros(n_pts, np.int64)
    mask[order[:n_obj]] = 1
    is_out = rng.random((n_kps + 1, n_pts)) < outlier_frac
    eps = rng.normal(size=(n_kps + 1, n_pts, 3)) * np.where(is_out, sig_out, sig_in)[..., None]
    offs = pcld[None].astype(np.float64) - tgt[:, None, :] + eps  # pcld - offs = tgt - eps
    feats = rng.normal(size=(6, n_pts)).astype(np.float32)
    return dict(pcld=pcld, mask=mask, ctr_of=offs[n_kps:].astype(np.float32),
                pred_kp_of=offs[:n_kps].astype(np.float32), R=R, t=t,
                mesh_kps=kps_ctr, feats=feats)


def synth_frame_ycb(frame=0, n_pts=12288, n_obj_total=6144, n_objs=5, n_kps=8, seed=1234,
                    outlier_frac=0.10, sig_in=0.005, sig_out=0.05):
    """YCB-style multi-instance frame (config 3): ``n_objs`` classes, 21-class mask."""
    rng = np.random.default_rng(seed + frame)
    pcld, _ = synth_cloud(rng, n_pts)
    z = obj_kps()
    classes = list(z["ycb_classes"])
    cls_ids = np.sort(rng.choice(np.arange(1, len(classes) + 1), size=n_objs, replace=False))
    mask = np.zeros(n_pts, np.int64)
    per = n_obj_total // n_objs
    perm = rng.permutation(n_pts)
    ctr_of = np.zeros((1, n_pts, 3), np.float64)
    kp_of = np.zeros((n_kps, n_pts, 3), np.float64)
    poses = {}
    # background votes: far-away noise, never selected (mask 0)
    ctr_of[0] = rng.normal(scale=0.3, size=(n_pts, 3))
    kp_of[:] = rng.normal(scale=0.3, size=(n_kps, n_pts, 3))
    for i, cid in enumerate(cls_ids):
        sel = perm[i * per:(i + 1) * per]
        mask[sel] = cid
        kc = mesh_kps(classes[cid - 1], "ycb", use_ctr=True).astype(np.float64)
        R = random_rotation(rng)
        t = np.array([-0.3 + 0.15 * i, 0.1 * ((i % 2) * 2 - 1), 0.8 + 0.05 * i])
        tgt = kc @ R.T + t
        is_out = rng.random((n_kps + 1, per)) < outlier_frac
        eps = rng.normal(size=(n_kps + 1, per, 3)) * np.where(is_out, sig_out, sig_in)[..., None]
        offs = pcld[sel][None].astype(np.float64) - tgt[:, None, :] + eps
        kp_of[:, sel] = offs[:n_kps]
        ctr_of[0, sel] = offs[n_kps]
        poses[int(cid)] = (R, t)
    return dict(pcld=pcld, mask=mask, ctr_of=ctr_of.astype(np.float32),
                pred_kp_of=kp_of.astype(np.float32), poses=poses, cls_ids=cls_ids,
                classes=classes, radius=z["ycb_radius"])
