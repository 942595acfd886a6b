#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own Python.

Runs only in the build container (needs /root/reference); the fixtures it writes are committed
so that tests never read /root/reference.  What is pinned by real reference outputs:
  meanshift_ref.npz : MeanShiftTorch(bandwidth).fit(A) of pvn3d/lib/utils/meanshift_pytorch.py
                      (module loaded file-level; cv2 / sklearn / neupeak stubbed -- they are only
                      used by its visual test functions), incl. the iteration count `it`
                      (read from the frame's locals with sys.settrace).
  kabsch_ref.npz    : best_fit_transform(A,B) of pvn3d/lib/utils/basic_utils.py (file-level load,
                      cv2 / plyfile / ip_basic stubbed).
  frames_ref.npz    : cal_frame_poses_lm / cal_frame_poses restated (oracle/posecal.py) but
                      DRIVEN BY the reference's MeanShiftTorch.fit and best_fit_transform.
  metrics_ref.npz   : Basic_Utils.cal_add_cuda / cal_adds_cuda / cal_auc (+ VOCap) of
                      pvn3d/lib/utils/basic_utils.py on CPU tensors (unbound calls).
  loss_ref.npz      : of_l1_loss value and autograd gradient, FocalLoss value, of pvn3d/lib/loss.py
                      (file-level load, lib.utils.meanshift_pytorch stubbed with the loaded module).
What cannot be pinned by the reference (CUDA-only native ops, no nvcc / NVIDIA GPU here):
  native_oracle.npz : outputs of the C oracle for the pointnet2 ops on seeded inputs --
                      regression vectors, labelled "oracle-generated".
Usage: python tests/golden/make_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference/pvn3d"
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    for n in ["cv2", "plyfile", "neupeak", "neupeak.utils", "neupeak.utils.webcv2",
              "lib", "lib.utils", "lib.utils.ip_basic", "lib.utils.ip_basic.ip_basic",
              "lib.utils.ip_basic.ip_basic.depth_map_utils_ycb", "lib.utils.ip_basic.ip_basic.vis_utils"]:
        if n not in sys.modules:
            _stub(n)
    sys.modules["neupeak.utils.webcv2"].imshow = None
    sys.modules["neupeak.utils.webcv2"].waitKey = None
    sys.modules["plyfile"].PlyData = None
    sys.modules["lib.utils.ip_basic.ip_basic"].depth_map_utils_ycb = sys.modules["lib.utils.ip_basic.ip_basic.depth_map_utils_ycb"]
    sys.modules["lib.utils.ip_basic.ip_basic"].vis_utils = sys.modules["lib.utils.ip_basic.ip_basic.vis_utils"]

    def _load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ms = _load("ref_meanshift_pytorch", os.path.join(REF, "lib/utils/meanshift_pytorch.py"))
    bu = _load("ref_basic_utils", os.path.join(REF, "lib/utils/basic_utils.py"))
    return ms, bu


def ref_fit_with_iters(ms_mod, A_np, bw, max_iter=300):
    """Run the reference fit and read its local `it` when the frame returns."""
    box = {}

    def tracer(frame, event, arg):
        if frame.f_code.co_name == "fit":
            def local(frame, event, arg):
                if event == "return":
                    box["it"] = frame.f_locals.get("it")
                return local
            return local
        return None
    ms = ms_mod.MeanShiftTorch(bandwidth=bw, max_iter=max_iter)
    sys.settrace(tracer)
    try:
        ctr, labels = ms.fit(torch.from_numpy(A_np))
    finally:
        sys.settrace(None)
    return ctr.numpy().astype(np.float32), labels.numpy().astype(bool), int(box["it"])


def vote_cloud(rng, n, sig_in, sig_out, out_frac, centre):
    is_out = rng.random(n) < out_frac
    eps = rng.normal(size=(n, 3)) * np.where(is_out, sig_out, sig_in)[:, None]
    return (np.asarray(centre)[None] + eps).astype(np.float32)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ms_mod, bu_mod = load_reference()
    from oracle import posecal
    from pvn3d_amd import synth

    # ------------------------------------------------------------------ mean shift
    rng = np.random.default_rng(20260925)
    cases = [
        dict(n=1, sig_in=0.005, sig_out=0.05, out=0.0, bw=0.08),
        dict(n=2, sig_in=0.005, sig_out=0.05, out=0.0, bw=0.08),
        dict(n=64, sig_in=0.005, sig_out=0.05, out=0.1, bw=0.08),
        dict(n=300, sig_in=0.005, sig_out=0.05, out=0.1, bw=0.08),
        dict(n=512, sig_in=0.01, sig_out=0.3, out=0.1, bw=0.08),
        dict(n=777, sig_in=0.005, sig_out=0.05, out=0.1, bw=0.05),
        dict(n=1024, sig_in=0.005, sig_out=0.0, out=0.0, bw=0.08),
        dict(n=1500, sig_in=0.02, sig_out=0.3, out=0.3, bw=0.08),
        dict(n=2048, sig_in=0.005, sig_out=0.05, out=0.1, bw=0.08),
    ]
    ms_out = {}
    for i, c in enumerate(cases):
        A = vote_cloud(rng, c["n"], c["sig_in"], c["sig_out"], c["out"], [0.05, -0.02, 0.9])
        ctr, labels, it = ref_fit_with_iters(ms_mod, A, c["bw"])
        ms_out["A%d" % i] = A
        ms_out["ctr%d" % i] = ctr
        ms_out["labels%d" % i] = labels
        ms_out["iters%d" % i] = np.int64(it)
        ms_out["bw%d" % i] = np.float64(c["bw"])
        print("meanshift case %d n=%d iters=%d" % (i, c["n"], it))
    # two-cluster case: the pick must be the denser cluster
    A = np.concatenate([vote_cloud(rng, 300, 0.004, 0, 0, [0.0, 0.0, 0.8]),
                        vote_cloud(rng, 200, 0.004, 0, 0, [0.3, 0.1, 0.9])], 0)
    A = A[rng.permutation(len(A))]
    i = len(cases)
    ctr, labels, it = ref_fit_with_iters(ms_mod, A, 0.08)
    ms_out.update({"A%d" % i: A, "ctr%d" % i: ctr, "labels%d" % i: labels,
                   "iters%d" % i: np.int64(it), "bw%d" % i: np.float64(0.08)})
    # max_iter cap: it must stop at max_iter+1
    i += 1
    A = vote_cloud(rng, 400, 0.02, 0.3, 0.3, [0.05, -0.02, 0.9])
    ctr, labels, it = ref_fit_with_iters(ms_mod, A, 0.08, max_iter=5)
    ms_out.update({"A%d" % i: A, "ctr%d" % i: ctr, "labels%d" % i: labels,
                   "iters%d" % i: np.int64(it), "bw%d" % i: np.float64(0.08),
                   "max_iter%d" % i: np.int64(5)})
    print("cap case iters", it)
    ms_out["n_cases"] = np.int64(i + 1)
    np.savez_compressed(os.path.join(HERE, "meanshift_ref.npz"), **ms_out)

    # ------------------------------------------------------------------ kabsch
    kb = {}
    kps = synth.mesh_kps("ape", "lm", True)
    nk = 0
    for trial in range(12):
        R = synth.random_rotation(rng)
        t = rng.normal(size=3) * 0.3
        if trial % 3 == 0:
            A = kps
        elif trial % 3 == 1:
            A = rng.random((10, 3)).astype(np.float32)     # lib/utils/icp/test.py style
        else:
            A = synth.mesh_kps("cat", "lm", False)
        noise = 0.0 if trial < 6 else 0.003
        B = (A.astype(np.float64) @ R.T + t + rng.normal(size=A.shape) * noise).astype(np.float32)
        if trial == 11:                                    # force the reflection branch
            B = B * np.array([1, 1, -1], np.float32)
        if trial == 10:                                    # planar (rank-2) point set
            A = A.copy(); A[:, 2] = 0
            B = (A.astype(np.float64) @ R.T + t).astype(np.float32)
        T = bu_mod.best_fit_transform(A, B)
        kb["A%d" % nk] = A.astype(np.float32); kb["B%d" % nk] = B; kb["T%d" % nk] = T
        nk += 1
    kb["n_cases"] = np.int64(nk)
    np.savez_compressed(os.path.join(HERE, "kabsch_ref.npz"), **kb)
    print("kabsch cases", nk)

    # ------------------------------------------------------------------ ADD / ADD-S / AUC
    # Basic_Utils.cal_add_cuda / cal_adds_cuda / cal_auc use `self` for nothing: call unbound.
    rng_m = np.random.default_rng(20260925)      # own stream: the other fixtures stay bit-identical
    mt = {}
    nm = 0
    for trial in range(8):
        n = [64, 257, 500, 1000, 1500, 2000, 333, 1][trial]
        pts = (rng_m.normal(size=(n, 3)) * [0.04, 0.06, 0.03]).astype(np.float32)
        Rg = synth.random_rotation(rng_m); tg = rng_m.normal(size=3) * 0.3 + [0, 0, 0.9]
        ang = [0.0, 0.01, 0.05, 0.2, 3.1, 0.02, 1.0, 0.3][trial]
        ax = rng_m.normal(size=3); ax /= np.linalg.norm(ax)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dR = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
        Rp = dR @ Rg; tp = tg + rng_m.normal(size=3) * [0.0, 0.002, 0.01, 0.05, 0.1, 0.0, 0.02, 0.01][trial]
        gt = np.concatenate([Rg, tg[:, None]], 1).astype(np.float32)
        pr = np.concatenate([Rp, tp[:, None]], 1).astype(np.float32)
        if trial == 5:
            pr = np.zeros((3, 4), np.float32)           # "no prediction" pose of eval_metric (:124)
        add = bu_mod.Basic_Utils.cal_add_cuda(None, torch.from_numpy(pr), torch.from_numpy(gt), torch.from_numpy(pts))
        adds = bu_mod.Basic_Utils.cal_adds_cuda(None, torch.from_numpy(pr), torch.from_numpy(gt), torch.from_numpy(pts))
        mt["pts%d" % nm] = pts; mt["gt%d" % nm] = gt; mt["pred%d" % nm] = pr
        mt["add%d" % nm] = np.float32(add.item()); mt["adds%d" % nm] = np.float32(adds.item())
        nm += 1
    mt["n_cases"] = np.int64(nm)
    na = 0
    for trial in range(6):
        n = [1, 5, 40, 200, 17, 3][trial]
        dis = np.abs(rng_m.normal(size=n) * [0.01, 0.05, 0.03, 0.08, 0.5, 0.0][trial]).tolist()
        if trial == 4:
            dis[0] = 0.1                                  # boundary: D > max_dis is strict
        mt["dis%d" % na] = np.asarray(dis, np.float64)
        mt["auc%d" % na] = np.float64(bu_mod.Basic_Utils.cal_auc(None, list(dis)))
        na += 1
    mt["n_auc"] = np.int64(na)
    np.savez_compressed(os.path.join(HERE, "metrics_ref.npz"), **mt)
    print("metric cases", nm, na)

    # ------------------------------------------------------------------ vote loss (lib/loss.py:45-73)
    if "lib.utils.meanshift_pytorch" not in sys.modules:
        _stub("lib.utils.meanshift_pytorch", MeanShiftTorch=ms_mod.MeanShiftTorch)
    spec = importlib.util.spec_from_file_location("ref_loss", os.path.join(REF, "lib/loss.py"))
    loss_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(loss_mod)
    rng_l = np.random.default_rng(20260926)
    lz = {}
    nl = 0
    for (bs, K, N, frac) in [(2, 8, 500, 0.3), (1, 1, 257, 1.0), (3, 8, 64, 0.0), (2, 3, 1000, 0.05)]:
        pred = torch.from_numpy(rng_l.normal(size=(bs, K, N, 3)).astype(np.float32)).requires_grad_(True)
        targ = torch.from_numpy(rng_l.normal(size=(bs, N, K, 3)).astype(np.float32))
        labels = torch.from_numpy((rng_l.random((bs, N, 1)) < frac).astype(np.int64) * rng_l.integers(1, 5, (bs, N, 1)))
        if nl == 0:
            with torch.no_grad():
                pred[0, 0, :7] = targ[0, :7, 0]                      # exact zeros: sign(0) = 0
        out = loss_mod.of_l1_loss(pred, targ, labels)
        g = torch.from_numpy(rng_l.normal(size=tuple(out.shape)).astype(np.float32))
        out.backward(g)
        lz["pred%d" % nl] = pred.detach().numpy(); lz["targ%d" % nl] = targ.numpy()
        lz["labels%d" % nl] = labels.numpy(); lz["loss%d" % nl] = out.detach().numpy()
        lz["gout%d" % nl] = g.numpy(); lz["gpred%d" % nl] = pred.grad.numpy()
        nl += 1
    lz["n_cases"] = np.int64(nl)
    # FocalLoss (lib/loss.py:13-42) -- plain-torch restatement in pvn3d_amd/lib/loss.py
    nf = 0
    for (shape, gamma, alpha, avg) in [((6, 4), 2, None, True), ((2, 3, 4, 5), 2, [0.2, 0.3, 0.5], True),
                                       ((50, 2), 0, 0.25, False), ((3, 22, 7), 1, None, True)]:
        logits = torch.from_numpy(rng_l.normal(size=shape).astype(np.float32))
        C = shape[1]
        tshape = (shape[0],) + tuple(shape[2:])
        target = torch.from_numpy(rng_l.integers(0, C, size=tshape).astype(np.int64))
        out = loss_mod.FocalLoss(gamma=gamma, alpha=alpha, size_average=avg)(logits, target)
        lz["f_logits%d" % nf] = logits.numpy(); lz["f_target%d" % nf] = target.numpy()
        lz["f_gamma%d" % nf] = np.float64(gamma); lz["f_avg%d" % nf] = np.bool_(avg)
        lz["f_alpha%d" % nf] = np.asarray([] if alpha is None else ([alpha, 1 - alpha] if isinstance(alpha, float) else alpha), np.float64)
        lz["f_out%d" % nf] = np.float32(out.item())
        nf += 1
    lz["n_focal"] = np.int64(nf)
    np.savez_compressed(os.path.join(HERE, "loss_ref.npz"), **lz)
    print("loss cases", nl)

    # ------------------------------------------------------------------ whole frames
    def ref_fit(A, bw):
        if len(A) == 0:
            raise RuntimeError("empty fit")
        ctr, labels, it = ref_fit_with_iters(ms_mod, np.ascontiguousarray(A, np.float32), bw)
        return ctr, labels, it
    fr = {}
    # config 1: N=2048, one object, all points on the object
    f = synth.synth_frame(frame=0, n_pts=2048, n_obj=2048)
    poses, cls_kps, iters = posecal.cal_frame_poses_lm(
        f["pcld"], f["mask"], f["ctr_of"], f["pred_kp_of"], True, 2, False, f["mesh_kps"],
        fit=ref_fit, bft=bu_mod.best_fit_transform, return_debug=True)
    fr.update(lm0_pose=poses[0], lm0_cls_kps=cls_kps, lm0_iters=iters)
    print("lm frame0 iters", iters)
    # LineMOD-style frame with the centre-cluster filter on, n_obj=1024 of 4096
    f = synth.synth_frame(frame=1, n_pts=4096, n_obj=1024)
    poses, cls_kps, iters = posecal.cal_frame_poses_lm(
        f["pcld"], f["mask"], f["ctr_of"], f["pred_kp_of"], True, 2, True, f["mesh_kps"],
        fit=ref_fit, bft=bu_mod.best_fit_transform, return_debug=True)
    fr.update(lm1_pose=poses[0], lm1_cls_kps=cls_kps, lm1_iters=iters)
    print("lm frame1 iters", iters)
    # YCB-style multi-instance frame
    y = synth.synth_frame_ycb(frame=2, n_pts=4096, n_obj_total=2000, n_objs=4)
    classes = y["classes"]
    ids, poses, cls_kps, new_mask = posecal.cal_frame_poses(
        y["pcld"], y["mask"], y["ctr_of"], y["pred_kp_of"], True, 22, True,
        lambda c: synth.mesh_kps(classes[c - 1], "ycb", True), y["radius"],
        fit=ref_fit, bft=bu_mod.best_fit_transform, return_debug=True)
    fr.update(ycb_ids=ids, ycb_poses=np.stack(poses, 0), ycb_cls_kps=cls_kps, ycb_new_mask=new_mask)
    np.savez_compressed(os.path.join(HERE, "frames_ref.npz"), **fr)
    print("frames done; ycb ids", ids)

    # ------------------------------------------------------------------ native ops (oracle)
    from oracle import native as orc
    nat = {}
    g = np.random.default_rng(7)
    cloud, _ = synth.synth_cloud(g, 4096, wrap_pad=0.1)
    xyz = cloud[None]
    nat["xyz"] = xyz
    fps = orc.furthest_point_sampling(xyz, 512)
    nat["fps"] = fps
    new_xyz = np.take_along_axis(xyz, fps[..., None].astype(np.int64).repeat(3, -1), 1)
    nat["bq_r0025_16"] = orc.ball_query(new_xyz, xyz, 0.025, 16)
    nat["bq_r005_32"] = orc.ball_query(new_xyz, xyz, 0.05, 32)
    d2, idx = orc.three_nn(xyz, new_xyz)
    nat["nn_d2"] = d2; nat["nn_idx"] = idx
    np.savez_compressed(os.path.join(HERE, "native_oracle.npz"), **nat)
    print("native vectors done")


if __name__ == "__main__":
    main()
