"""Pin the oracle against outputs of the REFERENCE's own Python (tests/golden/*_ref.npz, made
by tests/golden/make_golden.py in the build container)."""
import numpy as np
import torch


def test_meanshift_c_oracle_vs_reference(orc, golden):
    z = golden("meanshift_ref.npz")
    for i in range(int(z["n_cases"])):
        A, bw = z["A%d" % i], float(z["bw%d" % i])
        mi = int(z["max_iter%d" % i]) if ("max_iter%d" % i) in z else 300
        ctr, labels, iters = orc.meanshift_fit(A, bw, mi)
        assert np.abs(ctr - z["ctr%d" % i]).max() < 2e-5, i
        assert abs(iters - int(z["iters%d" % i])) <= 1, (i, iters, int(z["iters%d" % i]))
        assert np.array_equal(labels, z["labels%d" % i]), i


def test_meanshift_torch_port_vs_reference(golden):
    from oracle import torch_port
    z = golden("meanshift_ref.npz")
    torch.set_num_threads(8)
    for i in range(int(z["n_cases"])):
        A, bw = z["A%d" % i], float(z["bw%d" % i])
        mi = int(z["max_iter%d" % i]) if ("max_iter%d" % i) in z else 300
        ctr, labels, it = torch_port.meanshift_fit_dense(torch.from_numpy(A), bw, mi)
        assert np.abs(ctr.numpy() - z["ctr%d" % i]).max() < 1e-6, i
        assert it == int(z["iters%d" % i]), i
        assert np.array_equal(labels.numpy(), z["labels%d" % i]), i


def test_kabsch_oracles_vs_reference(orc, golden):
    from oracle import torch_port
    z = golden("kabsch_ref.npz")
    for i in range(int(z["n_cases"])):
        A, B, T = z["A%d" % i], z["B%d" % i], z["T%d" % i]
        assert np.abs(torch_port.best_fit_transform_np(A, B) - T).max() < 1e-6, i
        Tc = orc.best_fit_transform(A, B)
        if i == 10:
            # planar (rank-2) point set: the rotation about the plane normal's sign is not
            # determined by the data; both solutions map A onto B equally well
            resid = np.abs(A @ Tc[:, :3].T + Tc[:, 3] - B).max()
            assert resid < 1e-5
            continue
        assert np.abs(Tc - T).max() < 2e-5, (i, np.abs(Tc - T).max())
        R = Tc[:, :3]
        assert np.abs(R @ R.T - np.eye(3)).max() < 1e-9 and np.linalg.det(R) > 0


def test_kabsch_icp_test_restated(orc):
    """pvn3d/lib/utils/icp/test.py:24-64 -- the reference's only asserting test -- restated for
    best_fit_transform(B, A) with its constants (N=10, noise 0.01, tol 6 sigma)."""
    rng = np.random.RandomState(0)
    N, noise_sigma, translation, rotation = 10, .01, .1, .1
    A = rng.rand(N, 3)

    def rotation_matrix(axis, theta):
        axis = axis / np.sqrt(np.dot(axis, axis))
        a = np.cos(theta / 2.)
        b, c, d = -axis * np.sin(theta / 2.)
        return np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)],
                         [2*(b*c+a*d), a*a+c*c-b*b-d*d, 2*(c*d-a*b)],
                         [2*(b*d-a*c), 2*(c*d+a*b), a*a+d*d-b*b-c*c]])
    for _ in range(100):
        B = np.copy(A)
        t = rng.rand(3) * translation
        B += t
        R = rotation_matrix(rng.rand(3), rng.rand() * rotation)
        B = np.dot(R, B.T).T
        B += rng.randn(N, 3) * noise_sigma
        T = orc.best_fit_transform(B.astype(np.float32), A.astype(np.float32))
        C = B @ T[:, :3].T + T[:, 3]
        assert np.allclose(C, A, atol=6 * noise_sigma)
        assert np.allclose(T[:, :3].T, R, atol=6 * noise_sigma)


def test_frames_native_oracle_vs_reference_driven(orc, golden):
    """cal_frame_poses(_lm) restatement driven by the C oracle vs the same restatement driven by
    the reference's MeanShiftTorch.fit + best_fit_transform (frames_ref.npz)."""
    from oracle import posecal
    from pvn3d_amd import synth
    z = golden("frames_ref.npz")
    f = synth.synth_frame(frame=0, n_pts=2048, n_obj=2048)
    poses, cls_kps, iters = posecal.cal_frame_poses_lm(
        f["pcld"], f["mask"], f["ctr_of"], f["pred_kp_of"], True, 2, False, f["mesh_kps"],
        return_debug=True)
    assert np.abs(poses[0] - z["lm0_pose"]).max() < 1e-4
    assert np.abs(cls_kps - z["lm0_cls_kps"]).max() < 1e-5
    assert np.abs(iters - z["lm0_iters"]).max() <= 1
    # the recovered pose is the synthetic ground truth up to vote noise
    assert np.abs(poses[0][:, :3] - f["R"]).max() < 2e-2 and np.abs(poses[0][:, 3] - f["t"]).max() < 2e-3
    f = synth.synth_frame(frame=1, n_pts=4096, n_obj=1024)
    poses, cls_kps, iters = posecal.cal_frame_poses_lm(
        f["pcld"], f["mask"], f["ctr_of"], f["pred_kp_of"], True, 2, True, f["mesh_kps"],
        return_debug=True)
    assert np.abs(poses[0] - z["lm1_pose"]).max() < 1e-4
    assert np.abs(cls_kps - z["lm1_cls_kps"]).max() < 1e-5
    y = synth.synth_frame_ycb(frame=2, n_pts=4096, n_obj_total=2000, n_objs=4)
    classes = y["classes"]
    ids, poses, cls_kps, new_mask = posecal.cal_frame_poses(
        y["pcld"], y["mask"], y["ctr_of"], y["pred_kp_of"], True, 22, True,
        lambda c: synth.mesh_kps(classes[c - 1], "ycb", True), y["radius"], return_debug=True)
    assert np.array_equal(ids, z["ycb_ids"])
    assert np.array_equal(new_mask, z["ycb_new_mask"])
    assert np.abs(np.stack(poses, 0) - z["ycb_poses"]).max() < 1e-4
    assert np.abs(cls_kps - z["ycb_cls_kps"]).max() < 1e-5


def test_metric_oracle_vs_reference(golden):
    """oracle/metrics.py (ADD, ADD-S, AUC) against the reference's own Basic_Utils.cal_add_cuda /
    cal_adds_cuda / cal_auc outputs (torch CPU)."""
    from oracle import metrics
    z = golden("metrics_ref.npz")
    for i in range(int(z["n_cases"])):
        pts, gt, pred = z["pts%d" % i], z["gt%d" % i], z["pred%d" % i]
        assert abs(metrics.cal_add(pred, gt, pts) - float(z["add%d" % i])) < 2e-6 * max(1.0, float(z["add%d" % i]))
        assert abs(metrics.cal_adds(pred, gt, pts) - float(z["adds%d" % i])) < 2e-6 * max(1.0, float(z["adds%d" % i]))
    for i in range(int(z["n_auc"])):
        assert abs(metrics.cal_auc(list(z["dis%d" % i])) - float(z["auc%d" % i])) < 1e-9


def test_host_auc_equals_reference(golden):
    """Basic_Utils.cal_auc / VOCap of the package (pure numpy, no GPU needed) vs the reference."""
    import os
    # basic_utils imports the ctypes library at module level; load only the pure functions
    src = open(os.path.join(os.path.dirname(__file__), "..", "pvn3d_amd", "lib", "utils", "basic_utils.py")).read()
    start = src.index("def _accuracy_curve_area(dists, n_total, max_dis=0.1):")
    end = src.index("def best_fit_transform(A, B):")
    ns = {}
    exec("import numpy as np\n" + src[start:end], ns)
    cls_src = src[src.index("    def cal_auc(self, add_dis, max_dis=0.1):"):src.index("    def cal_add_cuda(self, pred_RT, gt_RT, p3ds):")]
    exec("import numpy as np\nclass _U(object):\n" + cls_src, ns)
    z = golden("metrics_ref.npz")
    for i in range(int(z["n_auc"])):
        assert abs(ns["_U"]().cal_auc(list(z["dis%d" % i])) - float(z["auc%d" % i])) < 1e-9


def test_vote_loss_oracle_vs_reference(golden):
    """oracle/loss.py against the reference's own of_l1_loss output and autograd gradient."""
    from oracle import loss as oloss
    z = golden("loss_ref.npz")
    for i in range(int(z["n_cases"])):
        pred, targ, labels = z["pred%d" % i], z["targ%d" % i], z["labels%d" % i]
        out = oloss.of_l1_loss(pred, targ, labels)
        assert np.allclose(out, z["loss%d" % i], rtol=2e-6, atol=1e-7)
        g = oloss.of_l1_loss_grad(pred, targ, labels, z["gout%d" % i])
        assert np.allclose(g, z["gpred%d" % i], rtol=2e-6, atol=1e-9)


def test_focal_loss_vs_reference(golden):
    """pvn3d_amd.lib.loss.FocalLoss (plain torch) against outputs of the reference's FocalLoss."""
    from pvn3d_amd.lib.loss import FocalLoss
    z = golden("loss_ref.npz")
    for i in range(int(z["n_focal"])):
        alpha = z["f_alpha%d" % i]
        alpha = None if alpha.size == 0 else [float(a) for a in alpha]
        fl = FocalLoss(gamma=float(z["f_gamma%d" % i]), alpha=alpha, size_average=bool(z["f_avg%d" % i]))
        out = fl(torch.from_numpy(z["f_logits%d" % i]), torch.from_numpy(z["f_target%d" % i]))
        assert abs(out.item() - float(z["f_out%d" % i])) < 1e-5 * max(1.0, abs(float(z["f_out%d" % i]))), i
