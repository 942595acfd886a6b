"""Host-side mirror of the reference interface: argument contract, module tree, synth data."""
import numpy as np
import pytest
import torch


def test_ext_rejects_cpu_tensors_like_reference():
    from pvn3d_amd.lib.pointnet2_utils import _ext
    xyz = torch.zeros(1, 8, 3)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.furthest_point_sampling(xyz, 4)
    with pytest.raises(RuntimeError, match="CPU not supported"):
        _ext.ball_query(xyz, xyz, 0.1, 4)
    with pytest.raises(RuntimeError, match="contiguous"):
        _ext.ball_query(xyz.transpose(1, 2), xyz, 0.1, 4)
    with pytest.raises(RuntimeError, match="float"):
        _ext.three_nn(xyz.double(), xyz)
    with pytest.raises(RuntimeError, match="int"):
        _ext.gather_points(torch.zeros(1, 3, 8), torch.zeros(1, 4, dtype=torch.int64))


def test_ext_exports_the_nine_reference_ops():
    from pvn3d_amd.lib.pointnet2_utils import _ext
    for n in ["gather_points", "gather_points_grad", "furthest_point_sampling", "three_nn",
              "three_interpolate", "three_interpolate_grad", "ball_query", "group_points",
              "group_points_grad"]:
        assert callable(getattr(_ext, n))


def test_sa_fp_module_state_dict_layout():
    from pvn3d_amd.lib.pointnet2_utils.pointnet2_modules import PointnetSAModuleMSG, PointnetFPModule
    sa = PointnetSAModuleMSG(npoint=16, radii=[0.1, 0.2], nsamples=[4, 8],
                             mlps=[[6, 16, 16, 32], [6, 32, 32, 64]], use_xyz=True)
    keys = set(sa.state_dict().keys())
    assert "mlps.0.layer0.conv.weight" in keys
    assert "mlps.1.layer2.normlayer.bn.running_var" in keys
    assert sa.state_dict()["mlps.0.layer0.conv.weight"].shape == (16, 9, 1, 1)  # +3 for xyz
    assert not any(k.endswith("conv.bias") for k in keys)                       # bias off with BN
    fp = PointnetFPModule(mlp=[32, 16, 16])
    assert "mlp.layer1.normlayer.bn.weight" in fp.state_dict()
    assert sa.groupers[1].radius == 0.2 and sa.groupers[1].nsample == 8 and sa.npoint == 16


def test_synth_is_seeded_and_shaped():
    from pvn3d_amd import synth
    a = synth.synth_frame(frame=3, n_pts=1024, n_obj=256)
    b = synth.synth_frame(frame=3, n_pts=1024, n_obj=256)
    assert np.array_equal(a["pred_kp_of"], b["pred_kp_of"])
    assert a["pcld"].shape == (1024, 3) and a["pcld"].dtype == np.float32
    assert a["pred_kp_of"].shape == (8, 1024, 3) and a["ctr_of"].shape == (1, 1024, 3)
    assert int((a["mask"] == 1).sum()) == 256
    assert a["mesh_kps"].shape == (9, 3)
    c, choose = synth.synth_cloud(np.random.default_rng(0), 1000, wrap_pad=0.1)
    assert len(np.unique(choose)) == 900  # 'wrap' padding duplicates


def test_shard_range_partitions():
    from pvn3d_amd.sharding import shard_range
    for n in [0, 1, 7, 64, 65]:
        for ws in [1, 2, 3, 8]:
            spans = [shard_range(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
