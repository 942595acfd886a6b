"""bench.py's roofline arithmetic against the figures of SURVEY.md section 8(d) (no GPU needed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402


def test_algorithmic_bytes_match_survey():
    alg = bench.algorithmic_bytes_per_frame(12288)
    mb = {k: v / 1e6 for k, v in alg.items()}
    assert abs(mb["ball_query"] - 1.183) < 2e-3
    assert abs(mb["group"] - 69.284) < 2e-3
    assert abs(mb["ball_query"] + mb["group"] - 70.47) < 1e-2          # "70.47 MB/frame for ball_query+group"
    assert abs(mb["fps"] - 0.205) < 2e-3
    assert abs(mb["gather"] - 0.104) < 2e-3
    assert abs(mb["three_nn"] - 0.616) < 2e-3
    assert abs(mb["three_interpolate"] - 27.12) < 1e-2


def test_mlp_flops_match_survey():
    from pvn3d_amd.lib.pointnet2_msg import Pointnet2MSG
    net = Pointnet2MSG(input_channels=6)
    sa, fp = bench.mlp_flops_per_frame(net, 1.0)
    assert abs(sa / 1e9 - 12.90) < 0.01          # "12.90 GFLOP/frame (SA)"
    assert abs(fp / 1e9 - 4.55) < 0.01           # "4.55 GFLOP/frame (FP)"


def test_peaks_are_the_dense_figures():
    assert bench.PEAK_HBM_GBS == 8000.0
    assert bench.PEAK_FP32_MFMA_TFLOPS == 157.3 and bench.PEAK_FP32_VALU_TFLOPS == 157.3


def test_bench_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        return
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py"), "--steps", "1"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
