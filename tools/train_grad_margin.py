#!/usr/bin/env python3
"""Margins of tests/test_gpu_train_mlp.py::test_full_size_training_gradients_vs_torch_autocast over repeated runs
(the torch fp32 / autocast legs accumulate with atomics, so the yardstick itself moves from run to run)."""
import copy
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from pvn3d_amd import train_step as ts  # noqa: E402
from pvn3d_amd.lib.pointnet2_utils import _train_mlp  # noqa: E402

dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for r in range(reps):
    torch.manual_seed(0)
    batch = ts.synthetic_batch(2, 12288, dev, seed_base=60, n_obj=3072)
    base = ts.PointVoteNet().to(dev).train()
    res = {}
    for mode in ("fp32", "autocast", "fused"):
        model = copy.deepcopy(base)
        _train_mlp.TRAIN_FUSED = mode == "fused"
        try:
            if mode == "fp32":
                kp, ctr = model(batch["pc"])
            else:
                with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                    kp, ctr = model(batch["pc"])
            loss = ts.vote_loss(kp.float(), ctr.float(), batch["kp_targ_ofst"], batch["ctr_targ_ofst"], batch["labels"])
            loss.backward()
        finally:
            _train_mlp.TRAIN_FUSED = "auto"
        res[mode] = (loss.item(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None})
    l32, g32 = res["fp32"]
    floor = 1e-3 * max(g.norm().item() for g in g32.values())
    err = {m: {n: (res[m][1][n] - g).norm().item() / max(g.norm().item(), floor) for n, g in g32.items()}
           for m in ("autocast", "fused")}
    worst = max(g32, key=lambda n: err["fused"][n] / (1.6 * err["autocast"][n] + 2e-2))
    mean = {m: float(np.mean(list(err[m].values()))) for m in err}
    print("run %2d loss f32 %.6f ac %.6f fused %.6f | worst %-44s fused %.4f ac %.4f margin %.3f | mean fused %.4f ac %.4f ratio %.3f"
          % (r, l32, res["autocast"][0], res["fused"][0], worst, err["fused"][worst], err["autocast"][worst],
             err["fused"][worst] / (1.6 * err["autocast"][worst] + 2e-2), mean["fused"], mean["autocast"],
             mean["fused"] / (1.15 * mean["autocast"] + 5e-3)), flush=True)
