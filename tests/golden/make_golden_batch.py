#!/usr/bin/env python3
"""Generate tests/golden/pointnet2msg_batch_ref.npz: the reference's OWN `Pointnet2MSG` (lib/pvn3d.py:46-154) on a
BATCH of 8 seeded N = 12 288 frames -- the fixture behind the end-to-end pin of the 64-frame dispatch that bench.py
times (nested FPS, geometry stream, pre-contraction, split-bf16 chains, split-GEMM FP levels).

What runs (CPU, build container only; needs /root/reference + g++): exactly what make_golden_modules.py runs -- the
reference's pointnet2_modules.py / pointnet2_utils.py / pytorch_utils.py / Pointnet2MSG imported file-level, over
oracle/_ref (= the reference's *_gpu.cu kernels compiled for the CPU) -- once in float32 and once in float64, on
input (8, 12288, 9).  Same deterministic state_dict as pointnet2msg_ref.npz (module_weights.weights, seed 4242).

Stored (kept small: the inputs are NOT stored, the test regenerates them from pvn3d_amd.synth and checks their SHA-256):
  pc_sha256            of the (8, 12288, 9) float32 input bytes
  fps<l>               (8, npoint) int16 FPS indices of SA level l
  idx_sha/<name>       SHA-256 of every frame's ball-query (8 tensors / frame) and three_nn (4 / frame) int32 indices
  <lvl>_cols           12 seeded point columns per level (shared by the frames)
  <lvl>_vals[_f64]     (8, C, 12) float32: those columns, all channels, float32 run / float64 run
  <lvl>_chan_sum       (8, C) float64 sum over the points (float64 run)       } projections that cover every element
  <lvl>_pt_sum         (8, n) float32 sum over the channels (float64 run)     }
  <lvl>_chan_abs / _pt_abs   matching sums of |x| (float32): the scale of the projection tolerance
  <lvl>_scale          (8,) max |x| of the level's output per frame

Usage: python tests/golden/make_golden_batch.py        (about ten minutes on 8 cores)
"""
import hashlib
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden_modules import Capture, RefExt, load_reference_modules, load_weights  # noqa: E402
from module_weights import weights_sha  # noqa: E402
from oracle import ref as kref  # noqa: E402
from pvn3d_amd import synth  # noqa: E402

B = 8
FIRST_FRAME = 200
N_COLS = 12
LEVELS = ["sa0", "sa1", "sa2", "sa3", "fp3", "fp2", "fp1", "fp0"]


def batch_input():
    """(8, 12288, 9) float32: xyz ++ 6 features of synthetic frames 200..207 (two of them with the reference's
    'wrap' duplicate padding, linemod_dataset.py:264, which makes FPS ties common)."""
    pcs = []
    for b in range(B):
        f = synth.synth_frame(frame=FIRST_FRAME + b, n_pts=12288, n_obj=3072, wrap_pad=0.1 if b in (3, 6) else 0.0)
        pcs.append(np.concatenate([f["pcld"], f["feats"].T], 1).astype(np.float32))
    return np.ascontiguousarray(np.stack(pcs))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    assert kref.build(), "needs /root/reference"
    torch.set_num_threads(8)
    ext = RefExt()
    cap = Capture(ext)
    pm, pu, net_mod = load_reference_modules(cap)
    pc = batch_input()
    out = {"pc_sha256": np.array(sha(pc)), "first_frame": np.int64(FIRST_FRAME), "wrap_frames": np.array([3, 6])}
    model = net_mod.Pointnet2MSG(input_channels=6).eval()
    keys, shapes, w = load_weights(model, seed=4242)
    out["weights_sha256"] = np.array(weights_sha(keys, w))
    rs = np.random.RandomState(17)
    for dtype, tag in ((torch.float32, ""), (torch.float64, "_f64")):
        t0 = time.time()
        feats, hooks = {}, []
        for i, m in enumerate(model.SA_modules):
            hooks.append(m.register_forward_hook(lambda mod, a, r, i=i: feats.__setitem__("sa%d" % i, r[1])))
        for i, m in enumerate(model.FP_modules):
            hooks.append(m.register_forward_hook(lambda mod, a, r, i=i: feats.__setitem__("fp%d" % i, r)))
        cap.log.clear()
        mdl = model.double() if dtype == torch.float64 else model.float()
        with torch.no_grad():
            y = mdl(torch.from_numpy(pc).to(dtype))
        for h in hooks:
            h.remove()
        assert y.shape == (B, 128, 12288)
        if tag == "":
            lvl = {"furthest_point_sampling": 0, "ball_query": 0, "three_nn": 0}
            for name, r in cap.log:
                k = lvl[name]
                lvl[name] += 1
                if name == "furthest_point_sampling":
                    out["fps%d" % k] = r.numpy().astype(np.int16)
                elif name == "ball_query":
                    for b in range(B):
                        out["idx_sha/bq%d_%d/%d" % (k // 2, k % 2, b)] = np.array(sha(r[b].numpy().astype(np.int32)))
                else:           # FP modules run in reverse: call k = 0 is FP_modules[3]
                    for b in range(B):
                        out["idx_sha/nn%d/%d" % (3 - k, b)] = np.array(sha(r[1][b].numpy().astype(np.int32)))
            assert lvl == {"furthest_point_sampling": 4, "ball_query": 8, "three_nn": 4}
        for name in LEVELS:
            t = feats[name].detach()
            if tag == "":
                cols = np.sort(rs.choice(t.shape[2], size=min(N_COLS, t.shape[2]), replace=False))
                out["%s_cols" % name] = cols.astype(np.int32)
                out["%s_vals" % name] = t[:, :, cols].float().numpy()
            else:
                cols = out["%s_cols" % name]
                out["%s_vals_f64" % name] = t[:, :, cols].float().numpy()
                td = t.double()
                out["%s_chan_sum" % name] = td.sum(2).numpy()
                out["%s_chan_abs" % name] = td.abs().sum(2).float().numpy()
                out["%s_pt_sum" % name] = td.sum(1).float().numpy()
                out["%s_pt_abs" % name] = td.abs().sum(1).float().numpy()
                out["%s_scale" % name] = td.abs().amax(dim=(1, 2)).float().numpy()
        print("run%s done in %.0f s" % (tag or "_f32", time.time() - t0), flush=True)
    model.float()
    path = os.path.join(HERE, "pointnet2msg_batch_ref.npz")
    np.savez_compressed(path, **out)
    print("wrote %s, %d KiB, %d arrays" % (path, os.path.getsize(path) // 1024, len(out)))


if __name__ == "__main__":
    main()
