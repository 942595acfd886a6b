#!/usr/bin/env python3
"""One vote -> cluster -> pose call of the headline batch (64 frames x 9 fits of 3072 votes) after warm-up, for a kernel trace:
rocprofv3 --kernel-trace --output-format csv -d out -- python tools/ms_trace_one.py [kernel-spec]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import make_inputs, run_postproc, StageTimer
from pvn3d_amd.lib.utils import _vote_engine as eng
dev = torch.device("cuda:0")
inp = make_inputs(64, 12288, 3072, dev, 0)
off = StageTimer(False)
eng.DEFAULT_KERNEL = sys.argv[1] if len(sys.argv) > 1 else "sgpr"
for _ in range(3):
    res = run_postproc(inp, off, 4)
torch.cuda.synchronize()
print("iters", res["iters"].min().item(), res["iters"].max().item(), "counts", res["counts"].min().item(), res["counts"].max().item())
