#!/usr/bin/env python3
"""Time the REFERENCE's own MeanShiftTorch.fit (pvn3d/lib/utils/meanshift_pytorch.py, imported file-level) beside
oracle/torch_port.meanshift_fit_dense -- the `kind: "port"` CPU baseline of bench.py -- on the same votes, same thread
count, so that the port has a measured relation to the file it restates.  Runs where /root/reference exists (the build
container, CPU only); writes profiles/<tag>_meanshift_ref_vs_port.json.  Usage: python tools/ms_ref_vs_port.py r04"""
import importlib.util
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
from oracle import torch_port  # noqa: E402
from pvn3d_amd import synth  # noqa: E402

REF = "/root/reference/pvn3d/lib/utils/meanshift_pytorch.py"


def load_ref():
    for n in ["cv2", "neupeak", "neupeak.utils", "neupeak.utils.webcv2"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["neupeak.utils.webcv2"].imshow = None
    sys.modules["neupeak.utils.webcv2"].waitKey = None
    spec = importlib.util.spec_from_file_location("ref_meanshift_pytorch", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    ref = load_ref()
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    f = synth.synth_frame(frame=0, n_pts=12288, n_obj=3072)
    sel = f["mask"] == 1
    fits = [("centre", (f["pcld"] - f["ctr_of"][0])[sel]), ("keypoint 0", (f["pcld"] - f["pred_kp_of"][0])[sel])]
    rows = []
    for name, A in fits:
        At = torch.from_numpy(np.ascontiguousarray(A))
        res = {}
        for kind, fn in (("reference", lambda: ref.MeanShiftTorch(bandwidth=0.08).fit(At)),
                         ("port", lambda: torch_port.meanshift_fit_dense(At, 0.08)[:2])):
            fn()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = fn()
                ts.append(time.perf_counter() - t0)
            res[kind] = (float(np.median(ts)), out)
        same_ctr = float((res["reference"][1][0] - res["port"][1][0]).abs().max())
        same_lab = bool(torch.equal(res["reference"][1][1].bool(), res["port"][1][1].bool()))
        rows.append(dict(fit=name, n=int(len(A)), seconds_reference=res["reference"][0], seconds_port=res["port"][0],
                         port_over_reference=res["port"][0] / res["reference"][0], max_abs_centre_diff=same_ctr,
                         labels_identical=same_lab))
        print(rows[-1])
    out = dict(tag=tag, host_threads=threads, host="build container (CPU only)",
               what="median of 3 runs of one fit each: the reference's MeanShiftTorch.fit (file-level import of "
                    "pvn3d/lib/utils/meanshift_pytorch.py:18-51, .repeat() tensors materialised) vs "
                    "oracle/torch_port.meanshift_fit_dense (broadcast instead of .repeat()), same votes, same threads",
               fits=rows,
               mean_port_over_reference=float(np.mean([r["port_over_reference"] for r in rows])))
    with open(os.path.join(ROOT, "profiles", "%s_meanshift_ref_vs_port.json" % tag), "w") as fh:
        json.dump(out, fh, indent=1)
    print("port / reference time:", out["mean_port_over_reference"])


if __name__ == "__main__":
    main()
