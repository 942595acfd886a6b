#!/usr/bin/env python3
"""HBM traffic of ONE training step (BASELINE config 5 on one GPU) from rocprofv3 PMC counters, collected and
corrected as tools/pmc_traffic.py does for the inference bench (MI355X_MICROARCH.md, HBM section: separate passes
for FETCH_SIZE and WRITE_SIZE, KB units, FETCH_SIZE x2 on gfx950): `tools/train_bench.py --frames 24 --steps 2`
(bf16, no prefetch) under the profiler, per-kernel bytes summed into the groups of the step.
Writes profiles/<tag>_train_pmc_traffic.json (+ latest_train_pmc_traffic.json, read by bench.py's train_step
config).  Usage on the GPU box:  python tools/pmc_train.py r04"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
WARM, STEPS, FRAMES = 2, 2, 24           # train_bench.py runs 2 warm-up steps + STEPS timed ones
GROUPS = [("mt_gemm_nt_splitk", "wgrad"), ("mt_wgrad", "wgrad"), ("mt_transpose", "wgrad"),
          ("mt_gemm_nt", "gemm_fwd_dgrad"), ("mt_pack_weight", "gemm_fwd_dgrad"),
          ("mt_bn_", "batchnorm_relu_pool"),
          ("mt_inv_gather", "layer0_gather_backward"), ("mt_csr", "layer0_gather_backward"),
          ("mt_unpack", "layout_fp32_bf16"), ("mt_pack", "layout_fp32_bf16"),
          ("mt_gather", "layer0_gather"),
          ("fps_", "geometry"), ("ball_query", "geometry"), ("grid_build", "geometry"), ("three_nn", "geometry"),
          ("nn_grid", "geometry"), ("gather_points", "geometry"),
          ("of_l1", "vote_loss"), ("vote_loss", "vote_loss")]
OWNED = ("wgrad", "gemm_fwd_dgrad", "batchnorm_relu_pool", "layer0_gather_backward", "layout_fp32_bf16", "layer0_gather")


def run_pass(counter, outdir):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", outdir, "-o", "p", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "tools", "train_bench.py"), "--frames", str(FRAMES), "--steps", str(STEPS)]
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900)
    per_kernel = collections.defaultdict(float)
    with open(os.path.join(outdir, "p_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            per_kernel[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0
    return per_kernel


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    out = os.path.join(ROOT, "gpurun_out", "pmc_train_%s" % tag)
    n = WARM + STEPS
    fetch = run_pass("FETCH_SIZE", out + "_fetch")
    write = run_pass("WRITE_SIZE", out + "_write")
    groups = collections.defaultdict(lambda: dict(read_bytes=0.0, write_bytes=0.0))
    for name in set(fetch) | set(write):
        rd, wr = 2.0 * fetch.get(name, 0.0) / n, write.get(name, 0.0) / n
        grp = "heads_loss_optimizer_glue"
        for key, g in GROUPS:
            if key in name:
                grp = g
                break
        groups[grp]["read_bytes"] += rd
        groups[grp]["write_bytes"] += wr
    for g in groups.values():
        g["total_bytes"] = g["read_bytes"] + g["write_bytes"]
    owned = sum(groups[g]["total_bytes"] for g in OWNED if g in groups)
    res = dict(tag=tag, frames_per_step=FRAMES,
               command="rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python tools/train_bench.py --frames %d "
                       "--steps %d" % (FRAMES, STEPS),
               correction="FETCH_SIZE x2 (gfx950), KB->bytes x1024, totals / %d steps (2 warm-up + %d)" % (n, STEPS),
               group_bytes_per_step=dict(groups), sa_fp_mlp_chain_bytes_per_step=owned,
               all_kernels_bytes_per_step=sum(g["total_bytes"] for g in groups.values()))
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for fn in ("%s_train_pmc_traffic.json" % tag, "latest_train_pmc_traffic.json"):
        with open(os.path.join(ROOT, "profiles", fn), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_train_pmc_traffic.json" % tag), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in res.items() if k != "command"}, indent=1))


if __name__ == "__main__":
    main()
