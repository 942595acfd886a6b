#!/usr/bin/env python3
"""HBM traffic per bench step from rocprofv3 PMC counters (MI355X_MICROARCH.md, HBM section):
separate passes (FETCH_SIZE, WRITE_SIZE; KB units; each once for the default fused path and
once for `--ops-only`, the unfused op chain the HBM rooflines are quoted on), FETCH_SIZE doubled (gfx950 reports half
of a wide coalesced read -- calibrated here on a 1 GiB hipMemcpy D2D: WRITE_SIZE = 1048576 KB
exactly, FETCH_SIZE = 524300 KB).  Runs `bench.py --serial` under the profiler and writes
profiles/<tag>_pmc_traffic.json (+ profiles/latest_pmc_traffic.json, read by bench.py to fill
roofline.traffic).  Usage on the GPU box:  python tools/pmc_traffic.py r01"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
STAGE_OF = [("[sa_mlp]", "sa_mlp"), ("[fp_mlp]", "fp_mlp"),      # split-GEMM launches, tagged by run_pass
            ("sa_chain_narrow_kernel", "sa_mlp"), ("fp_chain_narrow_kernel", "fp_mlp"),
            ("mlp_chain_s3_kernel<true", "sa_mlp"), ("mlp_chain_s3_kernel<false", "fp_mlp"),
            ("mlp_chain_kernel<true", "sa_mlp"), ("mlp_chain_wide_kernel<true", "sa_mlp"),
            ("mlp_chain_mid_kernel<true", "sa_mlp"), ("mlp_chain_cols_kernel<true", "sa_mlp"),
            ("mlp_chain_kernel<false", "fp_mlp"), ("mlp_chain_wide_kernel<false", "fp_mlp"),
            ("mlp_chain_mid_kernel<false", "fp_mlp"), ("mlp_chain_cols_kernel<false", "fp_mlp"),
            ("group_points", "group"), ("group_xyz_rel", "group"), ("three_interpolate", "three_interpolate"),
            ("ball_query", "ball_query"), ("grid_build", "ball_query"), ("three_nn", "three_nn"),
            ("fps_", "fps"), ("gather_points", "gather"), ("ms_", "vote_cluster_pose"),
            ("vote_compact", "vote_cluster_pose"), ("best_fit", "vote_cluster_pose")]
WARMUP, STEPS = 3, 2


OP_STAGES = ("group", "three_interpolate", "ball_query", "three_nn", "fps", "gather")


def run_pass(counter, outdir, extra=()):
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", outdir, "-o", "p", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--serial", "--steps", str(STEPS), "--warmup", str(WARMUP),
           "--no-cpu-baseline", "--no-stage-events", "--no-extra-configs"] + list(extra)
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=600)
    per_kernel = collections.defaultdict(float)
    with open(os.path.join(outdir, "p_counter_collection.csv")) as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    # The split GEMM (csrc/split_gemm.hip: sg_gemm_kernel, sg_split_rows_kernel) serves both MLP stages: a launch belongs
    # to the stage of the next fused-chain kernel after it -- an SA level's pre-contraction precedes that level's chains,
    # FP levels 3 / 2 (GEMM only) precede FP level 1's chain, FP level 0's pre-contraction precedes its chain.
    nxt, stage_after = None, [None] * len(rows)
    for i in range(len(rows) - 1, -1, -1):
        k = rows[i]["Kernel_Name"]
        if "mlp_chain" in k or "chain_narrow" in k:
            nxt = "sa_mlp" if ("<true" in k or "sa_chain_narrow" in k) else "fp_mlp"
        stage_after[i] = nxt
    for r, st in zip(rows, stage_after):
        name = r["Kernel_Name"]
        if ("sg_gemm_kernel" in name or "sg_split_rows_kernel" in name or "absmax_kernel" in name
                or "sg_bound_affine_kernel" in name):
            name = "%s [%s]" % (name.split("(")[0] if not name.startswith("(") else name[:60], st or "fp_mlp")
        per_kernel[name] += float(r["Counter_Value"]) * 1024.0
    return per_kernel


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    out = os.path.join(ROOT, "gpurun_out", "pmc_%s" % tag)
    n_steps = WARMUP + STEPS
    stages = collections.defaultdict(lambda: dict(read_bytes=0.0, write_bytes=0.0))
    kernels = {}
    for mode, extra in (("fused", ()), ("ops_only", ("--ops-only",))):
        fetch = run_pass("FETCH_SIZE", "%s_%s_fetch" % (out, mode), extra)
        write = run_pass("WRITE_SIZE", "%s_%s_write" % (out, mode), extra)
        for name in set(fetch) | set(write):
            rd = 2.0 * fetch.get(name, 0.0) / n_steps      # gfx950 FETCH_SIZE correction
            wr = write.get(name, 0.0) / n_steps
            for key, st in STAGE_OF:
                if key in name:
                    # data-movement ops are quoted on the unfused chain, everything else on the fused path
                    if (st in OP_STAGES) == (mode == "ops_only"):
                        kernels["%s:%s" % (mode, name[:80])] = dict(read_bytes_per_step=rd, write_bytes_per_step=wr)
                        stages[st]["read_bytes"] += rd
                        stages[st]["write_bytes"] += wr
                    break
    res = dict(tag=tag, command="rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python bench.py --serial "
               "--steps %d --warmup %d --no-cpu-baseline --no-stage-events --no-extra-configs [--ops-only]" % (STEPS, WARMUP),
               frames_per_step=64, correction="FETCH_SIZE x2 (gfx950), KB->bytes x1024, totals / %d steps" % n_steps,
               stage_bytes_per_step={k: dict(v, total_bytes=v["read_bytes"] + v["write_bytes"]) for k, v in stages.items()},
               kernels=kernels)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    for fn in ("%s_pmc_traffic.json" % tag, "latest_pmc_traffic.json"):
        with open(os.path.join(ROOT, "profiles", fn), "w") as f:
            json.dump(res, f, indent=1, sort_keys=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_pmc_traffic.json" % tag), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(json.dumps(res["stage_bytes_per_step"], indent=1))


if __name__ == "__main__":
    main()
