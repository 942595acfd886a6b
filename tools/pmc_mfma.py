#!/usr/bin/env python3
"""MFMA-busy of every fused-MLP launch of one Pointnet2MSG forward (64 frames), from rocprofv3 PMC counters.

One counter pass (kernel trace + SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES,
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY) of `tools/bench_ops.py --ops msg`, then per launch:
  duration_us            End - Start of the dispatch (kernel trace of the same pass)
  effective_clock_ghz    GRBM_GUI_ACTIVE / 8 / duration    (the counter is summed over the 8 XCDs; DVFS:
                         MI355X_MICROARCH.md, "DVFS give-back")
  mfma_busy              SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)   -- share of SIMD-cycles with
                         the matrix pipe executing, at the clock the kernel actually ran at
  mfma_busy_vs_sq_busy   SQ_VALU_MFMA_BUSY_CYCLES / (32 x SQ_BUSY_CYCLES)  (SQ_BUSY is summed over the 32 shader
                         engines, 32 SIMDs each) -- the same ratio from SQ's own busy window
  tflops                 algorithmic flops of the launch / duration; at_clock = 64 flop/clk/SIMD x 1024 x clock
Writes profiles/<tag>_mfma_busy.json.  Usage on the GPU box: python tools/pmc_mfma.py r02
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
CTRS = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY",
        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
# (K0, M...) chains of Pointnet2MSG in launch order and the columns they run on per frame (lib/pvn3d.py:67-118)
CHAINS = [("SA0 ns16", [9, 16, 16, 32], 2048 * 16), ("SA0 ns32", [9, 32, 32, 64], 2048 * 32),
          ("SA1 ns16", [99, 64, 64, 128], 1024 * 16), ("SA1 ns32", [99, 64, 96, 128], 1024 * 32),
          ("SA2 ns16", [259, 128, 196, 256], 512 * 16), ("SA2 ns32", [259, 128, 196, 256], 512 * 32),
          ("SA3 ns16", [515, 256, 256, 512], 128 * 16), ("SA3 ns32", [515, 256, 384, 512], 128 * 32),
          ("FP3", [1536, 512, 512], 512), ("FP2", [768, 512, 512], 1024), ("FP1", [608, 256, 256], 2048),
          ("FP0", [262, 128, 128], 12288)]
FRAMES = 64
N_XCD = 8          # GRBM_GUI_ACTIVE comes back summed over the XCDs


def is_chain(name):
    """a fused-chain kernel: the fp32 / split 4 + 4 wave families (mlp_chain_*) or the narrow-chain kernel of SA levels 0-1"""
    return "mlp_chain" in name or "chain_narrow" in name


def group_launches(launches):
    """launches: [(dispatch id, kernel name)] of the whole run, in order -> one list of dispatch ids per entry of CHAINS
    for the LAST fused forward.  A forward starts at SA level 0's first kernel.  A chain is one mlp_chain* launch, except:
    FP levels 3 / 2 run layer by layer on the split GEMM (three consecutive sg_gemm launches, csrc/split_gemm.hip), and
    a single sg_gemm in front of a chain kernel is that level's pre-contraction (_ext.sa_precontract, the FP level-0
    form in _ext.fp_interp_mlp) -- booked on the chain that follows it."""
    first = next(k for _, k in launches if is_chain(k))
    start = max(i for i, (_, k) in enumerate(launches) if k == first)
    groups, pending = [], []
    for d, k in launches[start:]:
        nxt = CHAINS[len(groups)][0] if len(groups) < len(CHAINS) else None
        if nxt is None:
            break
        if "sg_gemm" in k:
            pending.append(d)
            if len(pending) == 3 and nxt in ("FP3", "FP2"):
                groups.append(pending); pending = []
        else:
            groups.append(pending + [d]); pending = []
    assert len(groups) == len(CHAINS), "launch pattern of the forward not recognised: %d groups" % len(groups)
    return groups


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    out = "/tmp/pmc_mfma"
    subprocess.run(["rm", "-rf", out])
    env = dict(os.environ, TMPDIR="/tmp", PVN3D_GEOMETRY_STREAM="0")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + CTRS + ["-d", out, "-o", "m", "--output-format", "csv", "--",
                    sys.executable, os.path.join(ROOT, "tools", "bench_ops.py"), "--ops", "msg", "--reps", "1"],
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=900)
    ctr = collections.OrderedDict()
    for r in csv.DictReader(open(glob.glob(out + "/**/*counter_collection.csv", recursive=True)[0])):
        if is_chain(r["Kernel_Name"]) or "sg_gemm" in r["Kernel_Name"]:
            ctr.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    dur = {}
    for r in csv.DictReader(open(glob.glob(out + "/**/*kernel_trace.csv", recursive=True)[0])):
        if is_chain(r["Kernel_Name"]) or "sg_gemm" in r["Kernel_Name"]:
            dur[int(r["Dispatch_Id"])] = ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                                          r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0])
    groups = group_launches([(d, dur[d][1]) for d in sorted(ctr)])
    rows, tot_fl, tot_us, tot_busy, tot_act = [], 0.0, 0.0, 0.0, 0.0
    for (name, dims, cols), g in zip(CHAINS, groups):
        c = {k: sum(ctr[d][k] for d in g) for k in ctr[g[0]]}
        us = sum(dur[d][0] for d in g)
        kern = dur[g[-1]][1] + (" x%d" % len(g) if len(g) == 3 else " + sg_gemm (feature half of layer 0)" if len(g) == 2 else "")
        fl = 2.0 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1)) * cols * FRAMES
        gui = c["GRBM_GUI_ACTIVE"] / N_XCD
        clk = gui / (us * 1e3)
        # split-bf16 chains (csrc/sa_mlp_split.hip) execute six bf16 partial products per fp32 multiply on
        # v_mfma_f32_32x32x16_bf16 (1024 flop/clk/SIMD); the fp32 chains v_mfma_f32_32x32x2_f32 (64 flop/clk/SIMD).
        # `tflops` stays the algorithmic (fp32-equivalent) rate; `tflops_peak_at_clock` is what the pipe the launch runs
        # on could deliver of THAT quantity at the measured clock (bf16 peak / 6 for the split chains).
        # fp16 x 2 chains (round 5; template argument AR = 1 of mlp_chain_s3_kernel / sg_gemm_kernel): three fp16 partial
        # products per multiply on v_mfma_f32_32x32x16_f16 -> bf16/fp16 peak / 3.
        split = "s3_kernel" in kern or "sg_gemm" in kern or "chain_narrow" in kern
        last = dur[g[-1]][1]
        fp16 = split and (last.rstrip().endswith(", 1>") or "sg_gemm_kernel<1>" in last or "chain_narrow" in last)
        per_clk = 1024.0 / 3.0 if fp16 else 1024.0 / 6.0 if split else 64.0
        rows.append(dict(chain=name, kernel=kern, arithmetic="fp16x2 split (3 fp16 MFMA products per fp32 multiply)" if fp16
                         else "bf16x3 split (6 bf16 MFMA products per fp32 multiply)" if split
                         else "fp32 MFMA", duration_us=us, effective_clock_ghz=clk,
                         mfma_busy=c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui),
                         mfma_busy_vs_sq_busy=c["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * c["SQ_BUSY_CYCLES"]),
                         tflops=fl / (us * 1e-6) / 1e12, tflops_peak_at_clock=per_clk * 1024 * clk * 1e9 / 1e12,
                         wave_cycles_share=dict(wait_any=c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
                                                wait_inst_any=c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                                active_inst_any=c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]),
                         counters=c))
        tot_fl += fl; tot_us += us; tot_busy += c["SQ_VALU_MFMA_BUSY_CYCLES"]; tot_act += gui
    res = dict(tag=tag, command="rocprofv3 --kernel-trace --pmc %s -- python tools/bench_ops.py --ops msg --reps 1" % " ".join(CTRS),
               frames=FRAMES, launches=rows,
               total=dict(duration_us=tot_us, tflops=tot_fl / (tot_us * 1e-6) / 1e12, mfma_busy=tot_busy / (1024.0 * tot_act),
                          effective_clock_ghz=tot_act / (tot_us * 1e3)))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "%s_mfma_busy.json" % tag), "w") as f:
        json.dump(res, f, indent=1)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "%s_mfma_busy.json" % tag), "w") as f:
        json.dump(res, f, indent=1)
    for r in rows:
        print("%-9s %-42s %8.1f us  clk %.2f GHz  MFMA-busy %.3f (%.3f)  %6.1f TF/s of %5.1f at clock" %
              (r["chain"], r["kernel"][:42], r["duration_us"], r["effective_clock_ghz"], r["mfma_busy"],
               r["mfma_busy_vs_sq_busy"], r["tflops"], r["tflops_peak_at_clock"]))
    print("total: %.1f us, %.1f TFLOP/s, MFMA-busy %.3f, clock %.2f GHz" %
          (tot_us, res["total"]["tflops"], res["total"]["mfma_busy"], res["total"]["effective_clock_ghz"]))


if __name__ == "__main__":
    main()
