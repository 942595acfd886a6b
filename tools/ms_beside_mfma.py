#!/usr/bin/env python
"""Do kernels keep their bits when kernels of the other island run on the same SIMDs?

tools/sg_fault_repro.hip pinned the round-4 split-GEMM epilogue fault to a packed-fp32 VALU instruction whose low half
selects the HIGH register of a source pair (`v_pk_mul_f32 ... op_sel:[0,1]`): it returned +0 in lanes 48-63 now and then
-- but only while the second wave of its SIMD was inside an MFMA / LDS K loop.  The MeanShift iteration kernel is built
from packed-fp32 instructions with such selectors (csrc/meanshift.hip), and in bench.py's pipelined step it runs beside
the MFMA kernels of the Pointnet2MSG forward.  This script runs
  (a) a MeanShift batch alone and beside a train of split-GEMM / fused-chain launches on a second stream,
  (b) the split GEMM (gathered-add epilogue) alone and beside MeanShift iterations,
and compares bits.  Every kernel of the library is deterministic, so any difference is a fault.
Usage (GPU box): python tools/ms_beside_mfma.py [trials]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pvn3d_amd._lib import lib, check  # noqa: E402
from pvn3d_amd.lib.pointnet2_utils import _fused_mlp as fm  # noqa: E402
from pvn3d_amd.lib.utils import _vote_engine as eng  # noqa: E402

dev = torch.device("cuda:0")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(5)

# ---- MeanShift batch: 144 fits of 3072 votes, 10 % outliers (tens of iterations under the reference's stop rule)
n, fits = 3072, 144
pts4 = np.zeros((fits * n, 4), np.float32)
for f in range(fits):
    a = rng.normal(size=(n, 3)) * 0.005 + np.array([0.1, -0.05, 0.9]) + rng.normal(size=3) * 0.02
    k = n // 10
    a[rng.permutation(n)[:k]] += rng.normal(size=(k, 3)) * 0.05
    pts4[f * n:(f + 1) * n, :3] = a
P = torch.from_numpy(pts4).to(dev)
so = torch.arange(fits, dtype=torch.int32, device=dev) * n
sc = torch.full((fits,), n, dtype=torch.int32, device=dev)


def ms(kernel):
    c, l, it = eng.meanshift_fit_batch(P, so, sc, n, 0.08, 300, kernel=kernel, aligned32=True)
    return c.clone(), l.clone(), it.clone()


# ---- split GEMM with the gathered-add epilogue at the FP-level-2 size of the 64-frame bench
Pn, K, N, B, zn, zm = 65536, 256, 512, 64, 1024, 512
torch.manual_seed(0)
X = torch.randn(Pn, K, device=dev)
W = torch.randn(N, K, device=dev) / K ** 0.5
S = fm._slabs(K)
xs = torch.empty(Pn * S * 96, dtype=torch.uint8, device=dev)
st0 = torch.cuda.current_stream().cuda_stream
check(lib.pvn3d_split_rows(Pn, K, X.data_ptr(), K, xs.data_ptr(), S, st0), "split_rows")
ws = fm._pack_weight_s16(W, S)
Np = ws.size(0)
bp = torch.randn(Np, device=dev)
Z = torch.randn(B * zm, Np, device=dev)
idx = torch.randint(0, zm, (Pn, 3), device=dev, dtype=torch.int32)
wg = torch.rand(Pn, 3, device=dev) * 0.8 + 0.1
Sout = fm._slabs(N)


def gemm(stream, out_s):
    check(lib.pvn3d_split_gemm(Pn, N, S, xs.data_ptr(), ws.data_ptr(), bp.data_ptr(), 1, Z.data_ptr(), Np, zn, zm,
                               idx.data_ptr(), wg.data_ptr(), None, 0, out_s.data_ptr(), Sout, stream.cuda_stream), "gemm")


side = torch.cuda.Stream()
main = torch.cuda.current_stream()
out_ref = torch.empty(Pn * Sout * 96, dtype=torch.uint8, device=dev)
gemm(main, out_ref)
torch.cuda.synchronize()
outs = [torch.empty_like(out_ref) for _ in range(2)]

report = {}
for kernel in ("packed+split+nowin", "sgpr+nowin", "sgpr", "packed+whole+nowin", "scalar+split+nowin"):
    base = ms(kernel)
    torch.cuda.synchronize()
    again = ms(kernel)
    torch.cuda.synchronize()
    solo_ok = all(torch.equal(a, b) for a, b in zip(base, again))
    bad_fits, bad_gemm = 0, 0
    for t in range(trials):
        side.wait_stream(main)
        with torch.cuda.stream(side):
            for r in range(300):
                gemm(side, outs[r & 1])
        got = ms(kernel)
        with torch.cuda.stream(side):
            g_bad = int((outs[0] != out_ref).sum().item()) + int((outs[1] != out_ref).sum().item())
        torch.cuda.synchronize()
        bad_gemm += g_bad
        diff = (got[0] != base[0]).any(dim=1) | (got[2] != base[2])
        bad_fits += int(diff.sum().item())
        if diff.any():
            i = int(diff.nonzero()[0])
            print("  %s trial %d: fit %d centre %s vs %s, iterations %d vs %d" % (
                kernel, t, i, got[0][i].tolist(), base[0][i].tolist(), int(got[2][i]), int(base[2][i])))
    report[kernel] = (solo_ok, bad_fits, bad_gemm)
    print("MeanShift %-22s: solo repeat identical %s; beside the split GEMM: %d of %d fits differ in %d trials "
          "(iterations %d-%d); GEMM bytes differing from its solo run: %d" % (
              kernel, solo_ok, bad_fits, fits * trials, trials, int(base[2].min()), int(base[2].max()), bad_gemm))

ok = all(v[0] and v[1] == 0 and v[2] == 0 for v in report.values())
print("RESULT:", "bits unchanged beside the other island" if ok else "BITS CHANGE under co-execution")
sys.exit(0 if ok else 1)
