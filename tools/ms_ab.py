import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth
from pvn3d_amd.lib.utils import _vote_engine as eng
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
nfit, n = 144, 3072
pts4 = np.zeros((nfit * n, 4), np.float32)
for f in range(nfit):
    a = rng.normal(size=(n, 3)) * 0.005 + np.array([0.1, -0.05, 0.9])
    k = n // 10
    a[rng.permutation(n)[:k]] += rng.normal(size=(k, 3)) * 0.3
    pts4[f * n:(f + 1) * n, :3] = a
P = torch.from_numpy(pts4).to(dev)
so = torch.arange(nfit, dtype=torch.int32, device=dev) * n
sc = torch.full((nfit,), n, dtype=torch.int32, device=dev)
for kern in ("packed+split+noearly", "packed+split"):
    for _ in range(2):
        c, l, it = eng.meanshift_fit_batch(P, so, sc, n, 0.08, 300, kernel=kern, poll_every=0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); c, l, it = eng.meanshift_fit_batch(P, so, sc, n, 0.08, 300, kernel=kern, poll_every=0); e1.record(); e1.synchronize()
    print(kern, "%.2f ms" % e0.elapsed_time(e1), "iters", int(it.min()), int(it.max()), flush=True)
import ctypes
from pvn3d_amd._lib import lib
if hasattr(lib, "pvn3d_ms_probe_read"):
    buf = (ctypes.c_int * 1024)()
    lib.pvn3d_ms_probe_read(buf, 1)
    c, l, it = eng.meanshift_fit_batch(P, so, sc, n, 0.08, 300, kernel="packed+split", poll_every=0)
    torch.cuda.synchronize()
    lib.pvn3d_ms_probe_read(buf, 0)
    a = np.array(buf[:])
    for t in (1, 5, 6, 10, 14, 20, 40, 80, 120, 160, 200, 240):
        print("t=%3d iterated %7d fixed %7d active WGs %5d" % (t, a[512 + t], a[t], a[768 + t]))
