import copy, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm, _train_mlp
from pvn3d_amd import synth
dev = torch.device("cuda:0")
def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))
torch.manual_seed(3)
B, N = 2, 1024
xyz = torch.from_numpy(np.stack([synth.synth_cloud(np.random.default_rng(i), N)[0] for i in range(B)], 0)).to(dev)
base = pm.PointnetSAModuleMSG(npoint=128, radii=[0.05, 0.1], nsamples=[16, 32], mlps=[[10, 16, 32], [10, 32, 24, 64]]).to(dev).train()
for p in base.parameters():
    if p.dim() == 1: p.data.uniform_(0.5, 1.5)
feats_pm = torch.randn(B, N, 10, device=dev)
gout = torch.randn(B, 96, 128, device=dev)
res = {}
for mode in ("fp32", "autocast", "fused"):
    mod = copy.deepcopy(base)
    f = feats_pm.clone().transpose(1, 2).requires_grad_(True)
    _train_mlp.TRAIN_FUSED = mode == "fused"
    if mode == "autocast":
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            new_xyz, out = mod(xyz, f)
    else:
        new_xyz, out = mod(xyz, f)
    (out.float() * gout).sum().backward()
    res[mode] = dict(out=out.detach().float(), df=f.grad.detach(), params={k: v.grad.detach() for k, v in mod.named_parameters()},
                     bufs={k: v.detach().clone() for k, v in mod.named_buffers()})
for m in ("autocast", "fused"):
    a, b = res[m], res["fp32"]
    print(m, "out %.4f df %.4f" % (rel(a["out"], b["out"]), rel(a["df"], b["df"])))
    for k in b["params"]: print("   ", k, "%.4f" % rel(a["params"][k], b["params"][k]))
    for k in b["bufs"]:
        if "num" not in k: print("   buf", k, "%.5f" % rel(a["bufs"][k], b["bufs"][k]))
print("---- FP")
torch.manual_seed(4)
B, n, mk = 2, 1024, 256
unknown = torch.from_numpy(np.stack([synth.synth_cloud(np.random.default_rng(i), n)[0] for i in range(B)], 0)).to(dev)
known = unknown[:, :mk].contiguous()
base = pm.PointnetFPModule(mlp=[40 + 7, 64, 48]).to(dev).train()
for p in base.parameters():
    if p.dim() == 1: p.data.uniform_(0.5, 1.5)
uf0, kf0 = torch.randn(B, 7, n, device=dev), torch.randn(B, mk, 40, device=dev)
gout = torch.randn(B, 48, n, device=dev)
res = {}
for mode in ("fp32", "autocast", "fused"):
    mod = copy.deepcopy(base)
    uf = uf0.clone().requires_grad_(True)
    kf = kf0.clone().transpose(1, 2).requires_grad_(True)
    _train_mlp.TRAIN_FUSED = mode == "fused"
    if mode == "autocast":
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            out = mod(unknown, known, uf, kf)
    else:
        out = mod(unknown, known, uf, kf)
    (out.float() * gout).sum().backward()
    res[mode] = dict(out=out.detach().float(), du=uf.grad.detach(), dk=kf.grad.detach(), params={k: v.grad.detach() for k, v in mod.named_parameters()})
for m in ("autocast", "fused"):
    a, b = res[m], res["fp32"]
    print(m, "out %.4f du %.4f dk %.4f" % (rel(a["out"], b["out"]), rel(a["du"], b["du"]), rel(a["dk"], b["dk"])))
    for k in b["params"]: print("   ", k, "%.4f" % rel(a["params"][k], b["params"][k]))
