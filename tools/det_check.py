import sys, numpy as np, torch
sys.path.insert(0,'/root/repo')
from pvn3d_amd import synth
from pvn3d_amd.lib.utils import pvn3d_eval_utils as ev
dev=torch.device('cuda:0')
T=lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
fr=[synth.synth_frame(frame=10+i,n_pts=2048,n_obj=400+100*i) for i in range(4)]
st=lambda k: torch.stack([T(f[k]) for f in fr],0)
def batch():
    r=ev.cal_batch_poses_lm(st("pcld"),st("mask"),st("ctr_of"),st("pred_kp_of"),True,2,False,1)
    return r["poses"].cpu().numpy(), r["cls_kps"].cpu().numpy(), r["iters"].cpu().numpy()
def single(f):
    r=ev.cal_batch_poses_lm(T(f["pcld"])[None],T(f["mask"])[None],T(f["ctr_of"])[None],T(f["pred_kp_of"])[None],True,2,False,1)
    return r["poses"].cpu().numpy()[0], r["cls_kps"].cpu().numpy()[0], r["iters"].cpu().numpy()[0]
b1=batch(); b2=batch()
print("batch repeat equal:", np.array_equal(b1[1],b2[1]))
for i,f in enumerate(fr):
    s1=single(f); s2=single(f)
    print(i,"single repeat equal:",np.array_equal(s1[1],s2[1]),"single==batch kps:",np.array_equal(s1[1],b1[1][i]), "maxdiff",np.abs(s1[1]-b1[1][i]).max(), "iters",s1[2],b1[2][i])
