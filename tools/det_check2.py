import sys, numpy as np, torch, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd import synth
from pvn3d_amd._lib import lib, check
from pvn3d_amd.lib.utils import _vote_engine as eng
dev=torch.device('cuda:0')
T=lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
fr=[synth.synth_frame(frame=10+i,n_pts=2048,n_obj=400+100*i) for i in range(4)]
st=lambda k: torch.stack([T(f[k]) for f in fr],0)
pcld,mask,ctr_of,kp_of=eng._prep(st("pcld"),st("mask"),st("ctr_of"),st("pred_kp_of"))
F,N,K=4,2048,8
inst_frame=torch.arange(F,dtype=torch.int32,device=dev); inst_cls=torch.ones(F,dtype=torch.int32,device=dev)
votes,so,sc=eng.vote_compact(pcld,mask,ctr_of,kp_of,inst_frame,inst_cls,0,K+1)
n_seg=F*(K+1); total=votes.size(0)
def run(max_iter, flags):
    ws_bytes=int(lib.pvn3d_meanshift_workspace_bytes(n_seg,total,max_iter))
    ws=torch.zeros(ws_bytes,dtype=torch.uint8,device=dev)
    ctr=torch.empty(n_seg,3,device=dev); lab=torch.empty(total,dtype=torch.uint8,device=dev); it=torch.empty(n_seg,dtype=torch.int32,device=dev)
    check(lib.pvn3d_meanshift_fit_batch(votes.data_ptr(),so.data_ptr(),sc.data_ptr(),n_seg,total,N,0.08,max_iter,ctr.data_ptr(),lab.data_ptr(),it.data_ptr(),ws.data_ptr(),ws_bytes,None,0,flags,torch.cuda.current_stream().cuda_stream),"ms")
    torch.cuda.synchronize()
    sz=(16*total+255)//256*256
    c0=ws[:16*total].view(torch.float32).view(total,4).clone(); c1=ws[sz:sz+16*total].view(torch.float32).view(total,4).clone()
    return ctr.cpu().numpy(), it.cpu().numpy(), c0.cpu().numpy(), c1.cpu().numpy()
cnt=sc.cpu().numpy(); off=so.cpu().numpy()
for mi in (0,1):
    a=run(mi,1); b=run(mi,1)
    buf=3 if (mi+1)%2==1 else 2   # iterations run = mi+1 -> cbuf[(mi+1)&1]
    A=a[buf]; B=b[buf]
    print("max_iter",mi,"iters",a[1][:12])
    bad=0
    for s in range(n_seg):
        x=A[off[s]:off[s]+cnt[s],:3]; y=B[off[s]:off[s]+cnt[s],:3]
        d=np.abs(x-y).max(1)
        idx=np.nonzero(~(d==0))[0]
        if len(idx):
            bad+=1
            print("  seg",s,"n",cnt[s],"diff rows",len(idx),"first",idx[:12],"max",np.nanmax(d))
    print("  segs with run-to-run diffs:",bad)
