import torch, time, sys
sys.path.insert(0,'/root/repo')
from pvn3d_amd._lib import lib
dev=torch.device('cuda:0'); st=torch.cuda.current_stream().cuda_stream
for rows,c,ld in ((65536,512,512),(131072,96,96),(786432,6,9),(65536,256,512),(1000,7,7),(33,130,132)):
    x=torch.randn(rows,ld,device=dev); x[rows//2, c-1]=-77.5
    o=torch.zeros(1,device=dev)
    assert lib.pvn3d_absmax(rows,c,x.data_ptr(),ld,o.data_ptr(),st)==0
    assert float(o)==float(x[:,:c].abs().max()), (rows,c,ld,float(o))
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): lib.pvn3d_absmax(rows,c,x.data_ptr(),ld,o.data_ptr(),st)
    e1.record(); torch.cuda.synchronize()
    print(rows,c,ld,"%.1f us"%(e0.elapsed_time(e1)*100), "%.2f TB/s"%(rows*c*4/(e0.elapsed_time(e1)/10*1e-3)/1e12))
