#!/usr/bin/env python3
"""SA levels 1-3 at 64 frames: time of the fused module call with the centres in FPS order (spatially scattered:
consecutive centres of a column block share no neighbours) against the same centres processed in a spatially coherent
order (sorted by the index of their first neighbour; geometry permuted before the call, outputs come out permuted) --
what a locality-sorted block order could buy the gather-bound layer 0 of these kernels."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pvn3d_amd.lib.pointnet2_utils import pointnet2_modules as pm, _ext  # noqa: E402
from pvn3d_amd import synth  # noqa: E402
from fp0_locality import med  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B = 64
    torch.manual_seed(0)
    cloud = torch.from_numpy(np.stack([synth.synth_frame(frame=i, n_pts=12288, n_obj=3072)["pcld"] for i in range(B)])).to(dev)
    with torch.no_grad():
        s0 = _ext.furthest_point_sampling(cloud, 2048).long()
        l1 = torch.gather(cloud, 1, s0[..., None].expand(-1, -1, 3)).contiguous()          # level-0 centres, FPS order
        for name, n_in, npoint, c_in, radii, nss, mlps in (
                ("SA1", 2048, 1024, 96, [0.025, 0.05], [16, 32], [[96, 64, 64, 128], [96, 64, 96, 128]]),
                ("SA2", 1024, 512, 256, [0.05, 0.1], [16, 32], [[256, 128, 196, 256], [256, 128, 196, 256]]),
                ("SA3", 512, 128, 512, [0.1, 0.2], [16, 32], [[512, 256, 256, 512], [512, 256, 384, 512]])):
            xyz = l1[:, :n_in].contiguous()
            sa = pm.PointnetSAModuleMSG(npoint=npoint, radii=radii, nsamples=nss, mlps=[list(m) for m in mlps]).to(dev).eval()
            sa._point_major_out = True
            feats = torch.randn(B, n_in, c_in, device=dev).transpose(1, 2)
            new_xyz, idxs = sa.sample_and_query(xyz)
            t_real = med(lambda: sa(xyz, feats, geometry=(new_xyz, idxs)))
            order = torch.argsort(idxs[-1][..., 0].long(), dim=1, stable=True)
            g = lambda t: torch.gather(t, 1, order[..., None].expand(-1, -1, t.size(-1))).contiguous()
            nx_s, idx_s = g(new_xyz), [g(i) for i in idxs]
            t_sort = med(lambda: sa(xyz, feats, geometry=(nx_s, idx_s)))
            print("%s (both scales, 64 frames): centres in FPS order %.3f ms   in the order of their first neighbour %.3f ms"
                  % (name, t_real, t_sort), flush=True)


if __name__ == "__main__":
    main()
