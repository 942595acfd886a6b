"""pvn3d_amd -- MI355X (gfx950) native implementation of PVN3D's per-point voting hot path.

Scope: the PointNet++ set-abstraction operators (ball_query, furthest
point sampling, gather/group_points, three_nn, three_interpolate) and the downstream
keypoint-vote -> MeanShift clustering -> least-squares pose fit, behind the reference's own
Python module / operator API:

    pvn3d_amd.lib.pointnet2_utils._ext                 <- pvn3d/_ext-src (pybind module `_ext`)
    pvn3d_amd.lib.pointnet2_utils.pointnet2_utils      <- pvn3d/lib/pointnet2_utils/pointnet2_utils.py
    pvn3d_amd.lib.pointnet2_utils.pointnet2_modules    <- pvn3d/lib/pointnet2_utils/pointnet2_modules.py
    pvn3d_amd.lib.utils.meanshift_pytorch              <- pvn3d/lib/utils/meanshift_pytorch.py
    pvn3d_amd.lib.utils.pvn3d_eval_utils               <- pvn3d/lib/utils/pvn3d_eval_utils.py
    pvn3d_amd.lib.utils.basic_utils                    <- pvn3d/lib/utils/basic_utils.py (best_fit_transform, get_kps/get_ctr)

All device work goes through the C ABI of ``libpvn3d_hip.so`` (include/pvn3d_hip.h).  There is
no CPU fallback: without the library, or with CPU tensors, the ops raise.
"""
__version__ = "0.1.0"
