"""the reference's OWN operator file
(pvn3d/lib/pointnet2_utils/pointnet2_utils.py) imports and wires up against our `_ext` when
`lib.pointnet2_utils._ext` is substituted -- i.e. the reference's callers need no edit.
Runs only where /root/reference exists (the build container); GPU execution of the ops is
covered by tests/test_gpu_ops.py."""
import importlib.util
import os
import sys
import types

import pytest

REF = "/root/reference/pvn3d/lib/pointnet2_utils/pointnet2_utils.py"


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
def test_reference_pointnet2_utils_runs_on_our_ext():
    import pvn3d_amd.lib.pointnet2_utils as ours
    saved = {k: sys.modules.get(k) for k in
             ["lib", "lib.pointnet2_utils", "lib.pointnet2_utils._ext", "lib.utils",
              "lib.utils.etw_pytorch_utils"]}
    try:
        lib = types.ModuleType("lib"); lib.__path__ = []
        lp = types.ModuleType("lib.pointnet2_utils"); lp.__path__ = []
        lp._ext = ours._ext
        lu = types.ModuleType("lib.utils"); lu.__path__ = []
        etw = types.ModuleType("lib.utils.etw_pytorch_utils")     # only used by RandomDropout
        sys.modules.update({"lib": lib, "lib.pointnet2_utils": lp, "lib.pointnet2_utils._ext": ours._ext,
                            "lib.utils": lu, "lib.utils.etw_pytorch_utils": etw})
        spec = importlib.util.spec_from_file_location("ref_pointnet2_utils", REF)
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        assert ref._ext is ours._ext
        for name in ["furthest_point_sample", "gather_operation", "three_nn", "three_interpolate",
                     "grouping_operation", "ball_query", "QueryAndGroup", "GroupAll"]:
            assert hasattr(ref, name) and hasattr(ours.pointnet2_utils, name)
        import torch
        with pytest.raises(RuntimeError, match="CPU not supported"):   # reaches our shim
            ref.ball_query(0.1, 4, torch.zeros(1, 8, 3), torch.zeros(1, 2, 3))
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
