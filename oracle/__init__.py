"""CPU oracle for the PVN3D hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this package.  The shipped path (``pvn3d_amd``) never imports it and never falls back
to it: without the HIP library ``pvn3d_amd`` raises.

Contents
--------
``native``      ctypes binding of ``libpvn3d_oracle.so`` (``pvn3d_oracle.c``): C restatement
                of every op in pvn3d/_ext-src plus MeanShiftTorch.fit and best_fit_transform.
``posecal``     numpy/torch-CPU restatement of ``cal_frame_poses`` / ``cal_frame_poses_lm``
                (pvn3d/lib/utils/pvn3d_eval_utils.py:37-110,156-201) on top of ``native``.
``torch_port``  dense O(n^2) torch-CPU restatement of MeanShiftTorch.fit -- same tensor
                algebra as the reference, used as the timed "reference CPU path".
``synth``       seeded synthetic frames.
"""
