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
``metrics``     numpy restatement of ADD / ADD-S / VOCap / cal_auc (basic_utils.py:32-44, 597-635).
``loss``        numpy restatement of of_l1_loss and its gradient (lib/loss.py:45-73).

``ref``         ctypes binding of ``oracle/_ref/libpvn3d_ref_{nofma,fma}.so``: the REFERENCE'S OWN
                kernels (pvn3d/_ext-src/src/*_gpu.cu) compiled for the CPU from where they lie under
                /root/reference by ``ref_shim/build_ref.py`` (CUDA execution model supplied by
                ``ref_shim/``: threads as fibers, real block barriers).  Built in the development
                container, shipped to the GPU box as prebuilt files, never committed.

Pinning: every restatement is checked against outputs of the reference itself --
the Python pieces against tests/golden/*_ref.npz (tests/golden/make_golden.py), the nine native ops
bit-for-bit against ``ref`` (tests/test_oracle_vs_ref.py) and against the fixture the reference's
kernels produced at the BASELINE shapes (tests/golden/native_ref.npz, make_golden_native.py).
"""
