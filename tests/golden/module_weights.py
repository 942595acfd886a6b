"""State-dict values for the module-level reference fixture (pointnet2msg_ref.npz) as a pure function of
(key order, shapes, seed) on numpy's frozen legacy stream (np.random.RandomState: its bit stream is guaranteed
stable across numpy versions).  Used by make_golden_modules.py (which loads them into the REFERENCE's modules with
strict=True and records their SHA-256) and by the tests (which load them into this package's modules and check
the same SHA-256): the 3.3 M parameters of Pointnet2MSG need not be stored."""
import hashlib

import numpy as np


def weights(keys, shapes, seed):
    """state_dict values as a pure function of (key order, shapes, seed) on numpy's frozen legacy stream
    (np.random.RandomState: bit stream guaranteed stable)."""
    rs = np.random.RandomState(seed)
    out = {}
    for k, shp in zip(keys, shapes):
        if k.endswith("conv.weight"):
            fan_in = shp[1]
            v = rs.standard_normal(size=shp) * np.sqrt(2.0 / fan_in)
        elif k.endswith("bn.weight"):
            v = rs.uniform(0.5, 1.5, size=shp)
        elif k.endswith("bn.bias"):
            v = rs.standard_normal(size=shp) * 0.1
        elif k.endswith("running_mean"):
            v = rs.standard_normal(size=shp) * 0.2
        elif k.endswith("running_var"):
            v = rs.uniform(0.5, 1.5, size=shp)
        elif k.endswith("num_batches_tracked"):
            out[k] = np.zeros(shp, np.int64)
            continue
        else:
            raise KeyError(k)
        out[k] = v.astype(np.float32)
    return out


def weights_sha(keys, w):
    h = hashlib.sha256()
    for k in keys:
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()
