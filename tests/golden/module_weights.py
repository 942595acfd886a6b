"""State-dict values for the module-level reference fixture (pointnet2msg_ref.npz) as a pure function of
(key order, shapes, seed) on numpy's frozen legacy stream (np.random.RandomState: its bit stream is guaranteed
stable across numpy versions).  Used by make_golden_modules.py (which loads them into the REFERENCE's modules with
strict=True and records their SHA-256) and by the tests (which load them into this package's modules and check
the same SHA-256): the 3.3 M parameters of Pointnet2MSG need not be stored."""
import hashlib

import numpy as np


def weights(keys, shapes, seed, style="benign"):
    """state_dict values as a pure function of (key order, shapes, seed, style) on numpy's frozen legacy stream
    (np.random.RandomState: bit stream guaranteed stable).  style "benign": BatchNorm scale and variance in [0.5, 1.5]
    (rounds 4-5); "trained" (round 6): running_var log-uniform over 1e-6 .. 1e2 with the conv rows and running means scaled
    by sqrt(var) -- a BatchNorm's statistics ARE those of its input, which also keeps the activations of a 24-layer
    network inside fp32's range -- and gamma log-uniform over 1e-3 .. 3: the raw tensors span eight decades per layer, the
    folded rows gamma / sqrt(var + eps) . W three and a half (channels a trained network has all but pruned beside
    channels at full scale)."""
    assert style in ("benign", "trained")
    rs = np.random.RandomState(seed)
    out = {}
    for k, shp in zip(keys, shapes):
        if k.endswith("conv.weight"):
            fan_in = shp[1]
            v = rs.standard_normal(size=shp) * np.sqrt(2.0 / fan_in)
        elif k.endswith("bn.weight"):
            v = rs.uniform(0.5, 1.5, size=shp) if style == "benign" else 10.0 ** rs.uniform(-3.0, np.log10(3.0), size=shp)
        elif k.endswith("bn.bias"):
            v = rs.standard_normal(size=shp) * 0.1
        elif k.endswith("running_mean"):
            v = rs.standard_normal(size=shp) * 0.2
        elif k.endswith("running_var"):
            v = rs.uniform(0.5, 1.5, size=shp) if style == "benign" else 10.0 ** rs.uniform(-6.0, 2.0, size=shp)
        elif k.endswith("num_batches_tracked"):
            out[k] = np.zeros(shp, np.int64)
            continue
        else:
            raise KeyError(k)
        out[k] = v.astype(np.float32)
    if style == "trained":
        for k in keys:
            if k.endswith("normlayer.bn.running_var"):
                pre = k[:-len("normlayer.bn.running_var")]
                sd = np.sqrt(out[k].astype(np.float64))
                out[pre + "conv.weight"] = (out[pre + "conv.weight"] * sd.reshape(-1, 1, 1, 1)).astype(np.float32)
                out[pre + "normlayer.bn.running_mean"] = (out[pre + "normlayer.bn.running_mean"] * sd).astype(np.float32)
    return out


def weights_sha(keys, w):
    h = hashlib.sha256()
    for k in keys:
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes())
    return h.hexdigest()
