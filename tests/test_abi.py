"""The C-ABI library loads and exports every symbol include/pvn3d_hip.h declares (no GPU)."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "pvn3d_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pvn3d_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    names = _declared()
    assert len(names) >= 17
    lib = ctypes.CDLL(os.path.join(ROOT, "pvn3d_amd", "libpvn3d_hip.so"))
    for n in names:
        assert hasattr(lib, n), "missing export: %s" % n


def test_loader_signatures_cover_header():
    from pvn3d_amd import _lib
    assert sorted(_lib.SIGNATURES.keys()) == _declared()
    assert _lib.lib.pvn3d_abi_version() == 1


def test_opt_n_threads_matches_reference_formula(orc):
    from pvn3d_amd import _lib
    for w in [1, 2, 3, 9, 63, 64, 127, 128, 511, 512, 513, 1024, 2048, 12288, 100000]:
        assert _lib.lib.pvn3d_opt_n_threads(w) == orc.opt_n_threads(w)
    assert _lib.lib.pvn3d_opt_n_threads(12288) == 512
    assert _lib.lib.pvn3d_opt_n_threads(128) == 128


def test_workspace_bytes_monotone():
    from pvn3d_amd import _lib
    a = _lib.lib.pvn3d_meanshift_workspace_bytes(9, 9 * 3072, 300)
    b = _lib.lib.pvn3d_meanshift_workspace_bytes(9, 9 * 12288, 300)
    assert 0 < a < b
    assert a >= 2 * 16 * 9 * 3072
