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


def test_host_side_planning_functions():
    """Entry points that only plan a launch (no device work): FPS workspace size, split-K choice of the small-batch
    linear layer, eligibility of the row-major weight-gradient kernel, partial-sum counts."""
    from pvn3d_amd import _lib
    lib = _lib.lib
    # FPS: the culled kernel's workspace is z by cell position (64 cells x 64 x spc) + one index per point
    for n, spc in ((4097, 2), (8192, 2), (8193, 3), (12288, 3)):
        assert lib.pvn3d_fps_ws_words(n) == 64 * spc * 64 + n
    assert lib.pvn3d_fps_ws_words(4096) >= 0
    # small-batch linear layer: many output tiles or a short K -> no split; few tiles and a long K -> up to 8 slices of
    # at least 128
    assert lib.pvn3d_sb_linear_splits(4096, 512, 516) == 1            # 512 tiles already
    assert lib.pvn3d_sb_linear_splits(512, 512, 1536) == 8            # FP level 3 at one frame: 64 tiles
    assert lib.pvn3d_sb_linear_splits(512, 512, 200) == 1             # K too short to cut
    s = lib.pvn3d_sb_linear_splits(1024, 512, 768)
    assert s in (2, 4) and 768 // s >= 128
    assert lib.pvn3d_sb_linear_splits(0, 5, 5) == 1
    # weight gradient straight from row-major matrices: layers up to 512 x 544 channels
    assert lib.pvn3d_mt_wgrad_tn_ok(16, 9) == 1 and lib.pvn3d_mt_wgrad_tn_ok(512, 528) == 1
    assert lib.pvn3d_mt_wgrad_tn_ok(512, 1536) == 0 and lib.pvn3d_mt_wgrad_tn_ok(0, 8) == 0
    # partial rows of the two-stage BatchNorm reductions: bounded, monotone, one per <= rows
    p_small, p_big = lib.pvn3d_mt_bn_bwd_partials(100), lib.pvn3d_mt_bn_bwd_partials(1572864)
    assert 1 <= p_small <= 2 and p_small <= p_big <= 1024
    assert 1 <= lib.pvn3d_mt_gemm_nt_stat_rows(1572864) <= 1024 and lib.pvn3d_mt_gemm_nt_stat_rows(100) == 1
