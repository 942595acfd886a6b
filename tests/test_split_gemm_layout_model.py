"""CPU: lane-level model of the LDS-DMA split GEMM's staging (csrc/split_gemm.hip: sg_gemm_dma_kernel).

`global_load_lds_dwordx4` writes lane l's 16 bytes at (wave-uniform base) + 16 l: the LDS image of an instruction is
lane-linear, so a swizzle can only be applied through the SOURCE addresses, and the reader has to apply the same one.
This test restates the kernel's three index maps in numpy --

    loader   instruction t of wave w covers rows R0 .. R0 + 15 of an operand; lane l fetches 16-byte part
             (l & 3) ^ ((r >> 2) & 3) of row r = R0 + (l >> 2) of the h16 matrix (a 16-k slab of a row = 64 B = 4 parts:
             piece p, k half h -> part 2 p + h) and lands at byte 64 r + 16 (l & 3) of the stage;
    reader   the MFMA fragment of piece p for lane (row m = lane & 31, k half h = lane >> 5) of a 32-row block is the 16
             bytes at 64 (block rows + m) + 16 (((2 p) | h) ^ ((m >> 2) & 3));
    banks    ds_read_b128 serves a wave in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the
             same + 32: MI355X guide, LDS table); bank of a dword = (byte address / 4) mod 64

-- and checks (i) every fragment read returns exactly the slab bytes the MFMA expects, for both tile shapes and every
wave, and (ii) every lane group of every fragment read touches 64 distinct banks (conflict-free).  The kernel's bits are
pinned on the GPU against the register-staged kernel (tests/test_gpu_ops.py); this pins the arithmetic of the layout
where it can be read."""
import numpy as np
import pytest

SG_T = 128
GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
GROUPS = GROUPS + [[l + 32 for l in g] for g in GROUPS]


def _swz(r):
    return (r >> 2) & 3


def _stage_image(W, X, TJ):
    """W (128, 64) / X (64 TJ, 64) uint8: one 16-k slab of the tile's rows -> the stage's LDS bytes as the DMA lands them"""
    WB, PT = SG_T * 64, 64 * TJ
    lds = np.full(WB + PT * 64, 0xEE, np.uint8)
    for wave in range(4):
        jobs = [(W, 0, wave * 32 + t * 16) for t in range(2)] + [(X, WB, wave * 16 * TJ + t * 16) for t in range(TJ)]
        for mat, region, R0 in jobs:
            for lane in range(64):
                r = R0 + (lane >> 2)
                part = (lane & 3) ^ _swz(r)
                dst = region + R0 * 64 + lane * 16                 # lane-linear destination of the instruction
                lds[dst:dst + 16] = mat[r, part * 16: part * 16 + 16]
    return lds


@pytest.mark.parametrize("TJ", [2, 4])
def test_source_side_swizzle_and_fragment_reads_agree_and_are_conflict_free(TJ):
    rng = np.random.default_rng(TJ)
    W = rng.integers(0, 256, (SG_T, 64), dtype=np.uint8)
    X = rng.integers(0, 256, (64 * TJ, 64), dtype=np.uint8)
    lds = _stage_image(W, X, TJ)
    assert not np.any(lds == 0xEE) or np.count_nonzero(lds == 0xEE) < lds.size // 64      # every byte of the stage written
    WB = SG_T * 64
    for wave in range(4):
        wr, wc = wave & 1, wave >> 1
        reads = [("W", W, 0, wr * 64 + i * 32) for i in range(2)] + [("X", X, WB, wc * 32 * TJ + j * 32) for j in range(TJ)]
        for name, mat, region, row0 in reads:
            for p in range(2):
                addr = np.zeros(64, np.int64)
                for lane in range(64):
                    m, h = lane & 31, lane >> 5
                    col = ((2 * p) | h) ^ _swz(m)
                    a = region + (row0 + m) * 64 + col * 16
                    addr[lane] = a
                    want = mat[row0 + m, (2 * p + h) * 16: (2 * p + h) * 16 + 16]     # piece p, k half h of the slab
                    assert np.array_equal(lds[a:a + 16], want), (name, wave, row0, p, lane)
                # the kernel derives piece 1's address from piece 0's as +-32 bytes: column ^ 2
                if p == 1:
                    for lane in range(64):
                        m, h = lane & 31, lane >> 5
                        c0 = h ^ _swz(m)
                        px = 32 - 2 * (c0 & 2) * 16
                        assert addr[lane] == region + (row0 + m) * 64 + c0 * 16 + px
                for g in GROUPS:
                    banks = np.concatenate([((addr[l] // 4) + np.arange(4)) % 64 for l in g])
                    assert len(set(banks.tolist())) == 64, (name, wave, p, g)


def test_ring_offsets_cover_three_stages_and_the_constants_behind_them():
    """Dynamic LDS of a launch: three stages + 1 KiB of per-channel constants; two workgroups of the wide form fit a CU."""
    for TJ, per_cu in ((4, 2), (2, 3)):
        stage = (SG_T + 64 * TJ) * 64
        total = 3 * stage + 1024
        assert total == 3 * (SG_T + 64 * TJ) * 64 + 1024
        assert per_cu * total <= 160 * 1024
        rb, seen = 0, []
        for _ in range(7):                                     # rb = rb == 2 STAGE ? 0 : rb + STAGE
            seen.append(rb)
            rb = 0 if rb == 2 * stage else rb + stage
        assert seen == [0, stage, 2 * stage, 0, stage, 2 * stage, 0]
