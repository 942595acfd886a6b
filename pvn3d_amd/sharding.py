"""Frame sharding over the GPUs of one node.

Frames are independent in every op of the hot path, so the multi-GPU path is data-parallel
with NO collective inside the path: rank r owns a contiguous block of frames, runs the whole
pipeline locally, and the per-frame results (a 3x4 pose + (K+1) keypoints + iteration counts,
~50 numbers per frame) are gathered once per batch.  One process per GPU, torch.distributed
("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (first ranks get the remainder)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_frame_results(local, n_items, group=None):
    """All-gather per-frame result rows: `local` (n_local, D) on every rank -> (n_items, D) in
    frame order on every rank.  Ranks may own different numbers of frames (padded exchange)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    ws = dist.get_world_size(group)
    sizes = [shard_range(n_items, r, ws) for r in range(ws)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.size(0)] = local
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)
