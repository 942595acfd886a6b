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


def _single(group, skip_single):
    """No process group, or a group of one with nothing to exchange.  skip_single=False runs the collective
    path even then (tests/test_gpu_rccl.py drives RCCL on a one-GPU box that way)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return skip_single and dist.get_world_size(group) == 1


def gather_frame_results(local, n_items, group=None, skip_single=True):
    """All-gather per-frame result rows: `local` (n_local, D) on every rank -> (n_items, D) in
    frame order on every rank.  Ranks may own different numbers of frames (padded exchange)."""
    if _single(group, skip_single):
        return local
    ws = dist.get_world_size(group)
    sizes = [shard_range(n_items, r, ws) for r in range(ws)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.size(0)] = local
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], 0)


# ---------------------------------------------------------------------------------------------
# Training (BASELINE config 5): data-parallel gradient averaging, one process per GPU.
# The reference wraps the model in nn.DataParallel (train/train_linemod_pvn3d.py:480): one
# process scatters the batch, replicates the weights every step and reduces gradients to GPU 0.
# Here every rank owns a model replica and the gradients are all-reduced in a few large buckets.
# xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring all-reduce is bound by one
# link: buckets are large (default 64 MiB, a few hundred microseconds of wire time each) to
# amortise the per-collective latency, and are issued asynchronously in reverse parameter order
# (the order backward produces them) so the first buckets fly while later ones are still packed.
# ---------------------------------------------------------------------------------------------

def all_reduce_gradients(parameters, bucket_bytes=64 << 20, group=None, average=True, skip_single=True):
    """Average (or sum) `.grad` of `parameters` over the process group in place.
    Returns the number of buckets used.  Parameters without a gradient are skipped on every rank
    alike (the model is the same on all ranks)."""
    if _single(group, skip_single):
        return 0
    ws = dist.get_world_size(group)
    params = [p for p in parameters if p.grad is not None]
    params.reverse()
    buckets, cur, cur_bytes = [], [], 0
    for p in params:
        nb = p.grad.numel() * p.grad.element_size()
        if cur and (cur_bytes + nb > bucket_bytes or p.grad.dtype != cur[0].grad.dtype):
            buckets.append(cur)
            cur, cur_bytes = [], 0
        cur.append(p)
        cur_bytes += nb
    if cur:
        buckets.append(cur)
    pending = []
    for bk in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in bk])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
        pending.append((work, flat, bk))
    for work, flat, bk in pending:
        work.wait()
        if average:
            flat.div_(ws)
        off = 0
        for p in bk:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
    return len(buckets)


def broadcast_parameters(module, src=0, group=None, skip_single=True):
    """Make every rank start from rank `src`'s weights and buffers (what DataParallel's per-step
    replication guarantees implicitly)."""
    if _single(group, skip_single):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)
    # the collective writes through data_ptr without bumping tensor._version: cached folded Conv+BN weights
    # of the fused inference kernels would otherwise survive the update
    from .lib.pointnet2_utils import _fused_mlp
    _fused_mlp.invalidate_packed()
