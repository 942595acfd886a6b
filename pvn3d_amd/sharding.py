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
# (the order backward produces them).  all_reduce_gradients() does it after backward() has returned;
# OverlappedGradientReducer issues every bucket from inside backward, as soon as its last gradient exists
# (train_step uses it).
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


class OverlappedGradientReducer(object):
    """The same bucketed gradient averaging, issued DURING backward: every parameter carries a post-accumulate-grad
    hook; when the last gradient of a bucket has been accumulated the bucket is flattened and its all-reduce is
    started asynchronously (RCCL runs it on its own stream over xGMI), so the buckets of the late layers -- the first
    ones backward produces -- are on the wire while the early layers' gradients are still being computed.  The
    reference reduces inside nn.DataParallel's backward (train_linemod_pvn3d.py:480); this is the one-process-per-GPU
    equivalent.  Buckets are laid out once, in reverse parameter order, and are issued in the order they complete:
    every rank runs the same model, so autograd visits the same graph in the same order and the ranks issue the same
    collectives in the same order (a rank-dependent graph -- data-dependent control flow in the model -- would break
    that, exactly as it breaks DistributedDataParallel without find_unused_parameters).  ``finalize()`` (after
    ``backward()``) issues what is left in bucket order -- buckets holding a parameter that received no gradient
    this step, which therefore never completed -- waits, averages and writes the results back into ``.grad``.

        red = OverlappedGradientReducer(model.parameters(), group=group)     # once
        red.arm(); loss.backward(); red.finalize(); optimizer.step()          # every step

    The hooks stay on the parameters but only act between ``arm()`` and ``finalize()``: a backward outside a training
    step (eval-time gradients, a loop that exchanges with ``all_reduce_gradients``) issues no collective, so ranks
    cannot be left with unmatched all-reduces.  Gradient accumulation: ``arm(n_backward=k)`` announces k backward
    passes inside the window; a bucket is issued once, when its parameters have accumulated their k-th gradient (from
    inside the LAST micro-batch's backward), so every bucket crosses the wire once and the overlap is kept.  More backward
    passes than announced raise (a rank-local "this bucket is stale" decision would let ranks issue different numbers of
    collectives -- a deadlock, not an error message).
    """

    def __init__(self, parameters, bucket_bytes=64 << 20, group=None, average=True):
        self.group, self.average, self.bucket_bytes = group, average, bucket_bytes
        params = [p for p in parameters if p.requires_grad]
        params.reverse()
        self.buckets, cur, cur_bytes = [], [], 0
        for p in params:
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > bucket_bytes or p.dtype != cur[0].dtype):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {}
        for bi, bk in enumerate(self.buckets):
            for p in bk:
                self._bucket_of[id(p)] = bi
        self._handles = [p.register_post_accumulate_grad_hook(self._on_grad) for p in params]
        self.launched_during_backward = 0
        self.collectives_issued = 0         # all-reduces issued over the reducer's life (bench: a run without exchange adds 0)
        self._armed = False
        self._n_backward = 1
        self._reset()

    def _reset(self):
        self._ready = [0] * len(self.buckets)
        self._issued = [False] * len(self.buckets)
        self._pending = []                  # (work, flat, params with a gradient, bucket index)
        self._in_backward = True

    def arm(self, n_backward=1):
        """The next `n_backward` backward() calls accumulate; buckets are exchanged as they complete in the last of them;
        ``finalize()`` ends the window."""
        if n_backward < 1:
            raise ValueError("n_backward must be >= 1")
        self._reset()
        self._n_backward = int(n_backward)
        self._armed = True
        return self

    def _issue(self, bi):
        self._issued[bi] = True
        bk = [p for p in self.buckets[bi] if p.grad is not None]
        if not bk:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in bk])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.collectives_issued += 1
        self._pending.append((work, flat, bk, bi))
        if self._in_backward:
            self.launched_during_backward += 1

    def _on_grad(self, p):
        if not self._armed:
            return
        bi = self._bucket_of[id(p)]
        self._ready[bi] += 1
        if self._issued[bi]:
            self._armed = False             # (no further collective from this window)
            raise RuntimeError("more backward passes inside one armed window than announced: arm(n_backward=k)")
        if self._ready[bi] == len(self.buckets[bi]) * self._n_backward:
            self._issue(bi)

    def finalize(self):
        """Call after backward(): issue the remaining buckets, wait for all, average, write back.  Returns the number
        of buckets that were exchanged."""
        if not self._armed:
            raise RuntimeError("OverlappedGradientReducer.finalize() without arm(): no backward was watched")
        self._in_backward = False
        self._armed = False
        for bi in range(len(self.buckets)):
            if not self._issued[bi]:
                self._issue(bi)
        ws = dist.get_world_size(self.group)
        n = len(self._pending)
        for work, flat, bk, bi in self._pending:
            work.wait()
            if self.average:
                flat.div_(ws)
            off = 0
            for p in bk:
                k = p.grad.numel()
                p.grad.copy_(flat[off:off + k].view_as(p.grad))
                off += k
        self._reset()
        return n

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []
        self._armed = False


def overlapped_reducer(model, bucket_bytes=64 << 20, group=None, skip_single=True):
    """The model's OverlappedGradientReducer (created on first use, kept on the model; rebuilt when the group or the
    bucket size changes), or None when there is no process group / a group of one (nothing to exchange).  The caller
    arms it for one backward: ``red.arm(); loss.backward(); red.finalize()``."""
    if _single(group, skip_single):
        return None
    red = getattr(model, "_pvn3d_grad_reducer", None)
    if red is None or red.group is not group or red.bucket_bytes != bucket_bytes:
        if red is not None:
            red.remove()
        red = OverlappedGradientReducer(model.parameters(), bucket_bytes=bucket_bytes, group=group)
        model._pvn3d_grad_reducer = red
    return red


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
