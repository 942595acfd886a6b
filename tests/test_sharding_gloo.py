"""N>1 path on CPU: world_size-2 gloo processes shard frames and gather per-frame rows."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from pvn3d_amd.sharding import shard_range, gather_frame_results
    lo, hi = shard_range(n_items, rank, ws)
    # each rank "computes" rows = frame id replicated
    local = torch.arange(lo, hi, dtype=torch.float64).view(-1, 1).repeat(1, 5)
    full = gather_frame_results(local, n_items)
    ok = bool(torch.equal(full, torch.arange(n_items, dtype=torch.float64).view(-1, 1).repeat(1, 5)))
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, ok, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_items, ws=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, ws, port, n_items, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    assert all(t == float(ws) for _, _, t in res)


def test_gather_even():
    _run(8)


def test_gather_ragged():
    _run(7)
