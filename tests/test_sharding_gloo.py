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


def _grad_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from pvn3d_amd.sharding import all_reduce_gradients, broadcast_parameters
    torch.manual_seed(100 + rank)                     # ranks start from DIFFERENT weights ...
    net = torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.ReLU(), torch.nn.BatchNorm1d(33),
                              torch.nn.Linear(33, 5))
    broadcast_parameters(net)                         # ... and are synchronised to rank 0's
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()
    torch.manual_seed(7)
    x_all = torch.randn(2 * 6, 7)
    y_all = torch.randn(2 * 6, 5)
    x, y = x_all[rank * 6:(rank + 1) * 6], y_all[rank * 6:(rank + 1) * 6]
    loss = ((net(x) - y) ** 2).mean()
    loss.backward()
    local = [p.grad.clone() for p in net.parameters()]
    n_buckets = all_reduce_gradients(net.parameters(), bucket_bytes=600)    # tiny buckets: several of them
    mine = torch.cat([g.reshape(-1) for g in local])
    gathered = [torch.empty_like(mine) for _ in range(ws)]
    dist.all_gather(gathered, mine)
    want = sum(gathered) / ws
    got = torch.cat([p.grad.reshape(-1) for p in net.parameters()])
    w_all = [torch.empty_like(w0) for _ in range(ws)]
    dist.all_gather(w_all, w0)
    q.put((rank, bool(torch.allclose(got, want, rtol=1e-6, atol=1e-7)), n_buckets,
           bool(all(torch.equal(w, w_all[0]) for w in w_all))))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce_and_weight_broadcast():
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _, _ in res)
    assert all(nb >= 2 for _, _, nb, _ in res)            # the bucket logic really split
    assert all(same for _, _, _, same in res)


def _bench_gather_worker(rank, ws, port, frames_total, strong, q):
    """bench.py's per-step collective (gather_step_results) on CPU tensors under gloo: the same code the
    driver's N>1 runs execute over RCCL."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    import bench
    from pvn3d_amd.sharding import shard_range
    K = 8
    if strong:
        lo, hi = shard_range(frames_total, rank, ws)
    else:
        lo, hi = rank * frames_total, (rank + 1) * frames_total       # weak: frames_total per rank
    n = hi - lo
    fid = torch.arange(lo, hi, dtype=torch.float64)
    res = dict(poses=fid.view(n, 1, 1).expand(n, 3, 4).clone(),                       # (n,3,4) float64 like the engine
               cls_kps=(fid.view(n, 1, 1).expand(n, K + 1, 3) + 0.5).float().clone(),
               iters=torch.full((n, K + 1), 4, dtype=torch.int32))
    got = bench.gather_step_results(res, n, frames_total, ws, strong)
    if strong:
        ok = got.shape == (frames_total, 12 + 3 * (K + 1) + (K + 1)) and \
            torch.equal(got[:, 0], torch.arange(frames_total, dtype=torch.float32))
    else:
        ok = len(got) == ws and all(torch.equal(got[r][:, 0], torch.arange(r * frames_total, (r + 1) * frames_total,
                                                                         dtype=torch.float32)) for r in range(ws))
    q.put((rank, bool(ok), 0.0))
    dist.barrier()
    dist.destroy_process_group()


def _run_bench_gather(frames_total, strong, ws=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_gather_worker, args=(r, ws, port, frames_total, strong, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)


def test_bench_step_gather_strong_ragged():
    _run_bench_gather(7, True)


def test_bench_step_gather_weak():
    _run_bench_gather(4, False)


def _overlap_worker(rank, ws, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from pvn3d_amd.sharding import OverlappedGradientReducer, all_reduce_gradients, broadcast_parameters
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(7, 64), torch.nn.ReLU(), torch.nn.Linear(64, 64), torch.nn.ReLU(),
                              torch.nn.Linear(64, 33), torch.nn.ReLU(), torch.nn.Linear(33, 5))
    unused = torch.nn.Parameter(torch.zeros(11))               # a parameter that never receives a gradient
    broadcast_parameters(net)
    params = list(net.parameters()) + [unused]
    torch.manual_seed(70 + rank)
    x, y = torch.randn(6, 7), torch.randn(6, 5)

    def backward():
        for p in params:
            p.grad = None
        ((net(x) - y) ** 2).mean().backward()

    backward()                                                  # reference: exchange after backward has returned
    all_reduce_gradients(params, bucket_bytes=6000)
    want = [p.grad.clone() for p in net.parameters()]
    red = OverlappedGradientReducer(params, bucket_bytes=6000)  # several buckets
    in_flight = []
    h = net[0].weight.register_hook(lambda g: in_flight.append(red.launched_during_backward))   # first layer: last gradient
    backward()                                                  # not armed: the hooks must not issue anything
    idle = red.launched_during_backward == 0 and not red._pending
    red.arm()
    backward()
    h.remove()
    during = red.launched_during_backward
    n = red.finalize()
    got = [p.grad.clone() for p in net.parameters()]
    same = all(torch.equal(a, b) for a, b in zip(got, want))    # same buckets, same summation: identical bits
    red.arm()
    backward()                                                  # a second step reuses the reducer
    n2 = red.finalize()
    same2 = all(torch.equal(p.grad, b) for p, b in zip(net.parameters(), want))
    # gradient accumulation inside one armed window: two ANNOUNCED backward passes, every bucket on the wire ONCE, issued
    # from inside the second pass
    for p in params:
        p.grad = None
    issued0, during0 = red.collectives_issued, red.launched_during_backward
    red.arm(n_backward=2)
    ((net(x) - y) ** 2).mean().backward()
    after_first = red.collectives_issued - issued0
    ((net(x) - y) ** 2).mean().backward()
    n_acc = red.finalize()
    acc_ok = all(torch.allclose(p.grad, 2 * b, rtol=1e-6, atol=1e-7) for p, b in zip(net.parameters(), want))
    acc_ok = acc_ok and after_first == 0 and red.collectives_issued - issued0 == n_acc == n and \
        red.launched_during_backward - during0 >= 1
    # an unannounced second backward is an error on every rank alike (never a rank-local extra collective)
    for p in params:
        p.grad = None
    red.arm()
    ((net(x) - y) ** 2).mean().backward()
    try:
        ((net(x) - y) ** 2).mean().backward()
        acc_ok = False
    except RuntimeError:
        pass
    red._armed = True                                           # (end the window properly: every rank waits for its buckets)
    red.finalize()
    try:
        red.finalize()
        unarmed_raises = False
    except RuntimeError:
        unarmed_raises = True
    q.put((rank, same and same2 and idle and acc_ok and unarmed_raises, len(red.buckets), n, n2, during,
           in_flight[-1] if in_flight else -1, unused.grad is None))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_buckets_are_exchanged_during_backward():
    """OverlappedGradientReducer: the all-reduce of a bucket starts from inside backward() as soon as its last gradient
    has been accumulated (before the first layer's gradient even exists), the results equal the after-backward
    exchange bit for bit, and a parameter without a gradient neither blocks nor receives one."""
    ws = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overlap_worker, args=(r, ws, port, q)) for r in range(ws)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(ws)]
    for p in procs:
        p.join(timeout=60)
    for _, same, n_buckets, n, n2, during, before_first_layer, unused_none in res:
        assert same and unused_none
        assert n_buckets >= 3 and n == n2 and 1 <= n <= n_buckets
        assert during >= 1                      # issued while backward was still running ...
        assert before_first_layer >= 1          # ... before the first layer's gradient had been produced
