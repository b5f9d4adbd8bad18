"""CPU, world_size 2, gloo: the data-parallel plumbing of fcaf3d_amd/dist.py (reduce_mean and the bucketed
gradient averager that bench.py uses on RCCL) — multi-process, rendezvous on 127.0.0.1."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from fcaf3d_amd import dist as D
    D.init_dist(backend='gloo')
    assert D.is_dist() and D.world_size() == world
    # reduce_mean == mmdet.core.reduce_mean
    t = torch.tensor([float(rank + 1), 10.0 * (rank + 1)])
    r = D.reduce_mean(t)
    assert torch.allclose(r, torch.tensor([1.5, 15.0])), r
    assert torch.equal(t, torch.tensor([float(rank + 1), 10.0 * (rank + 1)]))      # input untouched
    # the batched normalisers of Fcaf3DNeckWithHead.loss travel in ONE all-reduce
    norms = torch.arange(6, dtype=torch.float32).reshape(3, 2) * (rank + 1)
    assert torch.allclose(D.reduce_mean(norms), torch.arange(6, dtype=torch.float32).reshape(3, 2) * 1.5)
    # bucketed gradient averaging: tiny buckets force several async all-reduces + the flush of unused params
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    unused = torch.nn.Parameter(torch.ones(5))
    params = list(model.parameters()) + [unused]
    avg = D.GradientAverager(params, bucket_mb=1e-5)
    assert len(avg.buckets) >= 3
    for step in range(2):                                   # hooks must re-arm after finish()
        for p in params:
            p.grad = None
        x = torch.full((3, 8), float(rank + 1 + step))
        model(x).sum().backward()
        local = [p.grad.clone() for p in model.parameters()]
        avg.finish()
        gathered = [[torch.zeros_like(g) for _ in range(world)] for g in local]
        for g, lst in zip(local, gathered):
            dist.all_gather(lst, g)
        for p, lst in zip(model.parameters(), gathered):
            assert torch.allclose(p.grad, sum(lst) / world, atol=1e-6)
        assert torch.equal(unused.grad, torch.zeros(5))
    # a parameter that receives a gradient on ONE rank only: its bucket completes at different times on the two ranks;
    # the collectives must still be issued in the same (index) order everywhere (wrong sums / a hang before)
    torch.manual_seed(1)
    a, b, c = (torch.nn.Parameter(torch.randn(6)) for _ in range(3))
    avg2 = D.GradientAverager([a, b, c], bucket_mb=1e-5)             # one bucket per parameter, order c, b, a
    assert len(avg2.buckets) == 3
    for step in range(2):
        for p in (a, b, c):
            p.grad = None
        loss = (a * (rank + 1)).sum() + (c * 3.0).sum()
        if rank == 0:
            loss = loss + (b * 5.0).sum()                              # b: gradient on rank 0 only
        loss.backward()
        avg2.finish()
        assert torch.allclose(a.grad, torch.full((6,), 1.5)) and torch.allclose(c.grad, torch.full((6,), 3.0))
        assert torch.allclose(b.grad, torch.full((6,), 2.5)), b.grad
    if rank == 0:
        out.put('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_reduce_mean_and_gradient_averager_world2():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == 'ok'


def test_single_process_is_identity():
    from fcaf3d_amd import dist as D
    t = torch.tensor([3.0])
    assert D.reduce_mean(t) is t and D.world_size() == 1
    avg = D.GradientAverager([torch.nn.Parameter(torch.ones(2))])
    avg.finish()                                             # no-op without a process group


def test_bench_step_count_is_rank_independent():
    """every rank must execute the same number of steps (each holds collectives): r1 shipped a rank-0-only
    FLOP-count step that deadlocked N>1; the extra-step count is now a pure function of the flags"""
    import inspect
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    assert list(inspect.signature(bench.count_steps).parameters) == ['args', 'n_batches']
    assert list(inspect.signature(bench.hbm_steps).parameters) == ['args', 'exec_on']
    src = inspect.getsource(bench.main)
    assert 'for b in range(count_steps(args, len(batches)))' in src and 'for b in range(hbm_steps(args, exec_on))' in src
    # `exec_on` itself is a function of the flags / environment only
    assert 'exec_on = EX.ENABLED and not args.no_executor' in src
    # the loops that run them are not nested under a rank / probe condition
    for key in ('count_steps(args, len(batches))', 'hbm_steps(args, exec_on)'):
        line = [l for l in src.splitlines() if key in l][0]
        assert line.startswith('    for '), 'extra-step loops must sit at function level, outside any rank-dependent branch'
    argv = sys.argv
    try:
        sys.argv = ['bench.py']
        a = bench.parse()
        assert bench.count_steps(a, 2) == 2
        assert bench.hbm_steps(a, True) == 1 and bench.hbm_steps(a, False) == 0
        sys.argv = ['bench.py', '--no-instrument']
        assert bench.count_steps(bench.parse(), 2) == 0
        sys.argv = ['bench.py', '--breakdown']
        assert bench.count_steps(bench.parse(), 2) == 0
    finally:
        sys.argv = argv
