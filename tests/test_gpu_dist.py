"""Model-level data parallelism on ONE GPU: two processes (gloo; RCCL needs one GPU per rank) run the real training step
(fcaf3d_amd.runner.TrainStep: forward_train + backward with bucketed gradient averaging + clip + AdamW) on different
scenes.  Checks what DDP must guarantee: bitwise-identical parameters on both ranks after every step, and agreement of the
rank-averaged loss with a single-process run over the global batch (the only designed difference is the reference's
`reduce_mean` of the per-scene loss normalisers across ranks, fcaf3d_neck_with_head.py:179,187)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _collect(q, procs, n, timeout):
    """n results from the workers' queue — failing at once when a worker has died (r5: a failed assertion in one rank left the
    other waiting in a collective and the test in q.get for its whole 15-minute timeout)"""
    import queue as _queue
    import time
    out, t0 = [], time.time()
    while len(out) < n:
        try:
            out.append(q.get(timeout=5))
        except _queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError(f'workers failed (exit codes {dead}) or timed out after {time.time() - t0:.0f} s')
    return sorted(out)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(fa, seeds, dev):
    from fcaf3d_amd.synthetic import make_scene
    sc = [make_scene(s, n_points=6000) for s in seeds]
    return dict(points=[torch.from_numpy(s[0]).to(dev) for s in sc],
                gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(s[1]), origin=(.5, .5, .5)) for s in sc],
                gt_labels_3d=[torch.from_numpy(s[2]).to(dev) for s in sc],
                img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes) for _ in sc])


def _model(fa):
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    m = cfg.model
    m.backbone['n_outs'] = 2
    m.neck_with_head['in_channels'] = (64, 128)
    m.neck_with_head.assigner['n_scales'] = 2
    return fa.build_detector(m, train_cfg=m.get('train_cfg'), test_cfg=m.get('test_cfg')), cfg


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      FC_DIST_BACKEND='gloo')
    import fcaf3d_amd as fa
    from fcaf3d_amd import dist as D
    from fcaf3d_amd.runner import TrainStep
    D.init_dist(backend='gloo')
    dev = torch.device('cuda:0')
    model, cfg = _model(fa)
    model = model.to(dev).train()
    for p in model.parameters():
        torch.distributed.broadcast(p.data, 0)
    tr = TrainStep.from_config(model, cfg, bucket_mb=8)
    assert len(tr.averager.buckets) >= 2
    losses = []
    for step in range(2):
        loss, _ = tr(_batch(fa, [100 + 10 * step + 2 * rank, 101 + 10 * step + 2 * rank], dev))
        losses.append(float(loss))
    torch.cuda.synchronize()
    digest = [float(p.detach().double().sum()) for p in model.parameters()]
    first = next(model.parameters()).detach().cpu().numpy().ravel()[:64].tobytes()
    q.put((rank, losses, digest, first))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_training_keeps_parameters_identical_and_matches_global_batch():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(q, procs, 2, 600)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, l0, d0, f0), (_, l1, d1, f1) = res
    assert d0 == d1 and f0 == f1, 'parameters diverged between the ranks'
    assert all(np.isfinite(l0 + l1))
    # single process, global batch of the first step (same initial weights)
    import fcaf3d_amd as fa
    from fcaf3d_amd.runner import parse_losses
    dev = torch.device('cuda:0')
    model, _ = _model(fa)
    model = model.to(dev).train()
    glob = float(parse_losses(model(return_loss=True, **_batch(fa, [100, 101, 102, 103], dev))))
    dp = 0.5 * (l0[0] + l1[0])
    # two 2-scene ranks against one 4-scene process: the designed difference is the per-rank BatchNorm statistics (2 small scenes
    # instead of 4 in every BatchNorm); r5: bound tightened from 5 % to 1e-3 (measured 1.4e-4, printed)
    print(f'two ranks x 2 scenes vs one process x 4 scenes: loss {dp:.6f} vs {glob:.6f}, relative difference {abs(dp - glob) / abs(glob):.2e}')
    assert abs(dp - glob) <= 1e-3 * abs(glob), (dp, glob)


def _worker_cfg4(rank, world, port, q, bucket_mb=8):
    """BASELINE config 4 as one of its GPUs sees it: fcaf3d_scannet-3d-18class (4 levels), 2 scenes of 100 000 points per rank."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0',
                      FC_DIST_BACKEND='gloo')
    import fcaf3d_amd as fa
    import fcaf3d_amd.functional as Fn
    from fcaf3d_amd import dist as D
    from fcaf3d_amd.runner import TrainStep
    from fcaf3d_amd.synthetic import make_scene
    D.init_dist(backend='gloo')
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    model = fa.build_detector(cfg.model, train_cfg=cfg.model.get('train_cfg'), test_cfg=cfg.model.get('test_cfg')).to(dev).train()
    model.async_maps = True
    Fn.WGRAD_ASYNC = True                                   # the bench's stream configuration
    for p in model.parameters():
        torch.distributed.broadcast(p.data, 0)
    # 8 MB buckets (r5): the first bucket holds nothing but head-stream gradients (scales, out_block norms, the packed head
    # kernels) — a bucket may only leave once the MAIN stream has joined the head branch that wrote it (executor._pready); with the
    # default 64 MB the first bucket reached into layer4 and could not show the ordering
    tr = TrainStep.from_config(model, cfg, bucket_mb=bucket_mb)
    assert len(tr.averager.buckets) >= (8 if bucket_mb <= 8 else 1), len(tr.averager.buckets)
    sc = [make_scene(200 + 2 * rank + i, n_points=100000) for i in range(2)]
    batch = dict(points=[torch.from_numpy(s[0]).to(dev) for s in sc],
                 gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(s[1]), origin=(.5, .5, .5)) for s in sc],
                 gt_labels_3d=[torch.from_numpy(s[2]).to(dev) for s in sc],
                 img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes) for _ in sc])
    losses = [float(tr(batch)[0]) for _ in range(2)]
    torch.cuda.synchronize()
    digest = [float(p.detach().double().sum()) for p in model.parameters()]
    q.put((rank, losses, digest))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_config4_per_gpu_shape_two_ranks():
    """BASELINE config 4 (global batch 16 over 8 GPUs = 2 ScanNet-shaped 100k-point scenes per GPU) at its PER-GPU shape on two
    ranks (gloo, one GPU): the full-size 4-level model through the native executor, weight gradients and head branch on their
    streams, bucketed gradient averaging.  Parameters must stay identical on both ranks; the rank-averaged loss agrees with a
    single process over the 4 scenes up to what the per-rank BatchNorm statistics cost (2 scenes instead of 4 in every
    BatchNorm of the step: measured 1.1e-4 relative at the first step; bound 1e-3)."""
    ctx = mp.get_context('spawn')
    runs = {}
    for bucket_mb in (8, 4096):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker_cfg4, args=(r, 2, port, q, bucket_mb)) for r in range(2)]
        for p in procs:
            p.start()
        res = _collect(q, procs, 2, 900)
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        runs[bucket_mb] = res
    # ONE bucket (launched when backward has ended) cannot leave early: the many-bucket run must end with the very same parameters —
    # a bucket reduced before a gradient of it was final (r4 ADVICE: head-stream gradients) would show here
    assert runs[8][0][2] == runs[4096][0][2], 'bucketed run differs from the single-bucket run: a bucket left before its gradients were final'
    (_, l0, d0), (_, l1, d1) = runs[8]
    assert d0 == d1, 'parameters diverged between the ranks'
    import fcaf3d_amd as fa
    from fcaf3d_amd.runner import parse_losses
    from fcaf3d_amd.synthetic import make_scene
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    cfg = fa.get_config('fcaf3d_scannet-3d-18class', voxel_size=0.02)
    model = fa.build_detector(cfg.model, train_cfg=cfg.model.get('train_cfg'), test_cfg=cfg.model.get('test_cfg')).to(dev).train()
    sc = [make_scene(200 + i, n_points=100000) for i in range(4)]
    batch = dict(points=[torch.from_numpy(s[0]).to(dev) for s in sc],
                 gt_bboxes_3d=[fa.DepthInstance3DBoxes(torch.from_numpy(s[1]), origin=(.5, .5, .5)) for s in sc],
                 gt_labels_3d=[torch.from_numpy(s[2]).to(dev) for s in sc],
                 img_metas=[dict(box_type_3d=fa.DepthInstance3DBoxes) for _ in sc])
    glob = float(parse_losses(model(return_loss=True, **batch)))
    dp = 0.5 * (l0[0] + l1[0])
    print(f'config-4 per-GPU shape: data-parallel loss {dp:.6f} (ranks {l0[0]:.6f} / {l1[0]:.6f}), single process over the 4 scenes '
          f'{glob:.6f}: relative difference {abs(dp - glob) / abs(glob):.2e}')
    assert abs(dp - glob) <= 1e-3 * abs(glob), (dp, glob)


@pytest.mark.parametrize('ranks,batch', [(2, 8), (4, 4), (8, 2)])
def test_bench_n_ranks_on_one_gpu_prints_the_contract_line(ranks, batch):
    """`bench.py --gpus N` as the driver launches it (torch.distributed.run, one rank per GPU) — here with all N ranks on cuda:0 and
    gloo instead of RCCL (FC_DIST_BACKEND=gloo; RCCL needs one GPU per rank): the ONE JSON line on rank 0 carries n_gpus = N, weak
    scaling, BASELINE config 4 (global batch 16 split over the ranks: 8 / 4 / 2 scenes per rank) and a non-empty bucket log; every
    rank runs the same number of steps (a rank that took one step more or fewer would hang the others in a collective: the run must
    end within the timeout, with the lookahead plan threads of all ranks alive beside the collectives).  r6:
    N = 4 and N = 8 — the shapes of the driver's first real SCALE pass — were never launched before.  RCCL itself with N > 1 is
    the driver's N = 2 / 4 / 8 pass (tools/dist_train.sh:7-9, configs/fcaf3d/fcaf3d.py:43)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FC_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(ranks), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', str(ranks), '--steps', '3', '--warmup', '2',
           '--batch', str(batch), '--no-cpu-baseline', '--infer-steps', '0', '--no-extras', '--no-fp32-route']
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == ranks and out['scaling'] == 'weak' and out['steps'] == 3 and out['value'] > 0
    assert out['config']['global_batch'] == batch * ranks and out['config']['parallelism'] == f'dp{ranks}'
    c4 = out['config']['config4_global_batch_16']
    assert c4 is not None and c4['global_batch'] == 16 and c4['scenes_per_gpu_per_step'] == 16 // ranks and c4['value'] > 0
    dp = out['config']['data_parallel']
    assert dp['buckets'] >= 2 and len(dp['last_step_launch_ms_after_first_grad']) == dp['buckets'] and dp['backend'] == 'gloo'
    assert 'worker thread' in out['config']['coordinate_phase']
    print(f'bench.py --gpus {ranks} (gloo, one GPU):', out['value'], 'scenes/s;', dp)
