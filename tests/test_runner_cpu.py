"""CPU: the optimizer / LR-schedule / step driver (fcaf3d_amd/runner.py) against the reference's recipe
(configs/fcaf3d/fcaf3d.py:30-33) written out by hand, and the C/OpenMP convolution oracle against the numpy/torch oracle."""
import numpy as np
import torch

import fcaf3d_amd as fa
from fcaf3d_amd import runner as R


class _Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(6, 8)
        self.b = torch.nn.Linear(8, 3)

    def forward(self, return_loss=True, x=None, y=None):
        out = self.b(torch.tanh(self.a(x)))
        return dict(loss_bbox=((out - y) ** 2).mean() * 50.0, loss_cls=out.abs().mean(), acc=out.detach().mean())


def test_step_lr_matches_mmcv_step_policy():
    cfg = fa.get_config('fcaf3d_scannet-3d-18class')
    assert cfg.lr_config == dict(policy='step', warmup=None, step=[8, 11]) and cfg.runner['max_epochs'] == 12
    tr = R.TrainStep.from_config(_Toy(), cfg)
    lrs = []
    for _ in range(12):
        lrs.append(tr.optimizer.param_groups[0]['lr'])
        tr.epoch_end()
    assert np.allclose(lrs, [1e-3] * 8 + [1e-4] * 3 + [1e-5])
    assert tr.optimizer.defaults['weight_decay'] == 1e-4 and tr.max_norm == 10 and tr.norm_type == 2


def test_train_step_equals_the_recipe_written_out():
    cfg = fa.get_config('fcaf3d_scannet-3d-18class')
    m1, m2 = _Toy(), _Toy()
    tr = R.TrainStep.from_config(m1, cfg)
    opt = torch.optim.AdamW(m2.parameters(), lr=1e-3, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1)
    for step in range(4):
        x, y = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)
        loss, losses = tr(dict(x=x, y=y))
        opt.zero_grad()
        out = m2(x=x, y=y)
        ref = out['loss_bbox'] + out['loss_cls']               # `acc` carries no 'loss' in its key: not part of the objective
        ref.backward()
        norm = torch.nn.utils.clip_grad_norm_(m2.parameters(), 10, norm_type=2)
        opt.step()
        assert torch.allclose(loss, ref) and torch.allclose(tr.last_grad_norm, norm)
        assert float(norm) > 10.0 or step > 0                   # the clip really bites on the first step of this toy
    for p, q in zip(m1.parameters(), m2.parameters()):
        assert torch.allclose(p, q, atol=1e-7)
    sd = tr.state_dict()
    tr.load_state_dict(sd)


def _dp_worker(rank, world, port, q):
    import os
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from fcaf3d_amd import dist as D
    D.init_dist(backend='gloo')
    cfg = fa.get_config('fcaf3d_scannet-3d-18class')
    model = _Toy()
    tr = R.TrainStep.from_config(model, cfg, bucket_mb=1e-4)          # several tiny buckets: hooks + ordered launches + flush
    assert len(tr.averager.buckets) >= 2
    g = torch.Generator().manual_seed(7)
    for step in range(3):
        x, y = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)      # the GLOBAL batch, same on both ranks
        half = slice(rank * 8, rank * 8 + 8)
        tr(dict(x=x[half], y=y[half]))
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    torch.distributed.all_gather(gathered, flat)
    if rank == 0:
        q.put([t.numpy() for t in gathered])
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_train_step_data_parallel_world2_equals_global_batch():
    """TrainStep under a 2-rank gloo group (one process per rank, gradients averaged bucket by bucket from the autograd hooks,
    then clip + AdamW on every rank): after 3 steps both ranks hold the SAME parameters, equal to one process stepping on the
    whole batch — the losses are means over samples, so the average of the two half-batch gradients is the global gradient
    (what MMDistributedDataParallel gives the reference, tools/dist_train.sh:7-9)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(got[0], got[1])                      # identical replicas
    cfg = fa.get_config('fcaf3d_scannet-3d-18class')
    model = _Toy()
    tr = R.TrainStep.from_config(model, cfg)
    g = torch.Generator().manual_seed(7)
    for step in range(3):
        x, y = torch.randn(16, 6, generator=g), torch.randn(16, 3, generator=g)
        tr(dict(x=x, y=y))
    ref = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy()
    assert np.allclose(got[0], ref, atol=2e-6), np.abs(got[0] - ref).max()


def test_c_openmp_conv_oracle_matches_numpy_oracle():
    from oracle import conv_c, me_oracle as mo
    rng = np.random.default_rng(0)
    n_in, n_out, K, Cin, Cout = 700, 500, 27, 32, 64
    nbr = np.full((K, n_out), -1, np.int32)
    for k in range(K):                                          # a real kernel map: input rows distinct within an offset
        m = rng.random(n_out) < 0.6
        nbr[k, m] = rng.permutation(n_in)[:m.sum()]
    x = torch.randn(n_in, Cin, requires_grad=True)
    w = torch.randn(K, Cin, Cout, requires_grad=True)
    out = mo.conv(x, w, nbr)
    g = torch.randn_like(out)
    out.backward(g)

    def rel(a, b):
        return float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(conv_c.conv_fwd(x.detach().numpy(), w.detach().numpy(), nbr), out.detach().numpy()) < 1e-5
    assert rel(conv_c.conv_dgrad(g.numpy(), w.detach().numpy(), nbr, n_in), x.grad.numpy()) < 1e-5
    assert rel(conv_c.conv_wgrad(x.detach().numpy(), g.numpy(), nbr, Cin, Cout), w.grad.numpy()) < 1e-5
    assert conv_c.num_threads() >= 1
