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
