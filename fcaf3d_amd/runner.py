"""Optimizer / LR-schedule / step driver with the reference's training recipe:

    optimizer        = dict(type='AdamW', lr=0.001, weight_decay=0.0001)      configs/fcaf3d/fcaf3d.py:30
    optimizer_config = dict(grad_clip=dict(max_norm=10, norm_type=2))         :31   (mmcv OptimizerHook: clip_grad_norm_)
    lr_config        = dict(policy='step', warmup=None, step=[8, 11])         :32   (mmcv StepLrUpdaterHook, gamma 0.1)
    runner           = dict(type='EpochBasedRunner', max_epochs=12)           :33

`TrainStep` is what mmcv's `EpochBasedRunner.run_iter` + `OptimizerHook.after_train_iter` do for one batch:
zero_grad -> model(return_loss=True, **batch) -> sum of the `loss*` entries (mmdet `_parse_losses`) -> backward ->
(data parallel: gradient averaging, overlapped with backward) -> clip_grad_norm_ -> optimizer.step; `epoch_end()` is the
LR hook's per-epoch update.  AdamW runs as ONE fused multi-tensor kernel (`fused=True`), the clip as torch's foreach path.
"""
import torch

from . import dist as D


def build_optimizer(model, cfg):
    """cfg: the `optimizer` dict of a config (type AdamW / Adam / SGD, mmcv's constructor keys)."""
    cfg = dict(cfg)
    kind = cfg.pop('type')
    params = [p for p in model.parameters() if p.requires_grad]
    fused = all(p.is_cuda for p in params)
    if kind == 'AdamW':
        return torch.optim.AdamW(params, fused=fused, **cfg)
    if kind == 'Adam':
        return torch.optim.Adam(params, fused=fused, **cfg)
    if kind == 'SGD':
        return torch.optim.SGD(params, **cfg)
    raise KeyError(f'optimizer type {kind!r} is not used by any FCAF3D config')


class StepLrUpdater:
    """mmcv StepLrUpdaterHook with by_epoch=True: lr = base_lr * gamma ** (number of `step` entries <= epoch)."""

    def __init__(self, optimizer, step, gamma=0.1, min_lr=None, warmup=None, **_ignored):
        assert warmup is None, 'the FCAF3D configs train without warm-up'
        self.optimizer = optimizer
        self.steps = [step] if isinstance(step, int) else sorted(step)
        self.gamma, self.min_lr = gamma, min_lr
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]
        self.epoch = 0

    def lr_at(self, base_lr, epoch):
        exp = sum(1 for s in self.steps if epoch >= s)
        lr = base_lr * self.gamma ** exp
        return max(lr, self.min_lr) if self.min_lr is not None else lr

    def set_epoch(self, epoch):
        self.epoch = epoch
        for g, base in zip(self.optimizer.param_groups, self.base_lrs):
            g['lr'] = self.lr_at(base, epoch)

    def epoch_end(self):
        self.set_epoch(self.epoch + 1)


def build_lr_updater(optimizer, cfg):
    cfg = dict(cfg)
    policy = cfg.pop('policy')
    if policy != 'step':
        raise KeyError(f'lr policy {policy!r} is not used by any FCAF3D config')
    return StepLrUpdater(optimizer, **cfg)


def parse_losses(losses):
    """mmdet BaseDetector._parse_losses: the training loss is the sum of every entry whose key contains 'loss'."""
    total = None
    for k, v in losses.items():
        if 'loss' in k:
            v = v if torch.is_tensor(v) else sum(v)
            total = v if total is None else total + v
    return total


class TrainStep:
    """One optimisation step of the reference's recipe on this process's GPU (one process per GPU; gradients are averaged
    over the process group, if any, by fcaf3d_amd.dist.GradientAverager while backward is still running)."""

    def __init__(self, model, optimizer_cfg, optimizer_config=None, lr_config=None, bucket_mb=64):
        self.model = model
        self.optimizer = build_optimizer(model, optimizer_cfg)
        clip = (optimizer_config or {}).get('grad_clip')
        self.max_norm = clip['max_norm'] if clip else None
        self.norm_type = clip.get('norm_type', 2) if clip else 2
        self.lr = build_lr_updater(self.optimizer, lr_config) if lr_config else None
        self.averager = D.GradientAverager(model.parameters(), bucket_mb=bucket_mb)
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.last_grad_norm = None

    @classmethod
    def from_config(cls, model, cfg, **kw):
        return cls(model, cfg.optimizer, cfg.get('optimizer_config'), cfg.get('lr_config'), **kw)

    def __call__(self, batch):
        self.optimizer.zero_grad(set_to_none=True)
        losses = self.model(return_loss=True, **batch)
        loss = parse_losses(losses)
        loss.backward()
        self.averager.finish()
        if self.max_norm is not None:
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm, norm_type=self.norm_type)
        self.optimizer.step()
        return loss, losses

    def epoch_end(self):
        if self.lr is not None:
            self.lr.epoch_end()

    def state_dict(self):
        return dict(optimizer=self.optimizer.state_dict(), epoch=self.lr.epoch if self.lr else 0)

    def load_state_dict(self, sd):
        self.optimizer.load_state_dict(sd['optimizer'])
        if self.lr is not None:
            self.lr.set_epoch(sd.get('epoch', 0))
