"""Optimizer / LR-schedule / step driver with the reference's training recipe:

    optimizer        = dict(type='AdamW', lr=0.001, weight_decay=0.0001)      configs/fcaf3d/fcaf3d.py:30
    optimizer_config = dict(grad_clip=dict(max_norm=10, norm_type=2))         :31   (mmcv OptimizerHook: clip_grad_norm_)
    lr_config        = dict(policy='step', warmup=None, step=[8, 11])         :32   (mmcv StepLrUpdaterHook, gamma 0.1)
    runner           = dict(type='EpochBasedRunner', max_epochs=12)           :33

`TrainStep` is what mmcv's `EpochBasedRunner.run_iter` + `OptimizerHook.after_train_iter` do for one batch:
zero_grad -> model(return_loss=True, **batch) -> sum of the `loss*` entries (mmdet `_parse_losses`) -> backward ->
(data parallel: gradient averaging, overlapped with backward) -> clip_grad_norm_ -> optimizer.step; `epoch_end()` is the
LR hook's per-epoch update.  On the GPU the parameters, gradients and AdamW moments live in flat buffers (flat.FlatParams)
and clip + AdamW are three launches of csrc/optim.hip (`FlatAdamW`); CPU parameters (the host-logic tests) take
torch.optim.AdamW + clip_grad_norm_, the calls the reference makes.
"""
import os

import torch

from . import _lib as L
from . import dist as D
from .flat import FlatParams


class FlatAdamW:
    """torch.optim.AdamW (betas 0.9 / 0.999, eps 1e-8, decoupled weight decay, no amsgrad) + clip_grad_norm_ over the flat
    buffers of a `FlatParams`: `step(max_norm)` = fc_grad_norm (global 2-norm and clip coefficient, on the device) +
    fc_adamw_step (one pass; the coefficient is applied while the gradient is read).  One difference to torch.optim.AdamW:
    a parameter that received NO gradient in a step has its slice zeroed by `FlatParams.gather` and is stepped with a zero
    gradient (weight decay and moment decay apply) where torch skips it — every parameter of the FCAF3D detector receives a
    gradient in every step.  Keeps the slice of torch's optimizer
    interface the runner and the LR hook use: `param_groups`, `defaults`, `zero_grad`, `state_dict` / `load_state_dict` in
    torch's own layout (per-parameter `step`, `exp_avg`, `exp_avg_sq`), so mmcv-style checkpoints round-trip."""

    def __init__(self, flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.flat = flat
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        self.param_groups = [dict(self.defaults, params=flat.params)]
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.norm_clip = torch.zeros(2, dtype=torch.float32, device=flat.data.device)      # [global grad norm, clip coefficient]
        self.steps = 0

    def zero_grad(self, set_to_none=True):
        for p in self.flat.params:
            p.grad = None                        # the flat gradient buffer is overwritten, not accumulated into
        self.flat.complete = False

    def step(self, max_norm=None, gathered=False):
        """-> the global gradient norm (0-d device tensor) when max_norm is given"""
        f = self.flat
        if not f.data.is_cuda:
            raise RuntimeError('FlatAdamW runs on the GPU only (HIP); CPU parameters take torch.optim.AdamW')
        if not gathered and not getattr(f, 'complete', False):      # complete: the executor's backward left every gradient in place
            f.gather()
        g = self.param_groups[0]
        self.steps += 1
        b1, b2 = g['betas']
        clip = None
        if max_norm is not None:
            ws = L.workspace(L.query('fc_grad_norm_ws_bytes', f.n), f.data.device)
            L.call('fc_grad_norm', L.ptr(f.grad), f.n, float(max_norm), L.ptr(self.norm_clip), L.ptr(ws), ws.numel(), L.stream())
            clip = self.norm_clip
        L.call('fc_adamw_step', L.ptr(f.data), L.ptr(f.grad), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq), f.n, float(g['lr']),
               float(b1), float(b2), float(g['eps']), float(g['weight_decay']), 1.0 - b1 ** self.steps, 1.0 - b2 ** self.steps,
               L.ptr(clip), L.stream())
        return self.norm_clip[0].clone() if clip is not None else None      # a copy: norm_clip is rewritten by the next step

    def _views(self, buf):
        return [buf[o:o + p.numel()].view(p.shape) for p, o in zip(self.flat.params, self.flat.offsets)]

    def state_dict(self):
        m, v = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        state = {i: dict(step=torch.tensor(float(self.steps)), exp_avg=m[i].clone(), exp_avg_sq=v[i].clone())
                 for i in range(len(m))} if self.steps else {}
        group = {k: val for k, val in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(self.flat.params)))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        m, v = self._views(self.exp_avg), self._views(self.exp_avg_sq)
        steps = 0
        with torch.no_grad():
            self.exp_avg.zero_()                  # entries the loaded state does not hold start from zero moments, as a
            self.exp_avg_sq.zero_()               # fresh torch.optim.AdamW state would
            for i, st in sd['state'].items():
                m[int(i)].copy_(st['exp_avg'])
                v[int(i)].copy_(st['exp_avg_sq'])
                steps = int(st['step'])
        self.steps = steps
        for k, val in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = tuple(val) if k == 'betas' else val


def build_optimizer(model, cfg, flat=None):
    """cfg: the `optimizer` dict of a config (type AdamW / Adam / SGD, mmcv's constructor keys).  `flat`: the model's
    FlatParams (GPU): AdamW then runs on the flat buffers (FlatAdamW)."""
    cfg = dict(cfg)
    kind = cfg.pop('type')
    params = [p for p in model.parameters() if p.requires_grad]
    fused = all(p.is_cuda for p in params)
    if kind == 'AdamW' and flat is not None:
        return FlatAdamW(flat, **cfg)
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
    """mmdet BaseDetector._parse_losses: every entry is reduced to a scalar (a tensor by its mean, a list of tensors by
    the sum of their means) and the training loss is the sum of the entries whose key contains 'loss'."""
    def mean(t):                     # the mean of a 0-dim tensor is the tensor: no reduction launch, none in its backward
        return t if t.dim() == 0 else t.mean()
    total = None
    for k, v in losses.items():
        if 'loss' in k:
            v = mean(v) if torch.is_tensor(v) else sum(mean(x) for x in v)
            total = v if total is None else total + v
    return total


def reserve_device_memory(gigabytes, device=None, streams=None, chunk_gb=8):
    """Take `gigabytes` of device memory into torch's caching allocator up front (as chunk_gb blocks, released to the cache at once:
    later requests are carved out of them).  A training step asks the allocator for its arenas (the coordinate phase's tables, the
    executor's activations and workspaces) every step; which cached block fits depends on how far the host runs ahead of the GPU, and a
    step that finds none goes to the driver — a device allocation of a few hundred MB takes 20-120 ms on this stack, inside the step
    (profiles/r6_notes.md section 14: every "slow first process" of r5 / r6 was one to five such calls inside the 20 timed steps).  On a
    288 GB device holding back a few tens of GB removes them.
    The allocator keeps one pool PER STREAM: `streams` = [(stream, share)] splits the reserve over the streams the step allocates under
    (default: the current stream 1/2, the coordinate stream 1/4, the weight-gradient stream 1/4).  Returns the bytes reserved (at most
    a quarter of what is free)."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    if streams is None:
        from . import functional as Fn
        from . import sparse as SP
        streams = [(torch.cuda.current_stream(dev), 0.5), (SP.map_stream(dev), 0.25), (Fn.wgrad_stream(dev), 0.25)]
    free, _ = torch.cuda.mem_get_info(dev)
    budget = min(float(gigabytes), free / 4 / 2 ** 30)
    tot = sum(w for _, w in streams) or 1.0
    got = 0
    for st, w in streams:
        n = int(budget * w / tot // chunk_gb)
        with torch.cuda.stream(st):
            blocks = []
            try:
                for _ in range(max(n, 0)):
                    blocks.append(torch.empty(chunk_gb << 30, dtype=torch.uint8, device=dev))
            except torch.OutOfMemoryError:         # somebody else took the memory meanwhile (several processes on one device): keep what there is
                pass
            got += sum(b.numel() for b in blocks)
            del blocks
    return got


class TrainStep:
    """One optimisation step of the reference's recipe on this process's GPU (one process per GPU; gradients are averaged
    over the process group, if any, by fcaf3d_amd.dist.GradientAverager while backward is still running)."""

    def __init__(self, model, optimizer_cfg, optimizer_config=None, lr_config=None, bucket_mb=64, flat=None):
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
        clip = (optimizer_config or {}).get('grad_clip')
        self.max_norm = clip['max_norm'] if clip else None
        self.norm_type = clip.get('norm_type', 2) if clip else 2
        # GPU: parameters and gradients move into flat buffers (flat.FlatParams re-points every p.data; call after
        # model.to(device) and do not move the model afterwards); `flat=False` keeps the per-tensor torch path
        on_gpu = all(p.is_cuda for p in self.params)
        use_flat = (on_gpu and optimizer_cfg.get('type') == 'AdamW' and self.norm_type == 2) if flat is None else flat
        self.flat = FlatParams(self.params) if use_flat else None
        self.optimizer = build_optimizer(model, optimizer_cfg, flat=self.flat)
        self.lr = build_lr_updater(self.optimizer, lr_config) if lr_config else None
        self.averager = D.GradientAverager(self.params, bucket_mb=bucket_mb, flat=self.flat)
        self.last_grad_norm = None
        # host seconds per phase, accumulated over the calls: [run-ahead bound + prefetch submit, forward_train enqueue, backward enqueue,
        # averager + clip + AdamW + weight images] (bench.py config.host.phases_ms: which phase grows when the host is slow)
        self.phase_s = [0.0, 0.0, 0.0, 0.0]
        # split-bf16 convolutions read pre-split weight images: with the flat optimizer this loop is the only writer of the
        # weights, so it builds all of them in one launch right after its step (on the weight-gradient stream, under the
        # next step's voxelisation and stem) instead of ~100 per-layer launches on the critical path of the next step.
        # Whoever else writes the parameters (load_state_dict, ...) does so through torch and moves their version counters
        # (the flat optimizer writes through raw pointers and does not); then the images are rebuilt before use.
        from . import functional as Fn
        self.images = None
        if self.flat is not None and Fn.X6 and os.environ.get('FC_X6_PREBUILD', '1') != '0':
            from .nn import MinkowskiConvolution
            ws = [m.kernel for m in model.modules() if isinstance(m, MinkowskiConvolution) and m.kernel.requires_grad]
            self.images = Fn.WeightImages(ws)
            self._image_ws, self.images_version = ws, None

    @classmethod
    def from_config(cls, model, cfg, **kw):
        return cls(model, cfg.optimizer, cfg.get('optimizer_config'), cfg.get('lr_config'), **kw)

    def invalidate_images(self):
        """Call after writing the weights in a way that does not move their version counters (`p.data.copy_`,
        `dist.broadcast(p.data)`, EMA / weight surgery through `.data`): the next step rebuilds the pre-split weight images
        (the module path's and the native executor's) before it uses them.  Writes through torch ops on the parameters
        themselves (load_state_dict, `with no_grad(): p.copy_()`) are detected by the version counters."""
        for pr in getattr(self.model, '_programs', {}).values():
            pr.weights_fresh = False
        self.images_version = None

    def _build_images(self, side_stream):
        from . import functional as Fn
        if side_stream:
            side, main = Fn.wgrad_stream(self.flat.data.device), torch.cuda.current_stream()
            side.wait_stream(main)                   # the optimizer step (and every reader of the old images) is enqueued on main
            with torch.cuda.stream(side):
                self.images.build()
        else:
            self.images.build()
        self.images_version = sum(w._version for w in self._image_ws)

    def _program(self):
        """the model's native-executor program for a training step, if this step will go through it (executor.py)"""
        from . import executor
        if self.flat is None or not hasattr(self.model, 'backbone') or not self.model.training:
            return None
        return executor.program_for(self.model, True)

    def __call__(self, batch, next_batch=None):
        """next_batch: the batch of the FOLLOWING call, if the caller has it (a data loader's prefetched item): its coordinate
        phase starts now on a worker thread (SingleStageSparse3DDetector.prefetch) and overlaps this step"""
        from . import functional as Fn
        import time
        ph = self.phase_s
        t0 = time.perf_counter()
        self._bound_run_ahead()
        if next_batch is not None and hasattr(self.model, 'prefetch'):
            self.model.prefetch(next_batch['points'], gt=True)
        t1 = time.perf_counter()
        ph[0] += t1 - t0
        self.optimizer.zero_grad(set_to_none=True)
        Fn._flat_pass_done()                         # a backward pass that raised never ran its final callback: start clean
        prog = self._program()
        if prog is not None:
            # the executor reads its own weight images (incl. the packed head kernel and the generative kernels' GEMM form); they
            # are refreshed right after the optimizer step below, or here if something else has written the weights since
            ver = sum(w._version for w in self._image_ws) if self.images is not None else None
            if not prog.weights_fresh or getattr(self, '_prog_version', None) != ver:
                prog.refresh_weights()
                self._prog_version = ver
        elif self.images is not None and self.images.n:
            if self.images_version != sum(w._version for w in self._image_ws):      # first step, or somebody wrote the weights through torch
                self._build_images(side_stream=False)
            Fn.PREBUILT, Fn.PREBUILT_EVENT = self.images.table, self.images.event
        from . import executor
        executor.TRUSTED = prog                      # this step's images were rebuilt after the last write of the weights
        try:
            losses = self.model(return_loss=True, **batch)
            loss = parse_losses(losses)
            t2 = time.perf_counter()
            ph[1] += t2 - t1
            loss.backward()
            t3 = time.perf_counter()
            ph[2] += t3 - t2
        finally:
            Fn.PREBUILT, Fn.PREBUILT_EVENT = {}, None
            executor.TRUSTED = None
        self.averager.finish()
        if isinstance(self.optimizer, FlatAdamW):
            # after finish() under data parallelism every gradient already sits (averaged) in the flat buffer
            self.last_grad_norm = self.optimizer.step(self.max_norm, gathered=bool(self.averager.buckets))
            for pr in getattr(self.model, '_programs', {}).values():
                pr.weights_fresh = False
            if prog is not None:
                side, main = Fn.wgrad_stream(self.flat.data.device), torch.cuda.current_stream()
                side.wait_stream(main)                   # the optimizer step (and every reader of the old images) is enqueued on main
                with torch.cuda.stream(side):
                    prog.refresh_weights()
                self.images_version = None               # the module path's images are stale now; rebuilt if a step takes that path
            elif self.images is not None and self.images.n:
                self._build_images(side_stream=True)
            ph[3] += time.perf_counter() - t3
            return loss, losses
        if self.max_norm is not None:
            self.last_grad_norm = torch.nn.utils.clip_grad_norm_(self.params, self.max_norm, norm_type=self.norm_type)
        self.optimizer.step()
        return loss, losses

    def _bound_run_ahead(self, depth=int(os.environ.get('FC_RUN_AHEAD', '2'))):
        """the host may enqueue at most `depth` steps ahead of the GPU: with the coordinate phase off the main thread nothing else
        in a step waits for the device, and every step in flight holds its arenas"""
        if not self.params or not self.params[0].is_cuda:
            return
        evs = self.__dict__.setdefault('_step_events', [])
        if len(evs) >= depth:
            e0 = evs.pop(0)
            if not e0.query():
                import time
                t0 = time.perf_counter()
                e0.synchronize()
                L.HOST_WAIT[0] += time.perf_counter() - t0
        e = torch.cuda.Event()
        e.record()
        evs.append(e)

    def epoch_end(self):
        if self.lr is not None:
            self.lr.epoch_end()

    def state_dict(self):
        return dict(optimizer=self.optimizer.state_dict(), epoch=self.lr.epoch if self.lr else 0)

    def load_state_dict(self, sd):
        self.invalidate_images()
        self.optimizer.load_state_dict(sd['optimizer'])
        if self.lr is not None:
            self.lr.set_epoch(sd.get('epoch', 0))
