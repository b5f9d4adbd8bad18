"""Data-parallel plumbing: one process per GPU, torch.distributed backend "nccl" (= RCCL over xGMI on
ROCm).  Mirrors what the reference gets from mmcv/mmdet: `init_dist` (tools/train.py:128-135),
`reduce_mean` (mmdet.core, used at fcaf3d_neck_with_head.py:179,187) and MMDistributedDataParallel's
gradient averaging."""
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def init_dist(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract)."""
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or is_dist():
        return
    if backend is None:
        # FC_DIST_BACKEND=gloo: multi-process smoke of the whole DP path on a box with fewer GPUs than ranks
        backend = os.environ.get('FC_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    dist.init_process_group(backend=backend)


def reduce_mean(tensor):
    """mmdet.core.reduce_mean: average over ranks (identity without a process group)."""
    if not is_dist():
        return tensor
    t = tensor.clone()
    dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t


class GradientAverager:
    """Bucketed gradient all-reduce (sum / world) for pure data parallelism.

    Parameters are packed, in reverse registration order (≈ the order backward produces them), into
    flat fp32 buckets of `bucket_mb`; each bucket is all-reduced with an async RCCL call as soon as all
    of its gradients have been accumulated (autograd post-accumulate hooks), on RCCL's own stream, so
    the collectives overlap the rest of backward.  xGMI is point-to-point (ring all-reduce is per-link
    bound), hence few, large buckets."""

    def __init__(self, params, bucket_mb=64):
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        self._handles = []
        if world_size() == 1:
            return
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_mb * (1 << 20):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self._flat = [torch.zeros(sum(p.numel() for p in b), dtype=torch.float32, device=b[0].device)
                      for b in self.buckets]
        self._views = []                       # per bucket: the flat buffer sliced into the parameters' shapes
        for flat, b in zip(self._flat, self.buckets):
            off, views = 0, []
            for p in b:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += p.numel()
            self._views.append(views)
        self._avg = dist.get_backend() == 'nccl'   # RCCL averages in the collective; gloo: sum then divide
        self._pending = [0] * len(self.buckets)
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
                p.register_post_accumulate_grad_hook(self._hook)
        self._reset()

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._handles = []
        self._next = 0                         # buckets are reduced strictly in index order (see _hook)
        self._t0 = None

    def _hook(self, p):
        if self.log is not None and self._t0 is None:
            import time
            self._t0 = time.perf_counter()     # first gradient of the step: backward is under way
        # Collectives must be issued in the SAME order on every rank.  A bucket becomes ready when its last gradient
        # arrives, which — if some parameter gets a gradient on one rank only — need not happen in the same order
        # everywhere; so a ready bucket is launched only once every bucket before it has been (torch DDP's rule), and
        # finish() flushes the rest in order.
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    log = None          # list: when set (bench.py, N > 1), (bucket, seconds since the step's first hook) per launch + waits

    def _launch(self, bi):
        from . import functional as Fn
        if self.log is not None:
            import time
            now = time.perf_counter()
            if self._t0 is None:
                self._t0 = now
            self.log.append(('launch', bi, now - self._t0))
        if Fn.WGRAD_ASYNC:
            Fn.join_wgrad_stream()            # weight gradients may still be in flight on their side stream
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.buckets[bi]]
        torch._foreach_copy_(self._views[bi], grads)          # one multi-tensor launch per bucket
        flat = self._flat[bi]
        if self._avg:
            h = dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True)
        else:
            flat.div_(world_size())
            h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        self._handles.append((bi, h))

    def finish(self):
        """Call after backward: flush buckets whose hooks did not all fire, wait, scatter back."""
        if world_size() == 1:
            return
        while self._next < len(self.buckets):          # buckets some hook never completed (unused parameters), in order
            self._launch(self._next)
            self._next += 1
        if self.log is not None and self._t0 is not None:
            import time
            t_f = time.perf_counter()
            self.log.append(('finish_enter', -1, t_f - self._t0))
        for bi, h in self._handles:
            h.wait()
            params = self.buckets[bi]
            for p in params:
                if p.grad is None:
                    p.grad = torch.empty_like(p)
            torch._foreach_copy_([p.grad for p in params], self._views[bi])
        self._reset()


def summarize_bucket_log(averager, log):
    """bench.py (N > 1): when, on the host clock and relative to the first gradient hook of a step, each gradient bucket's
    all-reduce was ENQUEUED, and when backward returned (finish() entered) — buckets enqueued well before that point are
    the ones RCCL can overlap with the rest of backward."""
    steps, cur = [], []
    for kind, bi, t in log:
        cur.append((kind, bi, t))
        if kind == 'finish_enter':
            steps.append(cur)
            cur = []
    if not steps:
        return None
    last = steps[-1]
    return dict(buckets=len(averager.buckets), bucket_MB=[round(f.numel() * 4 / 2 ** 20, 1) for f in averager._flat],
                last_step_launch_ms_after_first_grad=[round(t * 1e3, 2) for k, b, t in last if k == 'launch'],
                last_step_backward_end_ms=round([t for k, b, t in last if k == 'finish_enter'][0] * 1e3, 2),
                backend=dist.get_backend())
