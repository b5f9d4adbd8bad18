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


# bench.py --gpus 1: run the gradient averager (hooks, buckets, RCCL launches) in a 1-rank process group, so that the
# overhead of the data-parallel path itself is measured without a second GPU
FORCE_AVERAGER = False
ACTIVE = None          # the GradientAverager that is averaging right now (the native executor hands it its buckets: executor.py)


def _averaging():
    return world_size() > 1 or (FORCE_AVERAGER and is_dist())


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
    """Bucketed gradient all-reduce (sum / world) for pure data parallelism, IN PLACE on the flat gradient buffer.

    The gradients of all parameters live in one flat fp32 buffer (`flat.FlatParams`; parameters in registration order).  A
    bucket is a run of consecutive parameters, i.e. one contiguous range of that buffer of about `bucket_mb`; buckets are
    formed from the LAST parameter backwards (≈ the order backward produces gradients) and each is all-reduced with one
    async RCCL call as soon as all of its gradients have been accumulated (autograd post-accumulate hooks), so the
    collectives overlap the rest of backward.  No copy-in / copy-out: the weight-gradient kernels write into the buffer,
    the collective reduces it where it lies, the optimizer reads it there (r2 copied 2 x 282 MB per step).  xGMI is
    point-to-point (ring all-reduce is per-link bound), hence few, large buckets.

    Streams: with the weight gradients on their own HIP stream (functional.WGRAD_ASYNC) the collective is issued with THAT
    stream current, after it has been made to wait for the main stream's position — RCCL's stream then waits for exactly
    the kernels that produce the bucket, and the main stream (backward-data chain) never waits for a weight gradient
    (r2 joined the two streams at every bucket)."""

    def __init__(self, params, bucket_mb=64, flat=None):
        from .flat import FlatParams, flat_of
        self.params = [p for p in params if p.requires_grad]
        self.buckets = []
        self._handles = []
        self.flat = flat
        self._hooks = []
        if not _averaging():
            return
        if self.flat is None:
            self.flat = flat_of(self.params[0]) or FlatParams(self.params)
        assert all(flat_of(p) is self.flat for p in self.params), 'all parameters must live in one FlatParams'
        order = sorted(self.params, key=lambda p: p._fc_flat[1])
        cur, cur_bytes = [], 0
        for p in reversed(order):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= bucket_mb * (1 << 20):
                self.buckets.append(cur[::-1])
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur[::-1])
        self._ranges = [self.flat.range_of(b) for b in self.buckets]           # contiguous [begin, end) per bucket
        for (lo, hi), b in zip(self._ranges, self.buckets):
            assert hi - lo == sum(-(-p.numel() // 64) * 64 for p in b), 'a bucket must be a contiguous run of parameters'
        global ACTIVE
        ACTIVE = self
        self._avg = dist.get_backend() == 'nccl'   # RCCL averages in the collective; gloo: divide, then sum
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[p] = bi
                self._hooks.append(p.register_post_accumulate_grad_hook(self._hook))
        self._reset()

    def close(self):
        """detach the autograd hooks (a second averager may then take over the same parameters)"""
        global ACTIVE
        if ACTIVE is self:
            ACTIVE = None
        for h in self._hooks:
            h.remove()
        self._hooks, self.buckets = [], []

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._handles = []
        self._next = 0                         # buckets are reduced strictly in index order (see _hook)
        self._t0 = None
        self._streams = {}                     # raw stream -> torch Stream: where this step's gradients were accumulated

    def _hook(self, p):
        if self.log is not None and self._t0 is None:
            import time
            self._t0 = time.perf_counter()     # first gradient of the step: backward is under way
        if p.is_cuda:
            # the stream this gradient was accumulated on (autograd runs a hook under the AccumulateGrad node's stream: the
            # stream of the parameter's forward use — the main stream, or the neck's head-branch stream): _launch orders the
            # bucket's gather copy and its collective behind every such stream
            from . import _lib as L
            raw = L.stream()
            if raw not in self._streams:
                self._streams[raw] = torch.cuda.current_stream(p.device)
        # Collectives must be issued in the SAME order on every rank.  A bucket becomes ready when its last gradient
        # arrives, which — if some parameter gets a gradient on one rank only — need not happen in the same order
        # everywhere; so a ready bucket is launched only once every bucket before it has been (torch DDP's rule), and
        # finish() flushes the rest in order.
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        while self._next < len(self.buckets) and self._pending[self._next] == 0:
            self._launch(self._next)
            self._next += 1

    def launch_next(self):
        """the native executor's backward (executor.py) has enqueued every operator that writes the next bucket's gradients:
        launch its all-reduce now (the per-operator path gets here through the autograd hooks)"""
        if self._next < len(self.buckets):
            if self.log is not None and self._t0 is None:
                import time
                self._t0 = time.perf_counter()
            if self.flat.grad.is_cuda:
                from . import _lib as L
                raw = L.stream()
                if raw not in self._streams:
                    self._streams[raw] = torch.cuda.current_stream(self.flat.grad.device)
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
        # gradients autograd produced outside the flat buffer (small tensors: norms, head) are copied into their slices,
        # parameters without a gradient contribute zeros — one multi-tensor launch each, on the current stream.  The
        # bucket's gradients may have been accumulated on SEVERAL streams (head branch of the neck on its own stream, the
        # rest on the main stream): every hook of the bucket has fired, so all producing kernels are enqueued — the current
        # stream waits for each of those streams before it reads them.
        if self.flat.grad.is_cuda:
            cur = torch.cuda.current_stream(self.flat.grad.device)
            for s in self._streams.values():
                if s != cur:
                    cur.wait_stream(s)
        self.flat.gather(self.buckets[bi])
        lo, hi = self._ranges[bi]
        buf = self.flat.grad[lo:hi]
        side = None
        if buf.is_cuda and Fn.WGRAD_ASYNC:
            main = torch.cuda.current_stream(buf.device)
            side = Fn.wgrad_stream(buf.device)
            side.wait_stream(main)             # the copies above, and every gradient the main stream itself produced
        with (torch.cuda.stream(side) if side is not None else _nullctx()):
            if self._avg:
                h = dist.all_reduce(buf, op=dist.ReduceOp.AVG, async_op=True)
            else:
                buf.div_(world_size())
                h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        self._handles.append((bi, h))

    def finish(self):
        """Call after backward: flush the buckets whose hooks did not all fire, wait, and make every `p.grad` the
        (averaged) slice of the flat buffer."""
        if not self.buckets:
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
        for p in self.params:
            v = self.flat.grad_view(p)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v
        self._reset()


class _nullctx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


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
    return dict(buckets=len(averager.buckets), bucket_MB=[round((hi - lo) * 4 / 2 ** 20, 1) for lo, hi in averager._ranges],
                last_step_launch_ms_after_first_grad=[round(t * 1e3, 2) for k, b, t in last if k == 'launch'],
                last_step_backward_end_ms=round([t for k, b, t in last if k == 'finish_enter'][0] * 1e3, 2),
                backend=dist.get_backend())
