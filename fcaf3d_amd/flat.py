"""Flat parameter / gradient storage: every trainable tensor of the model is a view into ONE fp32 buffer, every gradient a
view into a second buffer of the same layout (r3).  With 288 GB of HBM nothing argues for 161 separate allocations, and
three things fall out of one layout:

  * the weight-gradient kernels write straight into their parameter's slice of the gradient buffer
    (`functional._SparseConv.backward`) and autograd adopts that view as `.grad` — no copy;
  * data parallelism all-reduces contiguous RANGES of the gradient buffer in place (`dist.GradientAverager`) — what torch
    DDP does with its bucket views, without a copy-in / copy-out of the 282 MB of gradients;
  * the optimizer is two launches over flat buffers (`csrc/optim.hip`: global gradient norm + clip coefficient, fused
    AdamW) instead of ~14 multi-tensor launches over 161 tensors.

The reference gets the same semantics from torch.optim.AdamW + clip_grad_norm_ + MMDistributedDataParallel
(configs/fcaf3d/fcaf3d.py:30-31, tools/train.py:128-135)."""
import torch

ALIGN = 64          # floats: every tensor starts on a 256-byte boundary (16-byte vector accesses, no shared cache lines)
GENERATION = [0]    # bumped whenever parameters are re-pointed (here; nn.Module._apply of the detector): address tables keyed on it


class FlatParams:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, 'no trainable parameters'
        dev = self.params[0].device
        assert all(p.device == dev and p.dtype == torch.float32 for p in self.params), 'one device, fp32'
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += -(-p.numel() // ALIGN) * ALIGN
        self.n = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)       # padding stays zero: norms over the buffer are exact
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.data[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v                                                  # the Parameter object (and its name) is unchanged
                p._fc_flat = (self, o)
        GENERATION[0] += 1

    def grad_view(self, p):
        """a FRESH view of p's gradient slice (autograd adopts a gradient only if nobody else references the tensor object)"""
        o = p._fc_flat[1]
        return self.grad[o:o + p.numel()].view(p.shape)

    def range_of(self, params):
        """[begin, end) of the flat buffers covered by a run of consecutive parameters"""
        o0 = params[0]._fc_flat[1]
        last = params[-1]
        o1 = last._fc_flat[1] + -(-last.numel() // ALIGN) * ALIGN
        return o0, o1

    def gather(self, params=None, adopt=False):
        """After backward: make every parameter's slice of `self.grad` hold its gradient — copy the gradients autograd
        produced elsewhere (one multi-tensor launch), zero the slices of parameters that received none.
        adopt=True: afterwards `p.grad` IS the slice (what the all-reduce and the optimizer see)."""
        src, dst, zero = [], [], []
        for p in (self.params if params is None else params):
            v = self.grad_view(p)
            if p.grad is None:
                zero.append(v)
            elif p.grad.data_ptr() != v.data_ptr():
                src.append(p.grad)
                dst.append(v)
            if adopt:
                p.grad = v
        if dst:
            torch._foreach_copy_(dst, src)
        if zero:
            torch._foreach_zero_(zero)


def flat_of(p):
    f = getattr(p, '_fc_flat', None)
    return f[0] if f is not None else None
