"""The network body (MEResNet3D backbone + Fcaf3DNeckWithHead.forward) as a STATIC operator list walked by the native executor
(csrc/exec.hip: one C-ABI call per direction) instead of ~1 000 Python-dispatched launches per step.

r4 (profiles/r4_hostprof.txt): the training step was host-bound at every batch size (16.7 ms of Python per step at 2 scenes per
GPU, 20.3 ms at 8).  The operator sequence of `SingleStageSparse3DDetector.extract_feat` (single_stage_sparse.py:43-50;
me_resnet.py:43-50; fcaf3d_neck_with_head.py:94-108, :256-279) is fixed by the module graph; only row counts and addresses change
from step to step.  `NetProgram` walks the module graph ONCE and emits

  * forward operators  — the entry points `nn.py` / `functional.py` would call, in the same order, with the same arguments;
  * backward operators — what torch.autograd would replay over them (backward-data, weight gradients on their own stream,
    normalisation backward, gradient accumulation where a tensor has two consumers), gradients written straight into the flat
    gradient buffer (flat.FlatParams) or a buffer of the same layout;

as rows of int64 (operand layouts: csrc/exec.hip `run_op`).  Operands are indices into three host tables refreshed per step with
a few vectorised numpy operations: `addr` (device addresses: arena tensors = base + aligned prefix sum of rows x bytes per row;
parameters, buffers, weight images: static), `dims` (row counts) and `maps` (KernelMap.desc()).

Parameter gradients are delivered by `backward()` into `p.grad` (views of the gradient buffer; accumulated into if gradients
are already there); `torch.autograd.grad(loss, parameters)` does not see the parameters behind the single autograd node of the
body — use the module path (FC_EXEC=0) for that.

The module path stays the general route (and the cross-check of the tests: forward bit for bit, gradients to rounding): the
program covers BasicBlock backbones (depth 14 / 18 / 34) with the FCAF3D neck when no pruning bites (`plan_maps` succeeded), the
split-bf16 convolution route (FC_X6=1) and heads of at most 64 fused columns; anything else falls back.
"""
import os
import struct

import numpy as np
import torch

from . import _lib as L
from . import functional as Fn
from . import nn as MEnn

ENABLED = os.environ.get('FC_EXEC', '1') != '0'
# bench.py's live roofline measurement: a list -> every step bound while it is set has its convolution operators bracketed by
# HIP events inside fc_exec (csrc/exec.hip, cfg[2]) and appends {pairs per kernel map (device scalars), operators per direction}
PROBE = None
# TrainStep (runner.py) sets this to the program whose weight images it has just rebuilt, for the duration of its step: only then
# (or for a detector that declares `static_weights` in eval mode) does a forward pass trust the images of the previous call.  Anyone
# else who steps the weights (an optimizer of their own, EMA, surgery) gets fresh images at every call, as on the per-operator path.
TRUSTED = None
# tests: keep the bound state of the last forward pass on the detector (`det._last_exec = (program, state)`) so that the recorded
# activations can be inspected (NetProgram.decisions)
KEEP_STATE = False

(OP_STEM_FWD, OP_COL_STATS, OP_NORM_FWD, OP_MAXPOOL_FWD, OP_CONV, OP_BN_FWD, OP_UNION_FWD, OP_HEAD_FWD, OP_RECORD, OP_WAIT,
 OP_HEAD_BWD, OP_WGRAD, OP_BN_BWD, OP_NORM_BWD, OP_MAXPOOL_BWD, OP_STEM_WGRAD, OP_GATHER, OP_ADD, OP_SMALL_GRADS,
 OP_PERMUTE_GENT, OP_HEAD_WFIN, OP_COPY, OP_COL_SUM, OP_ROW_SUM, OP_AMAX, OP_CLEAR) = range(1, 27)
OPW, MAPW = 24, 20
ALIGN = 256
S_MAIN, S_HEAD, S_WGRAD = 0, 1, 2
EV_FORK, EV_HEAD_DONE, EV_W0, EV_W1, EV_WEND, EV_BWD0, EV_HB = 0, 8, 9, 10, 11, 12, 16       # EV_FORK + level, EV_HB + level


def _f(x):
    """a float immediate: the bit pattern of a double in an int64 word"""
    return struct.unpack('<q', struct.pack('<d', float(x)))[0]


def supported(det):
    bb, nh = det.backbone, det.neck_with_head
    if not (ENABLED and Fn.X6 and Fn.X6_CONV and Fn.X6_WGRAD and not (Fn.FLAGS & 1) and Fn.STEM_COL and Fn.DGRAD_WT):
        return False
    if getattr(bb.BLOCK, 'expansion', 1) != 1 or det.spatial_sort:
        return False
    if 1 + nh.n_reg_outs + nh.n_classes > 64 or bb.conv1[0].in_channels != 3:
        return False
    return type(nh).forward_single is _FORWARD_SINGLE and type(nh).forward is _NECK_FORWARD


_FORWARD_SINGLE = _NECK_FORWARD = None       # set by fcaf3d_neck_with_head at import: a subclass that overrides them falls back


class NetProgram:
    def __init__(self, det, training, wgrad_async, head_overlap, tail0=False):
        """tail0: the finest neck level is PRUNED this step (`pts_threshold` bites there, fcaf3d_neck_with_head.py:110-126): the
        program ends at that level's union — `x0`, exported as a fourth differentiable output — and `_prune` + out_block_0 +
        forward_single of level 0 run on the per-operator path, attached by autograd (single_stage_sparse._exec_forward)"""
        self.det = det
        self.tail0 = tail0
        self.training = training
        self.wgrad_async = wgrad_async and training
        bb, nh = det.backbone, det.neck_with_head
        self.nl = min(bb.n_outs, 4)
        self.head_overlap = head_overlap and self.nl > 1
        self.dev = next(det.parameters()).device
        self.ops_f, self.ops_b = [], []
        self.n_addr = 0
        self.static = []            # (addr index, address)
        self.arena = {'f': [], 'b': []}     # (addr index, dims index of the row count, bytes per row, extra bytes)
        self.alias = []             # (addr index, parent addr index, dims index of the row offset, bytes per row)
        self.dyn = {}               # name -> addr index
        self.dim_names = {}
        self.map_names = {}
        self.grad_refs = []         # (addr index, float offset into the gradient buffer)
        self.keep = []              # tensors that must outlive the program build (persistent buffers)
        self._params = [p for p in det.parameters() if p.requires_grad]
        self._goff, off = {}, 0
        flat = getattr(self._params[0], '_fc_flat', None)
        self.flat = flat[0] if flat is not None else None
        for p in self._params:
            if self.flat is not None:
                assert p._fc_flat[0] is self.flat
                self._goff[id(p)] = p._fc_flat[1]
            else:
                self._goff[id(p)] = off
                off += -(-p.numel() // 64) * 64
        self._gtotal = self.flat.n if self.flat is not None else off
        self._sig = self.signature(det)
        self._build_weights()
        self._build()
        self._finalise()

    @staticmethod
    def signature(det):
        """what the program's static address table depends on: cheap to compare per step (parameters are re-pointed only by
        flat.FlatParams and by nn.Module._apply, which drops the detector's programs)"""
        from .flat import GENERATION
        return GENERATION[0]

    # ---- tables ------------------------------------------------------------------------------------------------------
    def _new(self):
        self.n_addr += 1
        return self.n_addr - 1

    def D(self, name):
        if name not in self.dim_names:
            self.dim_names[name] = len(self.dim_names)
        return self.dim_names[name]

    def M(self, name):
        if name not in self.map_names:
            self.map_names[name] = len(self.map_names)
        return self.map_names[name]

    def T(self, rows, cols, arena='f', elem=4, extra=0):
        i = self._new()
        self.arena[arena].append((i, self.D(rows), cols * elem, extra))
        return i

    def S(self, tensor):
        """a static address: parameter data, module buffer, persistent scratch"""
        i = self._new()
        self.static.append((i, tensor.data_ptr()))
        self.keep.append(tensor)
        return i

    def SA(self, address):
        i = self._new()
        self.static.append((i, int(address)))
        return i

    def DY(self, name):
        if name not in self.dyn:
            self.dyn[name] = self._new()
        return self.dyn[name]

    def AL(self, parent, off_dim, cols, elem=4):
        i = self._new()
        self.alias.append((i, parent, self.D(off_dim), cols * elem))
        return i

    def G(self, p):
        i = self._new()
        self.grad_refs.append((i, self._goff[id(p)]))
        return i

    def emit(self, lst, *words):
        w = list(words) + [0] * (OPW - len(words))
        assert len(w) == OPW, len(w)
        lst.append(w)

    # ---- r6: amax words of the convolutions' operands (csrc/conv_x6.h "h3": the scale of the two-piece fp16 split) ------------
    # One fc_amax pass per operand TENSOR and stream instead of one per consuming launch: a forward activation's word serves its
    # forward convolution(s) and, a pass later, their weight gradients; a gradient's word serves the backward-data convolution and the
    # weight gradient of its layer.  Slots are 2 KB of one persistent buffer (32 sub-words, fc_common.h; fc_amax returns its scratch words to zero).
    AMAX_SLOTS = 1024
    AMAX_SLOT_BYTES = 2048           # include/fcaf3d_hip.h FC_AMAX_SLOT_BYTES: 32 sub-words at a 64-byte stride

    def amax_slot(self):
        if getattr(self, '_amax_buf', None) is None:
            self._amax_buf = torch.zeros(self.AMAX_SLOTS * self.AMAX_SLOT_BYTES // 4, dtype=torch.int32, device=self.dev)
            self._amax_used = 0
            self.keep.append(self._amax_buf)
        assert self._amax_used < self.AMAX_SLOTS
        i = self.SA(self._amax_buf.data_ptr() + self.AMAX_SLOT_BYTES * self._amax_used)
        self._amax_used += 1
        return i

    def amax_op(self, lst, stream, x, rows, cols):
        """emit the amax pass of tensor x ((rows, cols) floats) on `stream` -> address index of its word"""
        slot = self.amax_slot()
        self.emit(lst, OP_AMAX, stream, x, self.D(rows), cols, slot)
        return slot

    # ---- weights: pre-split images of every kernel, incl. the packed head kernel and the generative convolutions' GEMM form; ONE
    # set per detector, shared by its programs (training / inference, stream configurations, pruned-tail variant)
    def _build_weights(self):
        holder = self.det.__dict__.get('_exec_weights')
        if holder is None or holder.sig != self._sig:
            holder = self.det.__dict__['_exec_weights'] = _Weights(self.det, self.dev, self._sig)
        self.w = holder
        self.packed, self.gent_w, self.gents, self.images, self.ncol = holder.packed, holder.gent_w, holder.gents, holder.images, holder.ncol

    @property
    def weights_fresh(self):
        return self.w.fresh

    @weights_fresh.setter
    def weights_fresh(self, v):
        self.w.fresh = v

    def refresh_weights(self):
        self.w.refresh()

    def IMG(self, w, transposed):
        v = self.w.img_of[w.data_ptr()]
        assert v is not None and v[1 if transposed else 0] is not None, 'no split-bf16 image for this kernel shape'
        return self.S(v[1 if transposed else 0])

    # ---- program ---------------------------------------------------------------------------------------------------------
    def _build(self):
        det = self.det
        bb, nh = det.backbone, det.neck_with_head
        tr = self.training
        F, Bk = self.ops_f, self.ops_b
        tape = []                        # (stream, backward emitter) in forward order
        self.relu_outs = []              # (tensor, rows dim, C) of every fused norm + ReLU output, forward order (NetProgram.decisions)
        self.small = []                  # (sums address index or dyn name, nseg dim name or None, C, weight param, bias param)
        relu, elu, none = Fn.ACT['relu'], Fn.ACT['elu'], Fn.ACT['none']
        grad = {}                        # forward tensor -> gradient tensor (backward arena)
        head_grads = {}                  # forward tensor -> (gradient from its head branch on the head stream, level)
        self._pready = {}                # id(parameter) -> number of backward operators after which its gradient is final
        small_rows = []                  # normalisation layers in BACKWARD emission order: (sums address | None, nseg dim, C, weight, bias)
        flushed = [0]

        head_lvl = [None]                # the neck level whose head branch (S_HEAD) is being emitted, None: main chain
        head_written = {}                # level -> parameters whose gradient is written on S_HEAD by that branch

        def wrote(*params):
            for prm in params:
                self._pready[id(prm)] = len(Bk)
            if head_lvl[0] is not None:
                # written on the head stream (or behind it): final for a data-parallel bucket only once the MAIN stream has
                # joined this branch (take(): OP_WAIT EV_HB + level) — GradientAverager.launch_next orders the collective behind
                # the main and the weight-gradient stream, not behind S_HEAD
                head_written.setdefault(head_lvl[0], []).extend(params)

        def flush_small(stream):
            """the d gamma / d beta sums of the normalisation layers whose backward has been emitted since the last flush -> their
            gradient slices (one launch); called where a group of layers ends, so that a data-parallel bucket is complete early"""
            n = len(small_rows) - flushed[0]
            if n:
                self.emit(Bk, OP_SMALL_GRADS, stream, self.DY('small_desc'), 2 * flushed[0], 2 * n)
                for _, _, _, w_, b_ in small_rows[flushed[0]:]:
                    wrote(w_, b_)
                flushed[0] = len(small_rows)
        head_wgrads = [0]

        def head_wfin(stream):
            lo = 1 if self.tail0 else 0
            self.emit(Bk, OP_HEAD_WFIN, stream, self.SA(head_part[lo].data_ptr()), self.nl - lo, Cn, 64, n_reg, n_cls,
                      self.G(nh.centerness_conv.kernel), self.G(nh.reg_conv.kernel), self.G(nh.cls_conv.kernel),
                      self.SA(bias_part[lo].data_ptr()), self.G(nh.cls_conv.bias))
            wrote(nh.centerness_conv.kernel, nh.reg_conv.kernel, nh.cls_conv.kernel, nh.cls_conv.bias)

        def wstream(cur):
            return S_WGRAD if self.wgrad_async else cur

        grad2 = {}                       # forward tensor -> (second gradient contribution, rows, cols, stream), not yet added

        def flush2(t):
            g2, rows, cols, stream = grad2.pop(t)
            self.emit(Bk, OP_ADD, stream, grad[t], g2, self.D(rows), cols)
            grad_src.pop(grad[t], None)          # added to in place: no longer the plain result of the convolution that wrote it
            amax_of.pop(grad[t], None)           # ... nor bounded by its producer's amax word

        def accumulate(t, g, rows, cols, stream):
            """gradient `g` (a backward tensor) arrives for forward tensor t.  The SECOND contribution is kept aside: if the
            consumer of the sum is a BatchNorm backward it adds the two on the fly (norm.hip gy2) and no add pass runs; a third
            arrival, or any other consumer, adds the first two as before (same operands, same order: same bits)"""
            if t not in grad:
                grad[t] = g
            elif Fn.BN_FUSE and t not in grad2:
                grad2[t] = (g, rows, cols, stream)
            else:
                if t in grad2:
                    flush2(t)
                self.emit(Bk, OP_ADD, stream, grad[t], g, self.D(rows), cols)
                grad_src.pop(grad[t], None)
                amax_of.pop(grad[t], None)

        def take(t, rows, cols, pair=False):
            """the complete gradient of forward tensor t, for the emitter of the operator that produced t (main stream); pair:
            the caller adds a second contribution itself -> (gradient, second contribution | None)"""
            if t in head_grads:
                g, lvl = head_grads.pop(t)
                self.emit(Bk, OP_WAIT, S_MAIN, EV_HB + lvl)          # the head branch of this level has delivered
                for prm in head_written.pop(lvl, ()):
                    self._pready[id(prm)] = max(self._pready[id(prm)], len(Bk))
                accumulate(t, g, rows, cols, S_MAIN)
            if pair:
                return grad[t], (grad2.pop(t)[0] if t in grad2 else None)
            if t in grad2:
                flush2(t)
            return grad[t]

        def got(t):
            """grad[t] for a consumer on the tensor's own (head) stream"""
            if t in grad2:
                flush2(t)
            return grad[t]

        def emit_wgrad(cur, *words):
            if self.wgrad_async:
                ev = EV_W0 if cur == S_MAIN else EV_W1
                self.emit(Bk, OP_RECORD, cur, ev)
                self.emit(Bk, OP_WAIT, S_WGRAD, ev)
            self.emit(Bk, OP_WGRAD, wstream(cur), *words)

        AMAX = Fn.X6 and Fn.split_mode() == 2          # the convolutions scale their operands by their amax words (csrc/conv_x6.h h3)
        amax_f = {}                      # (forward tensor, stream) -> address index of its amax word
        amax_of = {}                     # tensor -> address index of the amax word its PRODUCER folds into (fc_amax_out_hint)
        if AMAX:
            # the words the producers fold into start every pass at zero: one fill per direction, its size patched in below
            self.amax_slot()             # (slot 0 doubles as the buffer's base address)
            self.emit(F, OP_CLEAR, S_MAIN, self.SA(self._amax_buf.data_ptr()), 0)

        def fwd_amax(x, rows, cols, stream):
            if not AMAX:
                return -1
            if x in amax_of:
                return amax_of[x]
            if (x, stream) not in amax_f:
                amax_f[(x, stream)] = self.amax_op(F, stream, x, rows, cols)
            return amax_f[(x, stream)]

        producer = {}                    # forward tensor -> (index of the OP_CONV that wrote it, rows, columns)
        grad_src = {}                    # backward tensor -> (index in Bk of the backward-data OP_CONV that wrote it, rows, columns, stream)

        def conv(x, mod, mname, rows_in, rows_out, stream=S_MAIN):
            """MinkowskiConvolution on a kernel map (functional._SparseConv)"""
            Cin, Cout = mod.in_channels, mod.out_channels
            y = self.T(rows_out, Cout)
            m = self.M(mname)
            ax = fwd_amax(x, rows_in, Cin, stream)
            self.emit(F, OP_CONV, stream, x, self.IMG(mod.kernel, False), m, 0, y, -1, Cin, Cout, *([0] * 10), ax + 1)
            producer[y] = (len(F) - 1, rows_out, Cout)
            if tr:
                def bwd():
                    gy = take(y, rows_out, Cout) if stream == S_MAIN else got(y)
                    gx = self.T(rows_in, Cin, 'b')
                    ag = (amax_of[gy] if gy in amax_of else self.amax_op(Bk, stream, gy, rows_out, Cout)) if AMAX else -1
                    self.emit(Bk, OP_CONV, stream, gy, self.IMG(mod.kernel, True), m, 1, gx, -1, Cout, Cin, *([0] * 10), ag + 1)
                    grad_src[gx] = (len(Bk) - 1, rows_in, Cin, stream)
                    emit_wgrad(stream, x, gy, m, self.G(mod.kernel), -1, Cin, Cout, ax + 1, ag + 1)
                    wrote(mod.kernel)
                    accumulate(x, gx, rows_in, Cin, stream)
                tape.append((stream, bwd))
            return y

        def gemm(x, w_tensor, rows, Cin, Cout, gw_dst, stream=S_MAIN):
            """dense GEMM (n, Cin) x (Cin, Cout): the generative transposed convolution, the fused 1x1 heads; gw_dst: the address
            index the (Cin, Cout) weight gradient goes to"""
            y = self.T(rows, Cout)
            ax = fwd_amax(x, rows, Cin, stream)
            self.emit(F, OP_CONV, stream, x, self.IMG(w_tensor, False), -1, 0, y, self.D(rows), Cin, Cout, *([0] * 10), ax + 1)
            producer[y] = (len(F) - 1, rows, Cout)
            if tr:
                def bwd():
                    gy = take(y, rows, Cout) if stream == S_MAIN else got(y)
                    gx = self.T(rows, Cin, 'b')
                    ag = (amax_of[gy] if gy in amax_of else self.amax_op(Bk, stream, gy, rows, Cout)) if AMAX else -1
                    self.emit(Bk, OP_CONV, stream, gy, self.IMG(w_tensor, True), -1, 1, gx, self.D(rows), Cout, Cin, *([0] * 10), ag + 1)
                    grad_src[gx] = (len(Bk) - 1, rows, Cin, stream)
                    emit_wgrad(stream, x, gy, -1, gw_dst, self.D(rows), Cin, Cout, ax + 1, ag + 1)
                    if w_tensor is self.packed:
                        head_wgrads[0] += 1
                        if head_wgrads[0] == self.nl - (1 if self.tail0 else 0) and self.wgrad_async:
                            # every level's partial of the packed head kernel is on its way, all on the weight-gradient stream:
                            # sum them there right away (the head's parameters sit in the first gradient bucket to leave)
                            head_wfin(S_WGRAD)
                    accumulate(x, gx, rows, Cin, stream)
                tape.append((stream, bwd))
            return y

        def bn(x, mod, act, rows, res=None, stream=S_MAIN):
            """MinkowskiBatchNorm (+ residual) + activation (functional.bn_train / norm_act)"""
            b = mod.bn
            C = b.num_features
            y = self.T(rows, C)
            mean, var, cnt = (self.T('one', C), self.T('one', C), self.T('one', 1)) if tr else (-1, -1, -1)
            if act == relu:
                self.relu_outs.append((y, rows, C))
            prod_word, groups = 0, 1
            if tr and Fn.BN_FUSE and x in producer:
                # the convolution that wrote x leaves the column sums of x and x^2 per row block in its epilogue (conv_x6.h /
                # k_sum_*_stats) and this BatchNorm takes its batch statistics from them: no statistics pass over x
                pi, prows, pcols = producer[x]
                groups = pcols // C
                assert pcols == groups * C and groups in (1, 8)
                # table: [blocks][2][pcols] floats, blocks <= rows / 16 + 1 (conv.hip fc_stat_rb)
                F[pi][10] = self.T(prows, pcols // 8, extra=8 * pcols + 256) + 1
                prod_word = pi + 1
            ay = -1
            if AMAX:                     # the apply kernel folds max |y| into this word: a convolution gathering y needs no amax pass
                ay = amax_of[y] = self.amax_slot()
            self.emit(F, OP_BN_FWD, stream, x, self.D(rows), C, _f(b.eps), self.S(b.weight), self.S(b.bias), -1 if res is None else res, act,
                      _f(b.momentum), y, mean, var, cnt, self.S(b.running_mean), self.S(b.running_var), self.S(b.num_batches_tracked),
                      1 if tr else 0, prod_word, groups, ay + 1)
            if tr:
                sums = torch.zeros((2, C), dtype=torch.float32, device=self.dev)
                si = self.S(sums)

                def bwd():
                    small_rows.append((sums.data_ptr(), None, C, b.weight, b.bias))
                    gy, gy2 = take(y, rows, C, pair=True) if stream == S_MAIN else (got(y), None)
                    gx = self.T(rows, C, 'b')
                    gres = self.T(rows, C, 'b') if res is not None else -1
                    prod = 0
                    # the LAST contribution to gy is the result of a backward-data convolution (and at most one other contribution
                    # arrived before it): that launch leaves this layer's two reductions (sum g', sum g' xhat) in its epilogue
                    # (fc_conv_fwd_bn_bwd_stats; `add` = the earlier contribution, bn_y = this layer's output where act' needs it)
                    # — no reduction pass over x, gy (, y)
                    last, first = (gy, None) if gy2 is None else (gy2, gy)
                    src = grad_src.get(last)
                    if Fn.BN_FUSE and src is not None and src[3] == stream and src[1:3] == (rows, C):
                        pi = src[0]
                        Bk[pi][10] = self.T(rows, C // 8, 'b', extra=8 * C + 256) + 1
                        Bk[pi][11:20] = [x + 1, mean, var, self.S(b.weight), self.S(b.bias), _f(b.eps), act,
                                         0 if first is None else first + 1, 0 if res is None else y + 1]
                        prod = pi + 1
                    agx = -1
                    if AMAX:
                        agx = amax_of[gx] = self.amax_slot()
                    self.emit(Bk, OP_BN_BWD, stream, x, y if res is not None else -1, gy, self.D(rows), C, mean, var, cnt, _f(b.eps),
                              self.S(b.weight), self.S(b.bias), act, gx, gres, si, 0 if gy2 is None else gy2 + 1, prod, agx + 1)
                    accumulate(x, gx, rows, C, stream)
                    if res is not None:
                        accumulate(res, gres, rows, C, stream)
                tape.append((stream, bwd))
            return y

        # ---- backbone (me_resnet.py:14-50) ----
        x0 = self.DY('x0')
        seg1 = self.DY('seg1')
        stem, inorm = bb.conv1[0], bb.conv1[1]
        t_stem = self.T('n1', 64)
        col = self.T('n1', 84) if tr else -1
        self.emit(F, OP_STEM_FWD, S_MAIN, x0, self.S(stem.kernel), self.M('stem'), t_stem, col)
        mean_in, var_in, cnt_in = self.T('B', 64), self.T('B', 64), self.T('B', 1)
        self.emit(F, OP_COL_STATS, S_MAIN, t_stem, seg1, self.D('n1'), 64, self.D('B'), mean_in, var_in, cnt_in)
        t_in = self.T('n1', 64)
        self.emit(F, OP_NORM_FWD, S_MAIN, t_stem, seg1, self.D('n1'), 64, mean_in, var_in, _f(inorm.eps), self.S(inorm.weight),
                  self.S(inorm.bias), -1, relu, t_in)
        self.relu_outs.append((t_in, 'n1', 64))
        t_pool, arg = self.T('n2', 64), self.T('n2', 64)
        self.pool_arg = (arg, 'n2', 64)
        ap = -1
        if AMAX:
            ap = amax_of[t_pool] = self.amax_slot()
        self.emit(F, OP_MAXPOOL_FWD, S_MAIN, t_in, self.M('pool'), 64, t_pool, arg, ap + 1)
        if tr:
            in_sums = self.T('B', 2 * 64, 'b')
            self.in_sums = in_sums

            def bwd_stem():
                small_rows.append((None, 'B', 64, inorm.weight, inorm.bias))
                g_pool = take(t_pool, 'n2', 64)
                g_in = self.T('n1', 64, 'b')
                self.emit(Bk, OP_MAXPOOL_BWD, S_MAIN, g_pool, arg, self.M('pool'), 64, g_in)
                g_stem = self.T('n1', 64, 'b')
                # instance norm + ReLU without a residual: y is not read (norm.hip bn_pre recomputes act')
                self.emit(Bk, OP_NORM_BWD, S_MAIN, t_stem, -1, g_in, seg1, self.D('n1'), 64, self.D('B'), mean_in, var_in, cnt_in,
                          _f(inorm.eps), self.S(inorm.weight), self.S(inorm.bias), relu, g_stem, -1, in_sums)
                if self.wgrad_async:
                    self.emit(Bk, OP_RECORD, S_MAIN, EV_W0)
                    self.emit(Bk, OP_WAIT, S_WGRAD, EV_W0)
                self.emit(Bk, OP_STEM_WGRAD, wstream(S_MAIN), col, g_stem, self.M('stem'), self.G(stem.kernel))
                wrote(stem.kernel)
            tape.append((S_MAIN, bwd_stem))
        cur, cur_rows, cur_C = t_pool, 'n2', 64
        levels = []
        for li in range(1, self.nl + 1):
            rows = f'L{li}'
            if tr:
                tape.append((S_MAIN, lambda: flush_small(S_MAIN)))       # reversed walk: reached after this layer's backward
            for j, blk in enumerate(getattr(bb, f'layer{li}')):
                planes = blk.conv1.out_channels
                if j == 0:
                    assert blk.downsample is not None and blk.conv1.stride == 2
                    t_ds = conv(cur, blk.downsample[0], f'ds{li}', cur_rows, rows)
                    res = bn(t_ds, blk.downsample[1], none, rows)
                    t1 = conv(cur, blk.conv1, f'down{li}', cur_rows, rows)
                else:
                    assert blk.downsample is None
                    res = cur
                    t1 = conv(cur, blk.conv1, f'same{li}', rows, rows)
                t1 = bn(t1, blk.norm1, relu, rows)
                t2 = conv(t1, blk.conv2, f'same{li}', rows, rows)
                cur = bn(t2, blk.norm2, relu, rows, res=res)
                cur_rows, cur_C = rows, planes
            levels.append((cur, rows, cur_C))
        # ---- neck + head (fcaf3d_neck_with_head.py:94-108, :256-279) ----
        n_reg, n_cls = nh.n_reg_outs, nh.n_classes
        Cn = nh.centerness_conv.in_channels
        cent_all, bbox_all = self.T('Nall', 1), self.T('Nall', n_reg)
        cls_all, cmax_all = self.T('Nall', n_cls), self.T('Nall', 1)
        self.out_idx = (cent_all, bbox_all, cls_all, cmax_all)
        head_part = None
        if tr:
            tape.append((S_MAIN, lambda: flush_small(S_MAIN)))           # ... after the whole neck's backward
            self.gout_idx = (self.DY('g_cent'), self.DY('g_bbox'), self.DY('g_cls'))
            head_part = torch.zeros((self.nl, Cn, 64), dtype=torch.float32, device=self.dev)
            # per level: the column sums of d loss / d cls_score (the class-bias gradient's share), left by OP_HEAD_BWD
            bias_part = torch.zeros((self.nl, n_cls), dtype=torch.float32, device=self.dev)
            self.keep += [head_part, bias_part]
        x, x_rows, x_C = levels[-1]
        for i in range(self.nl - 1, -1, -1):
            if i < self.nl - 1:
                up = getattr(nh, f'up_block_{i + 1}')
                gi = self.gents.index(up[0])
                g_rows = f'g{i}'
                Ci = up[0].out_channels
                # generative transposed convolution: one dense GEMM (n, Cin) x (Cin, 8 Cout) whose (n, 8 Cout) output IS the
                # (8 n, Cout) children matrix (row 8 i + k); its weight gradient comes out in the GEMM's layout and is permuted
                # into the layer's (8, Cin, Cout) kernel gradient afterwards
                gw_tmp = torch.zeros((x_C, 8 * Ci), dtype=torch.float32, device=self.dev)
                self.keep.append(gw_tmp)
                gw_i = self.S(gw_tmp)
                if tr:
                    def bwd_perm(gw_i=gw_i, mod=up[0], Cin=x_C, Ci=Ci):
                        self.emit(Bk, OP_PERMUTE_GENT, wstream(S_MAIN), gw_i, self.G(mod.kernel), Cin, Ci)
                        wrote(mod.kernel)
                        flush_small(S_MAIN)                     # ... and after every neck level's up-block
                    tape.append((S_MAIN, bwd_perm))          # forward order: BEFORE the GEMM, so the reversed walk reaches it after
                t = gemm(x, self.gent_w[gi], x_rows, x_C, 8 * Ci, gw_i)
                t = bn(t, up[1], elu, g_rows)
                t = conv(t, up[3], f'gsame{i}', g_rows, g_rows)
                t = bn(t, up[4], elu, g_rows)
                # x = inputs[i] + t: every backbone voxel lies in the generated set -> the union IS the generated set (sparse.CoordMap.union)
                fb, fb_rows, fb_C = levels[i]
                u = self.T(g_rows, Ci)
                rows_i = self.DY(f'rows{i}')
                self.emit(F, OP_UNION_FWD, S_MAIN, t, fb, rows_i, self.D(g_rows), self.D(fb_rows), self.D(g_rows), Ci, u)
                if AMAX:                 # no producer folds a union: ONE pass here, before the head branch forks, serves both streams' consumers
                    amax_of[u] = self.amax_op(F, S_MAIN, u, g_rows, Ci)
                if tr:
                    def bwd_union(u=u, t=t, fb=fb, fb_rows=fb_rows, rows_i=rows_i, g_rows=g_rows, Ci=Ci):
                        gu = take(u, g_rows, Ci)
                        accumulate(t, gu, g_rows, Ci, S_MAIN)          # d out / d fa = identity on all rows (n_a == n_union)
                        gfb = self.T(fb_rows, Ci, 'b')
                        self.emit(Bk, OP_GATHER, S_MAIN, gu, rows_i, self.D(fb_rows), Ci, gfb)
                        accumulate(fb, gfb, fb_rows, Ci, S_MAIN)
                    tape.append((S_MAIN, bwd_union))
                x, x_rows, x_C = u, g_rows, Ci
                kname = f'gsame{i}'
            else:
                kname = f'same{self.nl}'
            if i == 0 and self.tail0:
                self.x0_idx = (x, x_rows, x_C)              # the pruned level's tail runs outside the program
                if tr:
                    grad[x] = self.DY('g_x0')               # ... and hands back d loss / d x0
                continue
            # out_block_i + forward_single: on the head stream for i > 0
            hs = S_HEAD if (self.head_overlap and i > 0) else S_MAIN
            if hs == S_HEAD:
                self.emit(F, OP_RECORD, S_MAIN, EV_FORK + i)
                self.emit(F, OP_WAIT, S_HEAD, EV_FORK + i)
            ob = getattr(nh, f'out_block_{i}')
            o = conv(x, ob[0], kname, x_rows, x_rows, stream=hs)
            o = bn(o, ob[1], elu, x_rows, stream=hs)
            gw_head = self.SA(head_part[i].data_ptr()) if tr else -1
            y = gemm(o, self.packed, x_rows, Cn, 64, gw_head, stream=hs)
            off = f'off{i}'
            outs = [self.AL(cent_all, off, 1), self.AL(bbox_all, off, n_reg), self.AL(cls_all, off, n_cls), self.AL(cmax_all, off, 1)]
            self.emit(F, OP_HEAD_FWD, hs, y, 64, self.S(nh.cls_conv.bias), self.S(nh.scales[i].scale), self.D(x_rows), n_reg, n_cls, *outs)
            if tr:
                def bwd_head(y=y, outs=outs, off=off, hs=hs, rows=x_rows, lvl=i):
                    gy = self.T(rows, 64, 'b')
                    gin = [self.AL(g, off, c) for g, c in zip(self.gout_idx, (1, n_reg, n_cls))]
                    # ... with d loss / d scale of this level and this level's share of the class-bias gradient (column sums of
                    # d loss / d cls_score; summed over the levels by OP_HEAD_WFIN) out of the same pass (r5: they were single-block
                    # reductions over every location, k_col_sum: 0.35 ms per step on the dependent chain)
                    agy = -1
                    if AMAX:                 # the kernel folds max |gy| for the backward-data GEMM behind it
                        agy = amax_of[gy] = self.amax_slot()
                    self.emit(Bk, OP_HEAD_BWD, hs, y, 64, self.S(nh.scales[lvl].scale), outs[1], gin[0], gin[1], gin[2], self.D(rows), n_reg,
                              n_cls, gy, -1, self.G(nh.scales[lvl].scale), self.SA(bias_part[lvl].data_ptr()), agy + 1)
                    wrote(nh.scales[lvl].scale)
                    grad[y] = gy
                tape.append((hs, bwd_head))
                if hs == S_HEAD:
                    tape.append((S_HEAD, ('join', x, x_rows, x_C, i)))      # marker: where this level's head branch begins in reverse
        if self.head_overlap:
            self.emit(F, OP_RECORD, S_HEAD, EV_HEAD_DONE)
            self.emit(F, OP_WAIT, S_MAIN, EV_HEAD_DONE)
        # ---- backward program -------------------------------------------------------------------------------------------
        if AMAX:
            n_fwd_slots = self._amax_used
            F[0][3] = self.AMAX_SLOT_BYTES * n_fwd_slots                     # the forward pass clears its own words; they stay valid through the backward pass
            if tr:
                self.emit(Bk, OP_CLEAR, S_MAIN, self.SA(self._amax_buf.data_ptr() + self.AMAX_SLOT_BYTES * n_fwd_slots), 0)
                clear_b = len(Bk) - 1
        if tr:
            if self.head_overlap:
                # the head branches depend on the loss gradients only: their backward is enqueued first, on the head stream, finest
                # forked level first (the main chain needs that one first); each leaves the gradient of its neck tensor behind an event
                self.emit(Bk, OP_RECORD, S_MAIN, EV_BWD0)          # the loss gradients are ready on the caller's stream
                self.emit(Bk, OP_WAIT, S_HEAD, EV_BWD0)
                branches, cur_b = [], None
                for s, e in reversed(tape):
                    if s != S_HEAD:
                        continue
                    if isinstance(e, tuple):
                        cur_b = [e, []]
                        branches.append(cur_b)
                    else:
                        cur_b[1].append(e)
                for (_, xt, rows, C, lvl), ems in sorted(branches, key=lambda b: b[0][4]):
                    saved = grad.pop(xt, None)
                    assert saved is None
                    head_lvl[0] = lvl
                    for e in ems:
                        e()
                    flush_small(S_HEAD)                    # this branch's normalisation layer
                    head_lvl[0] = None
                    self.emit(Bk, OP_RECORD, S_HEAD, EV_HB + lvl)
                    head_grads[xt] = (got(xt), lvl)
                    grad.pop(xt)
            for s, e in reversed(tape):
                if s == S_MAIN:
                    e()
            assert not head_grads and not head_written, 'a head branch was never joined'
            if not self.wgrad_async:
                # without the weight-gradient stream the partials were written on the streams of their levels (main, head): the
                # main stream has joined every head branch by now
                head_wfin(S_MAIN)
            flush_small(S_MAIN)                            # what is left: the stem's instance norm
            assert head_wgrads[0] == self.nl - (1 if self.tail0 else 0)
            self.small = small_rows
            missing = [n for n, prm in det.named_parameters() if prm.requires_grad and id(prm) not in self._pready]
            if self.tail0:                                 # out_block_0 and scales.0 get their gradients on the per-operator path
                missing = [n for n in missing if not (n.startswith('neck_with_head.out_block_0.') or n == 'neck_with_head.scales.0.scale')]
            assert not missing, missing
            if self.wgrad_async:
                self.emit(Bk, OP_RECORD, S_WGRAD, EV_WEND)
                self.emit(Bk, OP_WAIT, S_MAIN, EV_WEND)
            if AMAX:
                Bk[clear_b][3] = self.AMAX_SLOT_BYTES * (self._amax_used - n_fwd_slots)

    # ---- per-step tables ---------------------------------------------------------------------------------------------------
    def _finalise(self):
        self.ops_f = np.ascontiguousarray(np.asarray(self.ops_f, dtype=np.int64).reshape(-1, OPW))
        self.ops_b = np.ascontiguousarray(np.asarray(self.ops_b, dtype=np.int64).reshape(-1, OPW)) if self.ops_b else np.zeros((0, OPW), np.int64)
        self.addr0 = np.zeros(self.n_addr, dtype=np.int64)
        for i, a in self.static:
            self.addr0[i] = a
        self._ar = {}
        for k in ('f', 'b'):
            a = self.arena[k]
            self._ar[k] = (np.array([t[0] for t in a], dtype=np.int64), np.array([t[1] for t in a], dtype=np.int64),
                           np.array([t[2] for t in a], dtype=np.int64), np.array([t[3] for t in a], dtype=np.int64))
        self._al = (np.array([t[0] for t in self.alias], dtype=np.int64), np.array([t[1] for t in self.alias], dtype=np.int64),
                    np.array([t[2] for t in self.alias], dtype=np.int64), np.array([t[3] for t in self.alias], dtype=np.int64))
        self._gr = (np.array([t[0] for t in self.grad_refs], dtype=np.int64), np.array([t[1] for t in self.grad_refs], dtype=np.int64) * 4)
        self.n_dims, self.n_maps = len(self.dim_names), len(self.map_names)
        self._ws = [None, None, None]
        self._need = np.zeros(3, dtype=np.int64)
        self._cfg = np.array([Fn.BN_SMALL_ELEMS, Fn.FLAGS, 0, 0], dtype=np.int64)
        self.n_conv_f = int((self.ops_f[:, 0] == OP_CONV).sum())
        self.n_conv_b = int((self.ops_b[:, 0] == OP_CONV).sum()) if len(self.ops_b) else 0
        self.n_amax = int((self.ops_f[:, 0] == OP_AMAX).sum()) + (int((self.ops_b[:, 0] == OP_AMAX).sum()) if len(self.ops_b) else 0)
        self._anchor = torch.zeros(1, device=self.dev, requires_grad=True)
        if self.training:
            # small-gradient descriptor template: (src, dst, C, nseg, stride, 0, 0, 0) per (bias, weight) of every normalisation layer
            n = len(self.small)
            self._small = np.zeros((2 * n, 8), dtype=np.int64)
            self._small_goff = np.zeros(2 * n, dtype=np.int64)
            self._small_in = None
            for k, (src, nseg, C, w, b) in enumerate(self.small):
                for h, p in ((0, b), (1, w)):          # sums[0] = d beta, sums[1] = d gamma
                    r = 2 * k + h
                    self._small[r, 2], self._small[r, 3], self._small[r, 4] = C, 1, 2 * C
                    self._small_goff[r] = self._goff[id(p)] * 4
                    if src is not None:
                        self._small[r, 0] = src + 4 * C * h
                    else:
                        self._small_in = (2 * k, C)

    def _fill_arena(self, addr, dims, which, base):
        idx, di, bpr, ext = self._ar[which]
        if len(idx) == 0:
            return
        sizes = (dims[di] * bpr + ext + (ALIGN - 1)) // ALIGN * ALIGN
        offs = np.cumsum(sizes) - sizes
        addr[idx] = base + offs

    def _arena_bytes(self, dims, which):
        idx, di, bpr, ext = self._ar[which]
        return int(((dims[di] * bpr + ext + (ALIGN - 1)) // ALIGN * ALIGN).sum()) + ALIGN

    def bind(self, x, batch_size, backward):
        """per-step tables from the input SparseTensor's coordinate maps (all cached lookups after plan_maps); None if this
        step's maps do not fit the program (a union that is not the generated set)"""
        nl = self.nl
        cm0 = x.cmap
        dims = np.zeros(self.n_dims, dtype=np.int64)
        maps = np.zeros((self.n_maps, MAPW), dtype=np.int64)
        dn, mn = self.dim_names, self.map_names

        def setd(k, v):
            if k in dn:
                dims[dn[k]] = v
        m1 = cm0.strided(2)
        m2 = m1.strided(2)
        maps[mn['stem']] = cm0.kernel_map(m1, 3).desc(conv=False)
        maps[mn['pool']] = m1.kernel_map(m2, 2).desc(conv=False)
        kms = {}
        setd('n0', cm0.n); setd('n1', m1.n); setd('n2', m2.n); setd('B', batch_size); setd('one', 1)
        prev, lv = m2, []
        for li in range(1, nl + 1):
            mi = prev.strided(2)
            kms[f'down{li}'], kms[f'ds{li}'], kms[f'same{li}'] = prev.kernel_map(mi, 3), prev.kernel_map(mi, 1), mi.kernel_map(mi, 3)
            for k in (f'down{li}', f'ds{li}', f'same{li}'):
                maps[mn[k]] = kms[k].desc(True, backward)
            setd(f'L{li}', mi.n)
            lv.append(mi)
            prev = mi
        xm = lv[-1]
        head_maps = [xm]
        dyn = {}
        for i in range(nl - 2, -1, -1):
            g = xm.generate()
            u, rows, swapped = lv[i].union(g)
            if not swapped or u is not g:
                return None
            kms[f'gsame{i}'] = g.kernel_map(g, 3)
            maps[mn[f'gsame{i}']] = kms[f'gsame{i}'].desc(True, backward)
            setd(f'g{i}', g.n)
            dyn[f'rows{i}'] = rows
            xm = g
            head_maps.append(g)
        head_maps = head_maps[::-1]                       # finest first
        off = 0
        for i, cm in enumerate(head_maps):
            if i == 0 and self.tail0:
                continue                                  # (its head outputs come from the per-operator tail)
            setd(f'off{i}', off)
            off += cm.n
        setd('Nall', off)
        st = dict(dims=dims, maps=maps, dyn=dyn, head_maps=head_maps, x=x, seg1=m1.coords, n_all=off)
        if PROBE is not None:
            # valid (input, output) pairs of every map, as device scalars (read back after the timed region): the FLOPs of a launch
            from .sparse import _rec
            st['pairs'] = {mn[k]: _rec(km._pairs[3].sum() if km._pairs is not None else (km.nbr >= 0).sum()) for k, km in kms.items()}
            PROBE.append(dict(pairs=st['pairs'], nf=self.n_conv_f, nb=self.n_conv_b, na=self.n_amax))
        return st

    def _streams(self):
        dev = self.dev
        nh = self.det.neck_with_head
        main = L.stream()
        head = nh._head_stream(dev).cuda_stream if self.head_overlap else main
        wg = Fn.wgrad_stream(dev).cuda_stream if self.wgrad_async else main
        return np.array([main, head, wg], dtype=np.int64)

    def _run(self, ops, addr, st, begin=0, end=None, size_only=False):
        """size_only: the sizing pass alone (cfg[3]) — grows the scratch buffers to what [begin, end) needs and launches nothing.
        The segmented data-parallel backward sizes its WHOLE list first: a buffer replaced between two segments would go back to
        the caching allocator while kernels of an earlier segment on the head / weight-gradient stream still use it"""
        streams = self._streams()
        self._cfg[2] = 1 if 'pairs' in st else 0
        self._cfg[3] = 1 if size_only else 0
        end = len(ops) if end is None else end
        while True:
            ws = np.array([w.data_ptr() if w is not None else 0 for w in self._ws], dtype=np.int64)
            wsb = np.array([w.numel() if w is not None else 0 for w in self._ws], dtype=np.int64)
            rc = L.lib().fc_exec(ops.ctypes.data, begin, end, addr.ctypes.data, st['dims'].ctypes.data, st['maps'].ctypes.data,
                                 streams.ctypes.data, ws.ctypes.data, wsb.ctypes.data, self._need.ctypes.data, self._cfg.ctypes.data)
            if rc == -2:                                  # nothing was launched: grow the scratch buffers and go again
                for i in range(3):
                    if self._need[i] > wsb[i]:
                        self._ws[i] = torch.empty(int(self._need[i] * 5 // 4) + (1 << 20), dtype=torch.uint8, device=self.dev)
                continue
            if rc != 0:
                raise RuntimeError(f'fc_exec failed: {"invalid argument" if rc == -1 else "hipError %d" % rc}')
            return

    def decisions(self, st):
        """(ReLU sign patterns in forward order, max-pool arg-max rows) of the forward pass bound in `st` — what
        oracle.DecisionTape replays (test infrastructure reads it; needs KEEP_STATE)"""
        fa, addr, dims = st['fa'], st['addr'], st['dims']

        def view(idx, rows, C, dtype):
            n = int(dims[self.dim_names[rows]])
            o = int(addr[idx] - fa.data_ptr())
            return fa[o:o + n * C * 4].view(dtype).view(n, C)
        relu = [(view(t, r, C, torch.float32) > 0).cpu() for t, r, C in self.relu_outs]
        return relu, view(*self.pool_arg, torch.int32).cpu()

    def forward(self, st):
        """-> (cent_all, bbox_all, cls_all, cmax_all) for all head locations (levels finest first), autograd-connected in training"""
        trusted = (TRUSTED is not None and TRUSTED.w is self.w) or (not self.training and getattr(self.det, 'static_weights', False))
        if not (self.weights_fresh and trusted):
            self.refresh_weights()
        ev = self.images.event
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)     # images may have been built on another stream (runner: after the optimizer step)
        if self.training:
            return _NetFn.apply(self._anchor, self, st)
        return self._forward(st)

    def _forward(self, st):
        dims = st['dims']
        fa = torch.empty(self._arena_bytes(dims, 'f'), dtype=torch.uint8, device=self.dev)
        addr = self.addr0.copy()
        base = (fa.data_ptr() + ALIGN - 1) // ALIGN * ALIGN
        self._fill_arena(addr, dims, 'f', base)
        addr[self.dyn['x0']] = st['x'].F.data_ptr()
        addr[self.dyn['seg1']] = st['seg1'].data_ptr()
        for k, t in st['dyn'].items():
            addr[self.dyn[k]] = t.data_ptr()
        ai, ap, ad, ab = self._al
        fwd_alias = addr[ap] != 0                         # aliases of forward tensors (the backward ones are resolved later)
        addr[ai[fwd_alias]] = addr[ap[fwd_alias]] + dims[ad[fwd_alias]] * ab[fwd_alias]
        st['fa'], st['addr'] = fa, addr
        self._run(self.ops_f, addr, st)
        n = st['n_all']
        nh = self.det.neck_with_head
        outs = []
        for idx, c in zip(self.out_idx, (1, nh.n_reg_outs, nh.n_classes, 1)):
            o = int(addr[idx] - fa.data_ptr())
            outs.append(fa[o:o + n * c * 4].view(torch.float32).view(n, c))
        if self.tail0:
            t, rows, C = self.x0_idx
            n0 = int(dims[self.dim_names[rows]])
            o = int(addr[t] - fa.data_ptr())
            outs.append(fa[o:o + n0 * C * 4].view(torch.float32).view(n0, C))
        return tuple(outs)

    def _backward(self, st, g_cent, g_bbox, g_cls, g_x0=None):
        dims, addr = st['dims'], st['addr']
        n = st['n_all']
        nh = self.det.neck_with_head
        gs = []
        for g, c in zip((g_cent, g_bbox, g_cls), (1, nh.n_reg_outs, nh.n_classes)):
            gs.append(g.contiguous() if g is not None else torch.zeros((n, c), dtype=torch.float32, device=self.dev))
        if self.tail0:
            t, rows, C = self.x0_idx
            n0 = int(dims[self.dim_names[rows]])
            g_x0 = g_x0.contiguous() if g_x0 is not None else torch.zeros((n0, C), dtype=torch.float32, device=self.dev)
            gs.append(g_x0)                                 # (kept alive until the operators have been enqueued)
        ba = torch.empty(self._arena_bytes(dims, 'b'), dtype=torch.uint8, device=self.dev)
        base = (ba.data_ptr() + ALIGN - 1) // ALIGN * ALIGN
        self._fill_arena(addr, dims, 'b', base)
        for k, g in zip(('g_cent', 'g_bbox', 'g_cls', 'g_x0'), gs):
            addr[self.dyn[k]] = g.data_ptr()
        ai, ap, ad, ab = self._al
        addr[ai] = addr[ap] + dims[ad] * ab
        # gradients that are already there (a second backward pass without zero_grad: accumulation) are added to afterwards;
        # then, and without flat storage, this pass writes into a buffer of its own (same layout)
        own = self._own_params()
        # Parameters that already hold a gradient when this pass starts: (a) a few, in tensors of their own — the pruned level's
        # per-operator tail has delivered the head kernels' share before this node ran: the program writes its share into the
        # buffer as always and the earlier one is added there afterwards; (b) views of the very buffer this pass writes (a second
        # backward without zero_grad): then, and without flat storage, the pass writes into a buffer of its own and is added on top
        pre = [(p, p.grad) for p in own if p.grad is not None]
        aliased = self.flat is not None and any(g.data_ptr() == self.flat.grad_view(p).data_ptr() for p, g in pre)
        accum = bool(pre) and (self.flat is None or aliased or len(pre) > 16)
        if self.flat is not None and not accum:
            gbuf = self.flat.grad
        else:
            gbuf = torch.zeros(self._gtotal, dtype=torch.float32, device=self.dev)
        gi, go = self._gr
        addr[gi] = gbuf.data_ptr() + go
        desc = self._small.copy()
        desc[:, 1] = gbuf.data_ptr() + self._small_goff
        r, C = self._small_in
        B = int(dims[self.dim_names['B']])
        desc[r, 0], desc[r + 1, 0] = addr[self.in_sums], addr[self.in_sums] + 4 * C
        desc[r:r + 2, 3] = B
        desc_dev = L.upload(desc, self.dev)
        addr[self.dyn['small_desc']] = desc_dev.data_ptr()
        gv = self._grad_views(gbuf)
        if not accum:
            for p in own:                                   # before the operators run: a data-parallel bucket may leave mid-way
                p.grad = gv[id(p)]
        from . import dist as D
        av = D.ACTIVE
        if av is not None and av.buckets and av.flat is self.flat and self.flat is not None and not accum and not self.tail0:
            # data parallel: every gradient bucket leaves (all-reduce on the weight-gradient stream) as soon as the operators that
            # write its parameters are enqueued — what the autograd hooks of the per-operator path do
            pos = 0
            self._run(self.ops_b, addr, st, size_only=True)
            for upto in self._bucket_ready(av):
                if upto > pos:
                    self._run(self.ops_b, addr, st, pos, upto)
                    pos = upto
                av.launch_next()
            if pos < len(self.ops_b):
                self._run(self.ops_b, addr, st, pos, len(self.ops_b))
        else:
            self._run(self.ops_b, addr, st)
        if accum:
            for p in own:
                if p.grad is None:
                    p.grad = gv[id(p)]
                else:
                    p.grad.add_(gv[id(p)])
        else:
            for p, g0 in pre:                               # (a): the share that was there first
                gv[id(p)].add_(g0)
            if self.flat is not None and not self.tail0:
                self.flat.complete = True
        st['ba'] = ba                                       # until the caller drops the step state

    def _own_params(self):
        """the parameters whose gradients this program writes (all of them, unless the pruned level's tail runs outside it)"""
        c = getattr(self, '_own', None)
        if c is None:
            c = self._own = [p for p in self._params if id(p) in self._pready]
        return c

    def _bucket_ready(self, av):
        """per bucket of the averager (in launch order): the number of backward operators after which all of its gradients are
        written (non-decreasing: buckets leave strictly in order)"""
        key = id(av)
        c = getattr(self, '_br', None)
        if c is None or c[0] != key:
            out, run = [], 0
            for b in av.buckets:
                run = max(run, max(self._pready[id(p)] for p in b))
                out.append(run)
            c = self._br = (key, out)
        return c[1]

    def _grad_views(self, gbuf):
        key = gbuf.data_ptr()
        cache = getattr(self, '_gv', None)
        if cache is None or cache[0] != key:
            views = {id(p): gbuf[self._goff[id(p)]:self._goff[id(p)] + p.numel()].view(p.shape) for p in self._params}
            cache = self._gv = (key, views, gbuf)
        return cache[1]


class _Weights:
    """the executor's view of a detector's weights: pre-split images of every convolution kernel (functional.WeightImages), the
    three 1x1 head kernels packed side by side (zero-padded to 64 columns), the generative transposed convolutions' kernels in
    their GEMM form (Cin, 8 Cout) — persistent buffers, rebuilt by `refresh` (5-6 launches on the current stream)"""

    def __init__(self, det, dev, sig):
        nh = det.neck_with_head
        self.det, self.sig = det, sig
        head = (nh.centerness_conv, nh.reg_conv, nh.cls_conv)
        convs = [m for m in det.modules() if isinstance(m, MEnn.MinkowskiConvolution) and m not in head]
        self.gents = [m for m in det.modules() if isinstance(m, MEnn.MinkowskiGenerativeConvolutionTranspose)]
        C = nh.centerness_conv.in_channels
        self.ncol = 1 + nh.n_reg_outs + nh.n_classes
        self.packed = torch.zeros((1, C, 64), dtype=torch.float32, device=dev)       # [centerness | reg | cls | 0...]
        self.gent_w = [torch.empty((1, m.in_channels, 8 * m.out_channels), dtype=torch.float32, device=dev) for m in self.gents]
        ws = [m.kernel for m in convs] + [self.packed] + self.gent_w
        self.images = Fn.WeightImages(ws)
        self.img_of = {}
        for w in ws:
            K = 1 if w.dim() == 2 else w.shape[0]
            self.img_of[w.data_ptr()] = self.images.table.get((w.data_ptr(), K, w.shape[-2], w.shape[-1]))
        self.fresh = False

    @torch.no_grad()
    def refresh(self):
        nh = self.det.neck_with_head
        torch.cat((nh.centerness_conv.kernel, nh.reg_conv.kernel, nh.cls_conv.kernel), dim=1, out=self.packed[0][:, :self.ncol])
        for m, w in zip(self.gents, self.gent_w):
            w.view(m.in_channels, 8, m.out_channels).copy_(m.kernel.permute(1, 0, 2))
        self.images.build()
        self.fresh = True


class _NetFn(torch.autograd.Function):
    """The whole network body as one autograd node: forward = one fc_exec call, backward = one fc_exec call that leaves every
    parameter gradient in the (flat) gradient buffer."""

    @staticmethod
    def forward(ctx, anchor, prog, st):
        ctx.prog, ctx.st = prog, st
        outs = prog._forward(st)
        ctx.mark_non_differentiable(outs[3])
        return outs

    @staticmethod
    def backward(ctx, g_cent, g_bbox, g_cls, _g_max, g_x0=None):
        ctx.prog._backward(ctx.st, g_cent, g_bbox, g_cls, g_x0)    # (the state is released with the graph; retain_graph keeps it)
        return None, None, None


def program_for(det, training, tail0=False):
    """the cached NetProgram of this detector for the current mode / stream configuration (None: not covered -> module path)"""
    if not supported(det):
        return None
    nh = det.neck_with_head
    if tail0 and min(det.backbone.n_outs, 4) < 2:
        return None
    key = (bool(training), bool(Fn.WGRAD_ASYNC), bool(nh.head_overlap), bool(tail0))
    fuse = bool(Fn.BN_FUSE)                    # changes the emitted program (statistics tables, producer words): part of the cache key
    # ... and so does the operand split (r6): the three-product route's program carries amax slots, hint words and clear operators,
    # and the weight images are mode-dependent
    mode = Fn.split_mode() if Fn.X6 else -1
    cache = det.__dict__.setdefault('_programs', {})
    sig = NetProgram.signature(det)
    prog = cache.get(key + (fuse, mode))
    if prog is None or prog._sig != sig:
        prog = cache[key + (fuse, mode)] = NetProgram(det, *key)
    if getattr(prog.w, 'split_mode', mode) != mode:      # the shared images were built in the other mode
        prog.w.fresh = False
    prog.w.split_mode = mode
    return prog
