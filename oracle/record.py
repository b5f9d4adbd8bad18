"""ORACLE-side test infrastructure: record the DISCRETE decisions of a HIP forward pass of the detector so that
oracle.model_oracle.DecisionTape can replay them (ReLU sign patterns of the backbone in execution order, the arg-max rows of the
stem's max-pooling, the coordinate sets `_prune` keeps).  Used by tests/test_gpu_model.py and __graft_entry__.smoke(); never by
the product.  Works for both routes of the product: the per-operator module path (hooks on the autograd Functions) and the native
executor (reads the recorded activations of the bound step, fcaf3d_amd.executor.NetProgram.decisions)."""


class RecordDecisions:
    def __init__(self, model=None):
        self.model = model

    def __enter__(self):
        import fcaf3d_amd.executor as E
        import fcaf3d_amd.functional as Fn
        import fcaf3d_amd.nn as MEnn
        self.Fn, self.E, self.MEnn = Fn, E, MEnn
        self.relu, self.pool, self.prune = [], [], []
        self.saved = (Fn._NormAct.forward, Fn._BNTrainSmall.forward, Fn._MaxPool.forward)
        self.fused0 = Fn._BNTrainFused.forward            # r5: BatchNorm with statistics from the convolution epilogue (module path)
        self.prune0 = MEnn.MinkowskiPruning.forward
        self.keep0 = E.KEEP_STATE
        E.KEEP_STATE = True
        if self.model is not None:
            self.model._last_exec = None
        na0, bs0, mp0 = self.saved
        rec = self

        def pr(mod, x, mask, **kw):
            if not bool(mask.all()):
                rec.prune.append(x.C[mask].cpu().numpy())
            return rec.prune0(mod, x, mask, **kw)
        MEnn.MinkowskiPruning.forward = pr

        def na(ctx, x, gamma, beta, residual, seg, nseg, eps, act, *rest):
            y = na0(ctx, x, gamma, beta, residual, seg, nseg, eps, act, *rest)
            if act == Fn.ACT['relu']:
                rec.relu.append((y > 0).cpu())
            return y

        def bs(ctx, x, gamma, beta, residual, eps, act, *rest):
            out = bs0(ctx, x, gamma, beta, residual, eps, act, *rest)
            if act == Fn.ACT['relu']:
                rec.relu.append((out[0] > 0).cpu())
            return out

        fu0 = self.fused0

        def fu(ctx, x, gamma, beta, residual, eps, act, *rest):
            out = fu0(ctx, x, gamma, beta, residual, eps, act, *rest)
            if act == Fn.ACT['relu']:
                rec.relu.append((out[0] > 0).cpu())
            return out
        Fn._BNTrainFused.forward = staticmethod(fu)

        def mp(ctx, feats, kmap):
            out = mp0(ctx, feats, kmap)
            rec.pool.append(ctx.to_save[0].cpu())          # the arg-max rows (saved for backward)
            return out
        Fn._NormAct.forward, Fn._BNTrainSmall.forward, Fn._MaxPool.forward = staticmethod(na), staticmethod(bs), staticmethod(mp)
        return self

    def __exit__(self, *a):
        Fn = self.Fn
        Fn._NormAct.forward, Fn._BNTrainSmall.forward, Fn._MaxPool.forward = (staticmethod(f) for f in self.saved)
        Fn._BNTrainFused.forward = staticmethod(self.fused0)
        self.MEnn.MinkowskiPruning.forward = self.prune0
        self.E.KEEP_STATE = self.keep0
        last = getattr(self.model, '_last_exec', None) if self.model is not None else None
        if last is not None:                               # the step went through the native executor: read its activations
            prog, st = last
            relu, pool = prog.decisions(st)
            assert not self.relu and not self.pool, 'a step is either executed natively or per operator'
            self.relu, self.pool = relu, [pool]
            self.model._last_exec = None

    def tape(self):
        from . import model_oracle as MO
        assert len(self.pool) == 1, 'exactly one forward pass must have been recorded'
        return MO.DecisionTape(self.relu, self.pool[0], self.prune)
