"""The coordinate phase of one step through the two native calls of csrc/plan.hip (r6).

`plan_step(det, points, ...)` replaces, for the standard FCAF3D topologies (BasicBlock backbone, default routing switches), what
`SingleStageSparse3DDetector.voxelize` + `SparseTensor(...)` + `plan_maps` did with ~420 launches and ~25 blocking read-backs driven
from Python (reference: the collate + ME.SparseTensor + every kernel-map build inside extract_feat,
mmdet3d/models/detectors/single_stage_sparse.py:32-40): `fc_plan_levels` (one read-back) + `fc_plan_maps` (one read-back).  The
result is the SAME object graph the per-operator path builds lazily — `CoordMap`s with their strided / generated / union links,
`KernelMap`s with their derived tables — only that every buffer is a view into one of two arenas and is materialised as a torch
tensor when somebody asks for it (the native executor takes the pre-filled descriptors and never does).

`Lookahead` runs `plan_step` for the NEXT batch on a worker thread while the main thread enqueues the current step (the maps
depend on the input points only): ctypes releases the GIL for the duration of the native calls, so the two read-backs of the
plan are waited for off the step's critical path.
"""
import threading
import weakref

import numpy as np
import torch

from . import _lib as L
from . import sparse as SP

# ---- layout of the native tables (csrc/plan.hip) --------------------------------------------------------------------------------
C_B, C_NL, C_NFEAT, C_VS, C_FEATDIV, C_TOTAL, C_BACKWARD, C_SORT_MIN, C_PAIR_ROWS, C_PTS_THR, C_TARGETS, C_COORDS_IN, C_FEATS_IN, \
    C_PT_STRIDE, C_NECK, C_VS_HEAD, C_PROBE = range(17)
HDR = 16
H_S, H_NEED2, H_PRUNE, H_NMAPS, H_STRUCT, H_NALL, H_F0, H_TGT_PTS, H_TGT_SCENE, H_TGT_LEVEL, H_TGT_ORDER, H_TGT_SEG, H_NHEAD, H_BAD = range(14)
SETW, MAXSETS, MAPR, METAW, MAXLV = 8, 24, 64, 8, 8
S_COORDS, S_N, S_STRIDE, S_KEYS, S_VALS, S_CAP, S_PARENT, S_ROWS = range(8)
(MW_IN, MW_OUT, MW_K, MW_NIN, MW_NOUT, MW_NBR, MW_NBRT, MW_SORT, MW_SORTI, MW_SORTT, MW_SORTTI, MW_PI, MW_PO, MW_POS, MW_CNT, MW_TILES,
 MW_TPI, MW_TPO, MW_TPOS, MW_TCNT, MW_TTILES, MW_FLAGS) = range(22)
MW_DESC_F, MW_DESC_B = 24, 44

PROBE = False           # True: HIP-event brackets around every launch of the plans (fc_plan_probe_read; bench.py hbm_kernels)
PROBE_KINDS = ('tables_init', 'collate_insert', 'flags_scan', 'finalize_insert', 'gen_coords', 'kernel_maps', 'children_maps', 'fill',
               'transpose', 'row_masks', 'radix_argsort', 'permute', 'pair_lists', 'union_rows', 'head_arrays')
TRACE = None            # a list: (tag, perf_counter) marks of every plan (tools/hostprof.py --lookahead)
ENABLED = True          # False: the per-operator coordinate phase (sparse.py), kept as the cross-check (tests/test_gpu_plan.py)


def _mark(tag):
    if TRACE is not None:
        import time
        TRACE.append((tag, time.perf_counter()))


def _f64_bits(v):
    return int(np.array([v], dtype=np.float64).view(np.int64)[0])


def default_switches():
    """the native plan implements the DEFAULT routing switches of sparse.py; an A/B switch set to anything else takes the
    per-operator path"""
    return (SP.SORT_ROWS and not SP.SORT_DENSE and SP.SORT_DENSE_MAX_ROWS == 0 and SP.STRUCTURED_MAPS and not SP.PAIRS_DENSE
            and SP.WGRAD_PAIRS)


class _Arenas:
    """the device memory behind one plan: tensors are views of it, made on demand"""

    def __init__(self, a1, a2):
        self.a = [(a1.data_ptr(), a1.numel(), a1), (a2.data_ptr(), a2.numel(), a2)]

    def view(self, ptr, shape, dtype=torch.int32):
        n = 1
        for s in shape:
            n *= s
        nb = n * (4 if dtype != torch.int64 else 8)
        for base, size, t in self.a:
            if base <= ptr and ptr + nb <= base + size:
                o = ptr - base
                return t[o:o + nb].view(dtype).view(shape)
        raise RuntimeError('plan buffer outside its arenas')


class StepPlan:
    """sets / maps of one batch; `x` the input SparseTensor (level 0), `head_maps` the head's coordinate sets finest first (None when
    pruning bites or the union is not the generated set), `targets` the head's location arrays"""
    __slots__ = ('x', 'sets', 'maps', 'head_maps', 'prune_level', 'targets', 'arenas', 'out', 'backward', 'structured', 'counts')


class Planner:
    """per-detector state of the native plan: cfg words, pinned read-back buffers (one set per thread that plans)"""

    def __init__(self, det):
        self.det = det
        self._tls = threading.local()

    def _bufs(self, B, nl):
        t = self._tls
        key = (B, nl)
        if getattr(t, 'key', None) != key:
            S = 3 + nl
            t.key = key
            t.counts = torch.zeros(METAW * S + S * B + 64, dtype=torch.int32).pin_memory()
            t.cnt = torch.zeros(64 * (2 + 4 * nl) + MAXLV + nl * B + 1 + 64, dtype=torch.int32).pin_memory()
            t.out = np.zeros(L.lib().fc_plan_out_words(B, nl), dtype=np.int64)
            t.cfg = np.zeros(L.lib().fc_plan_cfg_words(), dtype=np.int64)
            t.scenes = np.zeros((max(B, 1), 3), dtype=np.int64)
        return t

    def applicable(self, points):
        det = self.det
        bb = det.backbone
        if not ENABLED or not default_switches() or getattr(bb.BLOCK, 'expansion', 1) == 4:
            return False
        p0 = points[0]
        if not (torch.is_tensor(p0) or hasattr(p0, 'voxelize_into')):
            return False
        if torch.is_tensor(p0) and not p0.is_cuda:
            return False
        return min(bb.n_outs, 4) >= 1 and len(points) <= 32767

    def run(self, points, training, want_targets, record_to=None, grad=None):
        """-> StepPlan.  Everything is enqueued on the CURRENT stream (the caller picks the coordinate stream).  record_to: the stream
        that will consume the buffers (default: what sparse.on_map_stream set); grad: torch.is_grad_enabled() of the CONSUMER (a
        worker thread has its own grad mode)."""
        det = self.det
        bb, nh = det.backbone, det.neck_with_head
        B, nl = len(points), min(bb.n_outs, 4)
        dev = points[0].device
        t = self._bufs(B, nl)
        cfg, out, scenes = t.cfg, t.out, t.scenes
        raw = all(torch.is_tensor(p) and p.dtype == torch.float32 and p.dim() == 2 and p.is_contiguous() and
                  p.shape[1] == points[0].shape[1] for p in points) and not det.spatial_sort
        total = sum(p.shape[0] for p in points)
        nfeat = points[0].shape[1] - 3
        cfg[:] = 0
        cfg[C_B], cfg[C_NL], cfg[C_NFEAT], cfg[C_TOTAL] = B, nl, nfeat, total
        cfg[C_VS], cfg[C_FEATDIV] = _f64_bits(float(det.voxel_size)), _f64_bits(255.0)
        cfg[C_BACKWARD] = 1 if (training and (torch.is_grad_enabled() if grad is None else grad)) else 0
        cfg[C_SORT_MIN], cfg[C_PAIR_ROWS] = SP.SORT_MIN_ROWS, SP.PAIR_CONV_ROWS
        cfg[C_PTS_THR] = nh.pts_threshold if nh.pts_threshold >= 0 else -1
        cfg[C_TARGETS] = 1 if want_targets else 0
        cfg[C_NECK] = 1
        cfg[C_VS_HEAD] = _f64_bits(float(nh.voxel_size))
        cfg[C_PROBE] = 1 if PROBE else 0
        keep = None
        if raw:
            cfg[C_PT_STRIDE] = points[0].shape[1]
            for b, p in enumerate(points):
                scenes[b, 0], scenes[b, 1], scenes[b, 2] = p.data_ptr(), p.shape[0], p.shape[1]
        else:
            coords, feats = det.voxelize(points)       # an augmenting pipeline / Z-order sort wrote the collate itself
            keep = (coords, feats)
            cfg[C_COORDS_IN], cfg[C_FEATS_IN] = coords.data_ptr(), feats.data_ptr()
        lib = L.lib()
        stream = L.stream()
        _mark('run0')
        a1 = torch.empty(L.query('fc_plan_stage1_bytes', total, B, nl, nfeat), dtype=torch.uint8, device=dev)
        rc = lib.fc_plan_levels(cfg.ctypes.data, scenes.ctypes.data, a1.data_ptr(), a1.numel(), out.ctypes.data, t.counts.data_ptr(), stream)
        _mark('levels')
        if rc:
            raise RuntimeError(f'fc_plan_levels failed: {rc}')
        if out[H_BAD]:
            raise ValueError('voxel coordinate outside [-32639, 32639] (or batch index outside [0, 32767]): the 64-bit voxel '
                             'hash keys hold 16 bits per field — check the input for outliers / non-finite points')
        need = int(lib.fc_plan_stage2_bytes(cfg.ctypes.data, out.ctypes.data, t.counts.data_ptr()))
        if need < 0:
            raise RuntimeError('fc_plan_stage2_bytes failed')
        a2 = torch.empty(need + 256, dtype=torch.uint8, device=dev)
        rc = lib.fc_plan_maps(cfg.ctypes.data, out.ctypes.data, t.counts.data_ptr(), a2.data_ptr(), a2.numel(), t.cnt.data_ptr(), stream)
        _mark('maps')
        if rc:
            raise RuntimeError(f'fc_plan_maps failed: {rc}')
        rec = record_to if record_to is not None else SP._record_to
        if rec is not None:
            for tns in (a1, a2) + (keep or ()):
                tns.record_stream(rec)
        sp = self._wrap(out.copy(), t.counts.numpy().copy(), _Arenas(a1, a2), B, nl, nfeat, bool(cfg[C_BACKWARD]), want_targets, dev)
        _mark('wrapped')
        return sp

    # ---- the object graph of the per-operator path over the plan's buffers -------------------------------------------------
    def _wrap(self, out, counts, ar, B, nl, nfeat, backward, want_targets, dev):
        S0, S = 3 + nl, int(out[H_S])
        sets = []
        scene_cnt = counts[METAW * S0:METAW * S0 + S0 * B].reshape(S0, B)
        for s in range(S):
            o = out[HDR + SETW * s:HDR + SETW * (s + 1)]
            n, stride = int(o[S_N]), int(o[S_STRIDE])
            cm = SP.CoordMap.__new__(SP.CoordMap)
            lazy = {'coords': (int(o[S_COORDS]), (n, 4), torch.int32)}
            if o[S_KEYS]:
                cap = int(o[S_CAP])
                lazy['_keys'] = (int(o[S_KEYS]), (cap,), torch.int64)
                lazy['_vals'] = (int(o[S_VALS]), (cap,), torch.int32)
            else:
                cm._keys = cm._vals = None
            cm.__dict__.update(_lazy=lazy, _arenas=ar, stride=stride, batch_size=B, n=n, _kmaps=weakref.WeakKeyDictionary(), _strided={},
                               _unions=weakref.WeakKeyDictionary(), _generated=None, _perm=None, _order=None, dense_hint=s >= S0, _grouped=True)
            if s < S0:
                cm._counts = [int(v) for v in scene_cnt[s]]
            else:
                par = sets[int(o[S_PARENT])]
                cm._counts = [8 * v for v in par._counts]
                cm._gen_parent = par
                par._generated = cm
            sets.append(cm)
        for s in range(S0 - 1):
            sets[s]._strided[2] = sets[s + 1]
        nm = int(out[H_NMAPS])
        mbase = HDR + SETW * MAXSETS
        maps = []
        for m in range(nm):
            o = out[mbase + MAPR * m:mbase + MAPR * (m + 1)]
            K, n_in, n_out = int(o[MW_K]), int(o[MW_NIN]), int(o[MW_NOUT])
            km = SP.KernelMap.__new__(SP.KernelMap)
            fl = int(o[MW_FLAGS])
            lazy = {'nbr': (int(o[MW_NBR]), (K, n_out), torch.int32)}
            d = dict(n_in=n_in, n_out=n_out, K=K, sort_rows=bool(fl & 1), use_pairs=bool(fl & 2), _arenas=ar, _desc={})
            conv = m >= 2
            if conv:
                if o[MW_PI]:
                    lazy['_pairs'] = [(int(o[MW_PI + w]), (K, n_out) if w < 3 else (K,), torch.int32) for w in range(4)]
                    if n_out <= SP.PAIR_CONV_ROWS:
                        d['_tiles'] = int(o[MW_TILES])
                else:
                    d['_pairs'] = None
                if o[MW_SORTI]:
                    lazy['_sorted'] = [(int(o[MW_SORT]), (K, n_out), torch.int32), (int(o[MW_SORTI]), (n_out,), torch.int32)]
                else:
                    d['_sorted'] = None
                if backward:
                    lazy['_nbr_t'] = (int(o[MW_NBRT]), (K, n_in), torch.int32)
                    if o[MW_TPI]:
                        lazy['_pairs_t'] = [(int(o[MW_TPI + w]), (K, n_in) if w < 3 else (K,), torch.int32) for w in range(4)]
                        d['_tiles_t'] = int(o[MW_TTILES])
                    else:
                        d['_pairs_t'] = None
                    if o[MW_SORTTI]:
                        lazy['_sorted_t'] = [(int(o[MW_SORTT]), (K, n_in), torch.int32), (int(o[MW_SORTTI]), (n_in,), torch.int32)]
                    else:
                        d['_sorted_t'] = None
                    d['_desc'][(True, True)] = o[MW_DESC_B:MW_DESC_B + 20]
                else:
                    d.update(_nbr_t=None, _pairs_t=None, _sorted_t=None)
                d['_desc'][(True, False)] = o[MW_DESC_F:MW_DESC_F + 20]
            else:
                d.update(_nbr_t=None, _pairs=None, _pairs_t=None, _sorted=None, _sorted_t=None)
                d['_desc'][(False, True)] = d['_desc'][(False, False)] = o[MW_DESC_F:MW_DESC_F + 20]
            d['_lazy'] = lazy
            km.__dict__.update(d)
            ks = {27: 3, 8: 2, 1: 1}[K]
            sets[int(o[MW_IN])]._kmaps.setdefault(sets[int(o[MW_OUT])], {})[ks] = km      # (weak key: sparse.CoordMap.__init__)
            maps.append(km)
        structured = bool(out[H_STRUCT])
        prune = int(out[H_PRUNE])
        head_maps = None
        if structured:
            for g in range(S0, S):
                i = nl - 2 - (g - S0)
                rows = ar.view(int(out[HDR + SETW * g + S_ROWS]), (sets[3 + i].n,))
                sets[3 + i]._unions[sets[g]] = (sets[g], rows, True)
            if prune < 0:
                head_maps = [sets[S0 - 1 + (nl - 1 - l)] if l < nl - 1 else sets[S0 - 1] for l in range(nl)]
        sp = StepPlan()
        n0 = sets[0].n
        F = ar.view(int(out[H_F0]), (n0, nfeat), torch.float32)
        sp.x = SP.SparseTensor(F, coordinate_map_key=sets[0])
        sp.sets, sp.maps, sp.head_maps, sp.prune_level, sp.arenas, sp.out, sp.backward, sp.structured, sp.counts = \
            sets, maps, head_maps, (prune if prune >= 0 else None), ar, out, backward, structured, counts
        sp.targets = None
        if want_targets and head_maps is not None and out[H_TGT_PTS]:
            n_all = int(out[H_NALL])
            sp.targets = dict(pts=ar.view(int(out[H_TGT_PTS]), (n_all, 3), torch.float32), scene=ar.view(int(out[H_TGT_SCENE]), (n_all,)),
                              level=ar.view(int(out[H_TGT_LEVEL]), (n_all,)), order=ar.view(int(out[H_TGT_ORDER]), (n_all,)),
                              seg_start=ar.view(int(out[H_TGT_SEG]), (nl * B + 1,)))
        return sp


def planner_of(det):
    p = det.__dict__.get('_planner')
    if p is None:
        p = det.__dict__['_planner'] = Planner(det)
    return p


class Lookahead:
    """The plan of the NEXT batch on a worker thread (the maps depend on the input points only — SingleStageSparse3DDetector.plan_maps):
    `submit(points, ...)` returns at once; `take(points, ...)` hands the StepPlan over if it was made for exactly these point tensors
    in this mode, else None (the caller plans in line).  The native calls release the GIL, so their two read-backs are waited for
    beside the main thread's enqueueing — off the step's critical path.  One plan in flight."""

    def __init__(self, det):
        import concurrent.futures
        self.det = det
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix='fc-plan')
        self.pending = {}                    # key -> (future, points): the current batch's plan is usually taken right after the
                                             # next one's is submitted, so two entries live side by side

    @staticmethod
    def _key(points, training, grad, want_targets):
        return (tuple((id(p), p.data_ptr() if torch.is_tensor(p) else 0, p.shape[0]) for p in points), bool(training), bool(grad),
                bool(want_targets), ENABLED)

    def submit(self, points, training, want_targets, side, main, wait_main):
        pl = planner_of(self.det)
        if not pl.applicable(points):
            return False
        grad = torch.is_grad_enabled()
        key = self._key(points, training, grad, want_targets)
        dev = points[0].device
        ev = None
        if wait_main:                         # the points were produced on the main stream (an upload, an on-device pipeline)
            ev = torch.cuda.Event()
            ev.record(main)

        def work():
            torch.cuda.set_device(dev)
            if ev is not None:
                side.wait_event(ev)
            with torch.cuda.stream(side), torch.no_grad():
                return pl.run(points, training, want_targets, record_to=main, grad=grad)
        _mark('submit')
        while len(self.pending) >= 2:         # plans nobody came for
            self._discard(next(iter(self.pending)))
        if key in self.pending:
            return True
        self.pending[key] = (self.pool.submit(work), points)
        return True

    def take(self, points, training, want_targets):
        if not self.pending:
            return None
        ent = self.pending.pop(self._key(points, training, torch.is_grad_enabled(), want_targets), None)
        if ent is None:
            return None
        _mark('take0')
        if not ent[0].done():
            import time
            t0 = time.perf_counter()
            ent[0].result()
            L.HOST_WAIT[0] += time.perf_counter() - t0
        sp = ent[0].result()                  # every kernel of the plan has completed (it ends with its own read-back)
        _mark('take1')
        return sp

    def _discard(self, key):
        fut, _ = self.pending.pop(key)
        try:
            fut.result()
        except Exception:                     # noqa: a plan nobody will use
            pass

    def drop(self):
        for key in list(self.pending):
            self._discard(key)


def probe_read():
    """[(kind name, compulsory bytes, ms)] of the plans run with PROBE since the last call (the device must have drained)"""
    import ctypes
    cap = 4096
    ms, by, kd = (ctypes.c_float * cap)(), (ctypes.c_double * cap)(), (ctypes.c_int * cap)()
    n = L.lib().fc_plan_probe_read(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(by, ctypes.c_void_p), ctypes.cast(kd, ctypes.c_void_p), cap)
    return [(PROBE_KINDS[kd[i]], float(by[i]), float(ms[i])) for i in range(n)]
