"""Sparse tensor + coordinate maps on the GPU (host-side mirror of ME.SparseTensor /
ME.CoordinateManager as FCAF3D uses them: single_stage_sparse.py:34-37, me_resnet.py,
fcaf3d_neck_with_head.py).  All coordinate work runs in libfcaf3d_hip.so (csrc/coords.hip).

Row-order rule: every coordinate set keeps its rows in order of first occurrence in the sequence
that produced it (ME's CPU rule) — deterministic, and identical to the
oracle's, so parity is checked row for row.
"""
import itertools
import os
import weakref

import numpy as np
import torch

from . import _lib as L


# ---- side stream for coordinate work --------------------------------------------------------------
# Hash / kernel-map construction depends only on the input coordinates, so it can run on its own HIP
# stream while the main stream is still busy with the previous step; its count read-backs then wait for
# microseconds of integer kernels instead of a full backward pass.  Tensors allocated under the side
# stream and consumed on the main stream are pinned with record_stream().
_side = {}
_record_to = None


def map_stream(device):
    """The coordinate stream of `device`.  Priority (r4): its kernels are microseconds long and the host waits for their counts, so
    they should overtake whatever the dependent chain has in flight — HIGH priority — but ONLY when the caller's stream is not a
    high-priority stream itself: two streams of one priority level can be multiplexed onto ONE hardware queue, and then the
    coordinate phase of the next step queues behind the whole backlog of this one (measured: bench.py's high-priority training
    stream + a high-priority coordinate stream fall into a 31 ms mode instead of 22.5 on some process layouts, always beside an
    RCCL communicator: 2 scenes per step through the averager 22 ms instead of 10).  FC_MAP_PRIO = -1 / 0 forces one."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    env = os.environ.get('FC_MAP_PRIO', 'auto')
    if env in ('-1', '0'):
        prio = int(env)
    else:
        prio = -1 if torch.cuda.current_stream(key).priority >= 0 else 0
    if (key, prio) not in _side:
        _side[(key, prio)] = torch.cuda.Stream(device=key, priority=prio)
    return _side[(key, prio)]


_plan_side = {}


def plan_stream(device):
    """the stream of the lookahead plan (plan.Lookahead): the coordinate stream itself.  r6 measured a stream of its own (FC_PLAN_STREAM=own:
    the main stream then never waits for a part of the NEXT batch's plan at the end of a coordinate phase) at 24.2 ms per 8-scene step
    against 20.9 on the coordinate stream, with identical kernel durations in the rocprofv3 trace — a FIFTH busy stream of the process
    beside main / head / weight-gradient / coordinate falls into the slow mode of profiles/r5_notes.md section 16 deterministically
    (profiles/r6_notes.md section 2), so the step keeps to four."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    if os.environ.get('FC_PLAN_STREAM') != 'own':
        return map_stream(key)
    if key not in _plan_side:
        _plan_side[key] = torch.cuda.Stream(device=key, priority=map_stream(key).priority)
    return _plan_side[key]


_inputs_ready = {}


def mark_inputs_ready(device=None):
    """Call on the stream that PRODUCED the points of the next batch (e.g. the on-device input pipeline) once they are
    written: the coordinate side stream waits for this event before it reads them.  Without a mark the inputs are
    taken to be resident already (bench / tests), and the side stream does not wait for the main stream at all —
    that is what lets the coordinate work of step i+1 overlap the backward pass of step i."""
    idx = None if device is None else torch.device(device).index
    dev = torch.cuda.current_device() if idx is None else idx
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dev))
    _inputs_ready[dev] = ev


class on_map_stream:
    """with on_map_stream(dev): ...   -> coordinate kernels go to the side stream; main waits on exit."""

    def __init__(self, device, inputs_resident=False):
        self.side = map_stream(device)
        self.dev = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        self.inputs_resident = inputs_resident

    def __enter__(self):
        global _record_to
        ev = _inputs_ready.pop(self.dev, None)
        self.main = torch.cuda.current_stream()
        if ev is not None:
            self.side.wait_event(ev)           # the producer of the points has finished writing them
        elif not self.inputs_resident:
            # nobody marked the inputs and the caller did not declare them resident: whatever produced them was enqueued
            # on the main stream (a non_blocking upload, a custom pipeline) — wait for it.  Skipping this wait is an
            # explicit opt-in (SingleStageSparse3DDetector.inputs_resident; bench.py: scenes sit in HBM before the step)
            self.side.wait_stream(self.main)
        self.prev = _record_to
        _record_to = self.main
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        global _record_to
        self.ctx.__exit__(*a)
        _record_to = self.prev
        self.main.wait_stream(self.side)


def _rec(*tensors):
    if _record_to is not None:
        for t in tensors:
            if t is not None:
                t.record_stream(_record_to)
    return tensors[0] if len(tensors) == 1 else tensors


def _next_pow2(n):
    p = 2
    while p < n:
        p *= 2
    return p


SORT_ROWS = True      # process conv rows in occupancy-mask order on sparse 27-offset maps
SORT_MIN_ROWS = int(os.environ.get('FC_SORT_MIN_ROWS', '8192'))
SORT_DENSE = os.environ.get('FC_SORT_DENSE', '0') != '0'     # mask-sorted rows also on the generated / union (neck) maps
# ... r3 A/B: only on the dense maps with at most this many rows (the 55k-row neck level issues 1.14x its useful MFMA work
# in natural order and 1.02x in mask order, and its argsort is cheap — unlike the 441k-row level's)
SORT_DENSE_MAX_ROWS = int(os.environ.get('FC_SORT_DENSE_MAX_ROWS', '0'))
STRUCTURED_MAPS = os.environ.get('FC_STRUCTURED_MAPS', '1') != '0'   # generated sets: maps by index arithmetic (r3), A/B switch
PAIRS_DENSE = os.environ.get('FC_PAIRS_DENSE', '0') != '0'   # pair lists (exact work) also on small dense maps (77 % occupied at 6.9k rows)
WGRAD_PAIRS = os.environ.get('FC_WGRAD_PAIRS', '1') != '0'    # weight gradients reduce over exact pair lists there
# ... and with at most this many result rows the convolution itself runs per offset over the pair lists
# (r6, three-product kernels: 8 192 — the 14.7k-row level on its mask-sorted table — 462.9 scenes/s against 456.1 at 16 384 and 452.5 at
# 2 048, same box; the six-product kernels of r3-r5 were level at 16 384 / 8 192: profiles/r5_notes.md section 7)
PAIR_CONV_ROWS = int(os.environ.get('FC_PAIR_CONV_ROWS', '8192'))
_offs_cache = {}


def kernel_offsets(kernel_size, tensor_stride, device):
    """(K,3) int32, x fastest; centred for odd kernels, {0..k-1} for even (Appendix A.3)."""
    key = (kernel_size, tensor_stride, str(device))
    if key not in _offs_cache:
        _offs_cache[key] = _kernel_offsets(kernel_size, tensor_stride, device)
    return _offs_cache[key]


def _kernel_offsets(kernel_size, tensor_stride, device):
    if kernel_size % 2 == 1:
        r = [i - kernel_size // 2 for i in range(kernel_size)]
    else:
        r = list(range(kernel_size))
    offs = [(dx * tensor_stride, dy * tensor_stride, dz * tensor_stride)
            for dz, dy, dx in itertools.product(r, r, r)]
    return torch.tensor(offs, dtype=torch.int32, device=device)


class _Lazy:
    """attributes of a plan-made object (plan.py) that are views of the plan's arenas: the tensor is made when first asked for"""

    def __getattr__(self, name):
        lz = self.__dict__.get('_lazy')
        if lz is not None and name in lz:
            spec = lz.pop(name)
            ar = self.__dict__['_arenas']
            v = tuple(ar.view(*t) for t in spec) if isinstance(spec, list) else ar.view(*spec)
            self.__dict__[name] = v
            return v
        raise AttributeError(name)


class KernelMap(_Lazy):
    """nbr (K, n_out) int32 plus, lazily, its transpose for the backward-data pass."""

    def __init__(self, nbr, n_in, n_out):
        self.nbr = _rec(nbr)
        self.n_in = n_in
        self.n_out = n_out
        self.K = nbr.shape[0]
        self._nbr_t = None
        self._sorted = None
        self._sorted_t = None
        self.sort_rows = False          # set by CoordMap.kernel_map for sparse-ish 27-offset maps
        self.use_pairs = False          # ditto: the weight-gradient pass walks exact pair lists
        self._pairs = None
        self._pairs_t = None

    @property
    def nbr_t(self):
        if self._nbr_t is None:
            t = torch.empty((self.K, self.n_in), dtype=torch.int32, device=self.nbr.device)
            L.call('fc_kernel_map_transpose', L.ptr(self.nbr), self.n_out, self.n_in, self.K, L.ptr(t), L.stream())
            self._nbr_t = _rec(t)
        return self._nbr_t

    def n_pairs(self):
        return int((self.nbr >= 0).sum().item())

    # ---- occupancy-mask row order (skips empty (tile, offset) work) ------------------------
    @staticmethod
    def _sort_table(nbr, n_rows, K):
        masks = torch.empty(n_rows, dtype=torch.int32, device=nbr.device)
        L.call('fc_nbr_row_masks', L.ptr(nbr), n_rows, K, L.ptr(masks), L.stream())
        order = torch.argsort(masks).to(torch.int32)
        tab = torch.empty_like(nbr)
        L.call('fc_permute_nbr', L.ptr(nbr), L.ptr(order), n_rows, K, L.ptr(tab), L.stream())
        return _rec(tab, order)

    @staticmethod
    def _pair_lists(nbr, n_rows, K):
        dev = nbr.device
        pi = torch.empty_like(nbr)
        po = torch.empty_like(nbr)
        pos = torch.empty_like(nbr)
        cnt = torch.empty(K, dtype=torch.int32, device=dev)
        ws = L.workspace(L.query('fc_kernel_map_pairs_ws_bytes', n_rows, K), dev)
        L.call('fc_kernel_map_pairs', L.ptr(nbr), n_rows, K, L.ptr(pi), L.ptr(po), L.ptr(pos), L.ptr(cnt),
               L.ptr(ws), ws.numel(), L.stream())
        return _rec(pi, po, pos, cnt)

    @staticmethod
    def _live_tiles(cnt):
        """sum_k ceil(cnt[k] / 128): the non-empty (offset, 128-row tile) workgroups of the per-offset convolution.  One small
        read-back per map, on the coordinate stream (where the other data-dependent sizes are read too): the launch then
        holds live workgroups only (fc_conv_fwd_pairs_tiles; +20...35 % on the deep levels, r2)."""
        return int(((cnt.cpu().numpy().astype(np.int64) + 127) // 128).sum())      # ONE device op: the 27-int read-back

    def pairs(self):
        """(pair_in, pair_out, pair_pos, cnt): per offset, the (input row, output row) pairs in ascending output row,
        and where each output row sits in each list (ME's in_maps / out_maps)."""
        if self._pairs is None:
            self._pairs = self._pair_lists(self.nbr, self.n_out, self.K)
        return self._pairs

    def pair_tiles(self, transposed=False):
        """live tile count of pairs() / pairs_t() for the per-offset convolution (lazy, cached)"""
        key = '_tiles_t' if transposed else '_tiles'
        v = getattr(self, key, None)
        if v is None:
            v = self._live_tiles((self.pairs_t() if transposed else self.pairs())[3])
            setattr(self, key, v)
        return v

    def pairs_t(self):
        """the same for the transposed table (backward-data pass): lists ascending in the INPUT row."""
        if self._pairs_t is None:
            self._pairs_t = self._pair_lists(self.nbr_t, self.n_in, self.K)
        return self._pairs_t

    def prefetch(self, backward=True):
        """Build now (i.e. on the coordinate side stream, SingleStageSparse3DDetector.plan_maps) every derived table
        the MFMA convolutions on this map will ask for, instead of lazily on the main stream in the middle of the
        convolution sequence: mask-sorted tables (an argsort each), pair lists, the transposed table."""
        if self.K != 27:
            return
        if self.use_pairs and self.n_out <= PAIR_CONV_ROWS:
            self.pair_tiles()
        else:
            self.sorted_fwd()
        if backward:
            self.nbr_t
            if self.use_pairs:
                self.pairs()
            if self.use_pairs and self.n_in <= PAIR_CONV_ROWS:
                self.pair_tiles(transposed=True)
            else:
                self.sorted_bwd()

    def desc(self, conv=True, backward=True):
        """int64 descriptor of this map for the native executor (csrc/exec.hip, MAPW words): sizes, the tables the
        convolution routes of functional._SparseConv would pick (built now if they are not yet: the same lazy tables, so a
        planned step finds them prefetched) and the route bits.  conv=False: the plain table only (stem, pooling)."""
        key = (conv, backward)
        d = self._desc.get(key) if hasattr(self, '_desc') else None
        if d is not None:
            return d
        if not hasattr(self, '_desc'):
            self._desc = {}
        d = np.zeros(20, dtype=np.int64)
        d[0], d[1], d[2], d[3] = self.n_in, self.n_out, self.K, self.nbr.data_ptr()
        if conv:
            flags = 0
            if self.use_pairs and self.n_out <= PAIR_CONV_ROWS:
                pi, po, pos, cnt = self.pairs()
                d[9:14] = (pi.data_ptr(), po.data_ptr(), pos.data_ptr(), cnt.data_ptr(), self.pair_tiles())
                flags |= 1
            else:
                tab, idx = self.sorted_fwd()
                d[5], d[6] = tab.data_ptr(), (idx.data_ptr() if idx is not None else 0)
            if backward:
                d[4] = self.nbr_t.data_ptr()
                if self.use_pairs:
                    pi, po, pos, cnt = self.pairs()
                    d[9:13] = (pi.data_ptr(), po.data_ptr(), pos.data_ptr(), cnt.data_ptr())
                    flags |= 4
                if self.use_pairs and self.n_in <= PAIR_CONV_ROWS:
                    pi, po, pos, cnt = self.pairs_t()
                    d[14:19] = (pi.data_ptr(), po.data_ptr(), pos.data_ptr(), cnt.data_ptr(), self.pair_tiles(transposed=True))
                    flags |= 2
                else:
                    tab, idx = self.sorted_bwd()
                    d[7], d[8] = tab.data_ptr(), (idx.data_ptr() if idx is not None else 0)
            d[19] = flags
        self._desc[key] = d
        return d

    def sorted_fwd(self):
        """(nbr permuted into mask order, order) for the forward / weight-gradient pass, or (nbr, None)."""
        if not self.sort_rows:
            return self.nbr, None
        if self._sorted is None:
            self._sorted = self._sort_table(self.nbr, self.n_out, self.K)
        return self._sorted

    def sorted_bwd(self):
        """the same for the transposed table of the backward-data pass."""
        if not self.sort_rows:
            return self.nbr_t, None
        if self._sorted_t is None:
            self._sorted_t = self._sort_table(self.nbr_t, self.n_in, self.K)
        return self._sorted_t


class CoordMap(_Lazy):
    """One coordinate set: coords (N,4) int32 [b,x,y,z], tensor stride, voxel hash, cached maps."""

    def __init__(self, coords, stride, keys, vals, batch_size):
        self.coords = _rec(coords)
        self._keys, self._vals = (_rec(keys, vals) if keys is not None else (None, None))
        self.stride = stride
        self.batch_size = batch_size
        self.n = coords.shape[0]
        # caches keyed by the OTHER set — weakly, and holding nothing that leads back here: a set, its kernel maps and the arenas they
        # view must die by reference count when the step lets go of them.  (r6: `_kmaps[(id(out), k)]` with `km._out_map = out` to keep
        # the id unique, parent <-> generated child and `_unions[id(other)] = (self, ...)` were cycles; the cyclic collector got to
        # them generations late — 0.66 GB of arenas per step waiting for it, +37 MB per step for good over 1 000 steps.)
        self._kmaps = weakref.WeakKeyDictionary()       # out_map -> {kernel_size: KernelMap}
        self._strided = {}
        self._unions = weakref.WeakKeyDictionary()      # other -> (union set | None = this set, rows, swapped)
        self._generated = None
        self._perm = None
        self._counts = None
        self._order = None
        self.dense_hint = False         # True for generated children sets and their unions
        self._gen_parent = None         # generated children set: the set it was generated from (rows 8i + k); held weakly

    @property
    def _gen_parent(self):
        r = self.__dict__.get('_gen_parent_ref')
        return r() if r is not None else None

    @_gen_parent.setter
    def _gen_parent(self, par):
        self.__dict__['_gen_parent_ref'] = weakref.ref(par) if par is not None else None

    # ---- voxel hash: built on demand for generated sets (their maps come from the parent level, see generate()) ----
    def _ensure_table(self):
        if self._keys is None:
            n = self.n
            cap = _next_pow2(max(2 * n, 2))
            dev = self.coords.device
            keys = torch.empty(cap, dtype=torch.int64, device=dev)
            vals = torch.empty(cap, dtype=torch.int32, device=dev)
            scratch = torch.empty((max(n, 1), 4), dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = L.workspace(L.query('fc_hash_unique_ws_bytes', n), dev)
            L.call('fc_hash_unique', L.ptr(self.coords), n, 1, L.ptr(keys), L.ptr(vals), cap, L.ptr(scratch), None, None,
                   L.ptr(cnt), L.ptr(ws), ws.numel(), L.stream())          # rows are unique already: table only
            self._keys, self._vals = _rec(keys, vals)

    @property
    def keys(self):
        self._ensure_table()
        return self._keys

    @property
    def vals(self):
        self._ensure_table()
        return self._vals

    @property
    def cap(self):
        return self.keys.numel()

    # ---- construction -------------------------------------------------------------------------
    @staticmethod
    def from_coords(coords, stride, batch_size, q=1, want_first=False, want_inverse=False, expect_n=None):
        """Unique rows of floor(coords/q)*q in order of first occurrence + hash of the result.
        expect_n: the caller knows the number of unique rows (no host read-back)."""
        assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        coords = coords.contiguous()
        dev = coords.device
        n = coords.shape[0]
        cap = _next_pow2(max(2 * n, 2))
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        vals = torch.empty(cap, dtype=torch.int32, device=dev)
        out = torch.empty((n, 4), dtype=torch.int32, device=dev)
        first = torch.empty(n, dtype=torch.int32, device=dev) if want_first else None
        inv = torch.empty(n, dtype=torch.int32, device=dev) if want_inverse else None
        cnt = torch.zeros(1, dtype=torch.int32, device=dev)
        wsb = L.query('fc_hash_unique_ws_bytes', n)
        ws = L.workspace(wsb, dev)
        L.call('fc_hash_unique', L.ptr(coords), n, q, L.ptr(keys), L.ptr(vals), cap, L.ptr(out), L.ptr(first),
               L.ptr(inv), L.ptr(cnt), L.ptr(ws), ws.numel(), L.stream())
        m = int(cnt.item()) if expect_n is None else expect_n     # the one host read-back of this op
        if m < 0:
            raise ValueError('voxel coordinate outside [-32639, 32639] (or batch index outside [0, 32767]): the 64-bit voxel '
                             'hash keys hold 16 bits per field — check the input for outliers / non-finite points')
        cm = CoordMap(out[:m] if m != n else out, stride, keys, vals, batch_size)
        _rec(first, inv)
        return cm, (first[:m] if want_first else None), inv

    def strided(self, s):
        """Output map of a stride-s conv / pooling (cached: k3s2 conv and k1s2 downsample share it)."""
        if s == 1:
            return self
        if s not in self._strided:
            cm, _, _ = CoordMap.from_coords(self.coords, self.stride * s, self.batch_size, q=self.stride * s)
            self._strided[s] = cm
        return self._strided[s]

    def generate(self):
        """Children set of MinkowskiGenerativeConvolutionTranspose(k2,s2): row 8i+k.  No hash is built for it (r3): its
        k3 kernel map and the rows of the backbone level inside it follow from THIS level by index arithmetic
        (fc_kernel_map_children / fc_child_rows); the table appears on demand (interpolation, a union that adds rows)."""
        if self._generated is None:
            assert self.stride % 2 == 0
            half = self.stride // 2
            out = torch.empty((self.n * 8, 4), dtype=torch.int32, device=self.coords.device)
            L.call('fc_gen_coords', L.ptr(self.coords), self.n, half, L.ptr(out), L.stream())
            # children of a unique stride-T set are unique: 8n rows, no read-back needed
            g = CoordMap(out, half, None, None, self.batch_size)
            g.dense_hint = True
            g._gen_parent = self
            self._generated = g
        return self._generated

    def kernel_map(self, out_map, kernel_size):
        """KernelMap from this (input) set to `out_map`, offsets in units of this set's stride."""
        per_out = self._kmaps.get(out_map)
        km = per_out.get(kernel_size) if per_out is not None else None
        if km is None:
            offs = kernel_offsets(kernel_size, self.stride, self.coords.device)
            K = offs.shape[0]
            nbr = torch.empty((K, out_map.n), dtype=torch.int32, device=self.coords.device)
            if out_map is self and kernel_size == 3 and self._gen_parent is not None and STRUCTURED_MAPS:
                par = self._gen_parent                       # generated set: from the parent level's own k3 table
                L.call('fc_kernel_map_children', L.ptr(par.kernel_map(par, 3).nbr), par.n, L.ptr(nbr), L.stream())
            else:
                L.call('fc_kernel_map', L.ptr(out_map.coords), out_map.n, L.ptr(self.keys), L.ptr(self.vals), self.cap,
                       L.ptr(offs), K, L.ptr(nbr), L.stream())
            km = KernelMap(nbr, self.n, out_map.n)
            # generated / union sets are ~94 % dense (2x2x2 blocks): nothing to skip there
            # ... and below ~8k rows the masks do not group well enough to pay for themselves (tools/convbench.py)
            # (r2: the generated / union sets are 77 % / 88 % / 94 % occupied; mask order would issue 1.10x / 1.02x / 1.00x the
            # useful MFMA work instead of 1.30x / 1.14x / 1.07x and the isolated kernels gain 3...10 %, but in the full step
            # the extra argsort + permuted tables + scattered output rows cancel it exactly: 233.5 vs 233.5 scenes/s on the
            # same box.  FC_SORT_DENSE=1 turns it on.)
            dense = self.dense_hint and out_map.dense_hint
            km.sort_rows = (SORT_ROWS and K == 27 and out_map.n >= SORT_MIN_ROWS
                            and (SORT_DENSE or not dense or out_map.n <= SORT_DENSE_MAX_ROWS))
            km.use_pairs = WGRAD_PAIRS and K == 27 and (not dense or (PAIRS_DENSE and out_map.n <= PAIR_CONV_ROWS))
            self._kmaps.setdefault(out_map, {})[kernel_size] = km
        return km

    def union(self, other):
        """Union map for `self + other`.  Returns (map, rows, swapped):
          swapped False: rows of self first, then other's new voxels; rows[i] = union row of other's row i;
          swapped True : every voxel of self already lies in `other` (the usual case: backbone level inside the
                         generated children set) -> the union IS other's set and other's map (with its cached
                         kernel maps) is reused; rows[i] = row in `other` of self's row i."""
        assert self.stride == other.stride
        ent = self._unions.get(other)
        if ent is not None:
            return (self if ent[0] is None else ent[0]), ent[1], ent[2]
        dev = self.coords.device
        if other._gen_parent is not None and STRUCTURED_MAPS:
            # `other` is a generated children set: where each of MY voxels sits in it follows from its parent level's hash
            # (8 * parent row + child bits); if all of them are inside — the usual case, the backbone level inside the
            # generated set — the union IS `other`, with one probe pass and one count read-back and no hash of `other`
            par = other._gen_parent
            rows = torch.empty(self.n, dtype=torch.int32, device=dev)
            found = torch.zeros(1, dtype=torch.int32, device=dev)
            L.call('fc_child_rows', L.ptr(self.coords), self.n, L.ptr(par.keys), L.ptr(par.vals), par.cap, self.stride,
                   L.ptr(rows), L.ptr(found), L.stream())
            if int(found.item()) == self.n:
                _rec(rows)
                self._unions[other] = (other, rows, True)
                return other, rows, True

        def probe(q, table):
            rows = torch.empty(q.n, dtype=torch.int32, device=dev)
            newc = torch.empty((q.n, 4), dtype=torch.int32, device=dev)
            cnt = torch.zeros(1, dtype=torch.int32, device=dev)
            ws = L.workspace(L.query('fc_union_map_ws_bytes', q.n), dev)
            L.call('fc_union_map', L.ptr(q.coords), q.n, L.ptr(table.keys), L.ptr(table.vals), table.cap, table.n,
                   L.ptr(rows), L.ptr(newc), L.ptr(cnt), L.ptr(ws), ws.numel(), L.stream())
            return rows, newc, cnt
        row_b, newc, cnt = probe(other, self)
        n_new = int(cnt.item())
        swapped = False
        if n_new == 0:
            cm, rows = self, row_b
        elif n_new == other.n - self.n:
            rows, _, _ = probe(self, other)                # all found: no count read-back needed
            cm, swapped = other, True
        else:
            coords = torch.cat([self.coords, newc[:n_new]])
            cm, _, _ = CoordMap.from_coords(coords, self.stride, self.batch_size, expect_n=coords.shape[0])
            cm.dense_hint = other.dense_hint
            rows = row_b
        _rec(rows)
        self._unions[other] = (None if cm is self else cm, rows, swapped)
        return cm, rows, swapped

    def pruned(self, kept):
        """Map of the kept rows (int32 ascending row indices), order preserved."""
        out = torch.empty((kept.numel(), 4), dtype=torch.int32, device=self.coords.device)
        L.call('fc_gather_coords', L.ptr(self.coords), L.ptr(kept), kept.numel(), L.ptr(out), L.stream())
        cm, _, _ = CoordMap.from_coords(out, self.stride, self.batch_size, expect_n=kept.numel())
        return cm

    # ---- per-scene decomposition --------------------------------------------------------------
    def _decompose(self):
        """device side of the per-scene decomposition: row order grouped by scene + per-scene counts (no read-back)"""
        if getattr(self, '_order', None) is None:
            if getattr(self, '_grouped', False) and self._counts is not None:
                # a set of the native plan: rows are grouped by scene already and the counts are known on the host
                dev = self.coords.device
                self._order = torch.arange(self.n, device=dev)
                self._counts_dev = L.upload(np.asarray(self._counts, dtype=np.int64), dev)
                _rec(self._order, self._counts_dev)
                return
            b = self.coords[:, 0].long()
            order = torch.argsort(b, stable=True)            # rows grouped by scene, ascending inside
            counts_dev = torch.zeros(self.batch_size, dtype=torch.int64, device=b.device).scatter_add_(0, b, torch.ones_like(b))   # (bincount syncs)
            self._order, self._counts_dev = _rec(order, counts_dev)

    @property
    def decomposition_permutations(self):
        if self._perm is None:
            self._decompose()
            self._perm = list(self._order.split(self.scene_counts))
        return self._perm

    @property
    def scene_counts(self):
        if self._counts is None:
            self._decompose()
            self._counts = self._counts_dev.cpu().tolist()                          # one read-back per map
        return self._counts


def compact_mask(mask, expect_n=None):
    """bool/uint8 mask (n,) -> int32 indices of set rows, ascending (ballot + prefix-sum kernel).  expect_n: the caller knows how
    many rows are set (a per-scene top-k keeps exactly k): no read-back of the count — the host does not wait for the mask"""
    flags = mask.to(torch.uint8).contiguous()
    n = flags.numel()
    dev = flags.device
    pos = torch.empty(n, dtype=torch.int32, device=dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = L.workspace(4 * (n // 1024 + 1), dev)
    L.call('fc_scan_flags', L.ptr(flags), n, L.ptr(pos), L.ptr(cnt), L.ptr(ws), ws.numel(), L.stream())
    m = int(cnt.item()) if expect_n is None else int(expect_n)
    kept = torch.empty(m, dtype=torch.int32, device=dev)
    L.call('fc_compact_rows', L.ptr(flags), L.ptr(pos), n, L.ptr(kept), m, L.stream())
    return kept


class SparseTensor:
    """features (N,C) fp32 on a CoordMap.  Attribute surface follows ME.SparseTensor as used by the
    reference: .F/.features, .C/.coordinates, .tensor_stride, .coordinate_map_key (= the CoordMap),
    .decomposition_permutations, .decomposed_coordinates, .features_at_coordinates()."""

    def __init__(self, features, coordinates=None, coordinate_map_key=None, coordinate_manager=None,
                 batch_size=None):
        if coordinate_map_key is not None:
            self.cmap = coordinate_map_key
            self.F = features
        else:
            assert coordinates is not None
            if batch_size is None:
                batch_size = int(coordinates[:, 0].max().item()) + 1 if coordinates.numel() else 0
            cm, first, _ = CoordMap.from_coords(coordinates.to(torch.int32), 1, batch_size, want_first=True)
            from .functional import gather_rows
            self.cmap = cm
            self.F = gather_rows(features.contiguous(), first)   # first occurrence wins (A.2)
        assert self.F.shape[0] == self.cmap.n

    @property
    def features(self):
        return self.F

    @property
    def C(self):
        return self.cmap.coords

    coordinates = C

    @property
    def coordinate_map_key(self):
        return self.cmap

    @property
    def coordinate_manager(self):
        return None

    @property
    def tensor_stride(self):
        return self.cmap.stride

    @property
    def decomposition_permutations(self):
        return self.cmap.decomposition_permutations

    @property
    def decomposed_coordinates(self):
        return [self.cmap.coords[p, 1:] for p in self.decomposition_permutations]

    def features_at_coordinates(self, query):
        """Trilinear interpolation at integer-valued query coordinates (M,4) [b,x,y,z]."""
        q = query.to(torch.int32).contiguous()
        out = torch.empty((q.shape[0], self.F.shape[1]), dtype=torch.float32, device=q.device)
        F = self.F.detach().contiguous()
        L.call('fc_interp', L.ptr(q), q.shape[0], L.ptr(self.cmap.keys), L.ptr(self.cmap.vals), self.cmap.cap,
               L.ptr(F), F.shape[1], self.cmap.stride, L.ptr(out), L.stream())
        return out

    def __add__(self, other):
        from .functional import union_add
        return union_add(self, other)

    def __len__(self):
        return self.cmap.n
