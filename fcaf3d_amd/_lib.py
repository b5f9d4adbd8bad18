"""ctypes binding of libfcaf3d_hip.so.

The prototypes are parsed from include/fcaf3d_hip.h (single source of truth), so an argument-type
mismatch between Python and the C ABI cannot creep in.  There is NO fallback: if the shared library
is missing the product path raises.
"""
import ctypes
import os
import re
import threading
import time

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('FC_LIB') or os.path.join(_HERE, 'libfcaf3d_hip.so')      # FC_LIB: an A/B build of the same library
HEADER = os.path.join(os.path.dirname(_HERE), 'include', 'fcaf3d_hip.h')

_SCALARS = {'int': ctypes.c_int, 'int64_t': ctypes.c_int64, 'float': ctypes.c_float,
            'hipStream_t': ctypes.c_void_p}


def parse_header(path=HEADER):
    """-> {name: (restype, [(argname, ctype)])}"""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    protos = {}
    for m in re.finditer(r'\b(int64_t|int)\s+(fc_\w+)\s*\(([^)]*)\)\s*;', txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argl = []
        for a in args.split(','):
            a = ' '.join(a.split())
            if not a or a == 'void':
                continue
            if '*' in a:
                argl.append((a.split('*')[-1].strip(), ctypes.c_void_p))
            else:
                ty, nm = a.rsplit(' ', 1)
                argl.append((nm, _SCALARS[ty.replace('const ', '')]))
        protos[name] = (_SCALARS[ret], argl)
    return protos


ABI_VERSION = 7          # include/fcaf3d_hip.h FC_ABI_VERSION
_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with `python -m fcaf3d_amd.build` '
                '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
        l = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (ret, args) in _protos.items():
            fn = getattr(l, name)            # AttributeError if the library lacks a declared symbol
            fn.restype = ret
            fn.argtypes = [t for _, t in args]
        if l.fc_abi_version() != ABI_VERSION:
            raise RuntimeError(f'{LIB_PATH} speaks ABI version {l.fc_abi_version()}, this host code was written for {ABI_VERSION}: rebuild '
                               'with `python -m fcaf3d_amd.build`')
        _lib = l
        if os.environ.get('FC_PRIO_OFF') or os.environ.get('FC_PRIO_MODE'):     # A/B switches of the MFMA-block wave priority
            l.fc_debug_set_prio(-1 if os.environ.get('FC_PRIO_OFF') else int(os.environ['FC_PRIO_MODE']))      # (conv.hip: g_fc_prio)
    return _lib


def ptr(t):
    if t is None:
        return None
    assert t.is_contiguous(), 'C ABI takes contiguous buffers'
    return t.data_ptr()


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream():
    """raw hipStream_t of torch's current stream on the current device (fast path: no Stream object)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


_fns = {}


def call(name, *args):
    """Invoke an `int`-returning entry point; raises on a non-zero status."""
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        kind = 'invalid argument' if rc == -1 else 'workspace too small' if rc == -2 else f'hipError {rc}'
        raise RuntimeError(f'{name} failed: {kind}')


_query_cache = {}


def query(name, *args):
    """Invoke an `int64_t`-returning size query (pure functions of their arguments: memoised)."""
    key = (name,) + args
    v = _query_cache.get(key)
    if v is None:
        if len(_query_cache) > 65536:
            _query_cache.clear()
        v = _query_cache[key] = int(getattr(lib(), name)(*args))
    return v


_ws_cache = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per (device, stream) — the library never allocates."""
    key = (device, stream())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


# seconds the host spent BLOCKED on the device in places where it waits by design (the run-ahead bound of runner.TrainStep, a
# staging buffer of the ring below still in use, the lookahead plan not ready): [total].  bench.py reports enqueue time minus this
# as the host's own work per step.
HOST_WAIT = [0.0]


class _PinnedRing:
    """Small host -> device uploads without a pinned allocation per call (hipHostMalloc costs ~0.1 ms) and without
    blocking the host (a pageable copy waits for the queue to drain): a ring of grow-only pinned staging buffers, each
    guarded by the event of the copy that last used it."""

    def __init__(self, n=16):
        self.bufs, self.evts, self.i = [None] * n, [None] * n, 0
        self.lock = threading.Lock()             # the main thread and the lookahead plan thread both upload: one slot each

    def upload(self, arr, device):
        """arr: numpy array (any shape, a dtype torch knows) -> device tensor, copied asynchronously on the current stream"""
        with self.lock:
            i = self.i
            self.i = (i + 1) % len(self.bufs)
        if self.evts[i] is not None and not self.evts[i].query():
            t0 = time.perf_counter()
            self.evts[i].synchronize()
            HOST_WAIT[0] += time.perf_counter() - t0
        nb = max(int(arr.nbytes), 1)
        if self.bufs[i] is None or self.bufs[i].numel() < nb:
            self.bufs[i] = torch.empty(max(nb, 4096), dtype=torch.uint8).pin_memory()
        src = torch.from_numpy(arr)
        stage = self.bufs[i][:arr.nbytes].view(src.dtype).reshape(src.shape)
        stage.copy_(src)
        out = stage.to(device, non_blocking=True)
        e = torch.cuda.Event()
        e.record(torch.cuda.current_stream(out.device))          # the stream of the TARGET device carried the copy
        self.evts[i] = e
        return out


_ring = _PinnedRing()


def upload(arr, device):
    return _ring.upload(arr, device)
