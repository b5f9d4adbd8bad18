"""Builds libfcaf3d_hip.so (gfx950) in-tree with hipcc.  No GPU needed (cross-compiles)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libfcaf3d_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-Wno-unused-result']
# r6: no SLP vectorisation in the MFMA kernels' files — the vectoriser packs the split arithmetic of adjacent channels into v_pk_mul_f32 /
# v_pk_fma_f32 / v_pk_add_f32, and beside MFMAs a packed fp32 instruction costs more than the two scalar ones it replaces
# (MI355X_MICROARCH.md "price of one filler beside MFMAs"): 454.9 / 453.8 -> 462.5 / 463.2 scenes/s (ABAB, one box), conv operator 93.2 -> 89.7 us
FILE_FLAGS = {'conv.hip': ['-fno-slp-vectorize'], 'norm.hip': ['-fno-slp-vectorize']}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def source_hash():
    """sha256[:16] over the kernel sources (csrc/*.hip, csrc/*.h, in name order) — recorded with every profile under profiles/ and
    in bench.py's line, so that a committed kernel table can be told from one taken on other kernels
    (tests/test_host_logic.py::test_r5_profiles_were_taken_on_the_committed_kernels)"""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), 'rb').read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    objs, jobs = [], []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s[:-4] + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + FILE_FLAGS.get(s, []) + ['-c', src, '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


def build_tools(verbose=True):
    """tools/nbench (native conv micro-benchmark) and, with FC_TRACE kernels, tools/nbench_trace — diagnostics only."""
    root = os.path.dirname(HERE)
    tools = os.path.join(root, 'tools')
    inc = os.path.join(root, 'include')

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    build(verbose=verbose)
    run([HIPCC, '-O2', '-std=c++17', '--offload-arch=gfx950', os.path.join(tools, 'nbench.cpp'), '-I' + inc, '-L' + HERE,
         '-lfcaf3d_hip', '-ldl', '-Wl,-rpath,$ORIGIN/../fcaf3d_amd', '-o', os.path.join(tools, 'nbench')])
    tdir = os.path.join(tools, 'trace')
    os.makedirs(tdir, exist_ok=True)
    objs = []
    for s in _sources():
        obj = os.path.join(tdir, s[:-4] + '.o')
        objs.append(obj)
        src = os.path.join(CSRC, s)
        if _stale(obj, [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]):
            run([HIPCC] + FLAGS + ['-DFC_TRACE', '-c', src, '-o', obj])
    tlib = os.path.join(tdir, 'libfcaf3d_hip.so')
    run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tlib] + objs)
    run([HIPCC, '-O2', '-std=c++17', '--offload-arch=gfx950', os.path.join(tools, 'nbench.cpp'), '-I' + inc, '-L' + tdir, '-lfcaf3d_hip', '-ldl',
         '-Wl,-rpath,$ORIGIN/trace', '-o', os.path.join(tools, 'nbench_trace')])


if __name__ == '__main__':
    if '--tools' in sys.argv:
        build_tools()
    else:
        build(force='--force' in sys.argv)
    print(LIB)
