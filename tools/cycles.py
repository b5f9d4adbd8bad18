"""Which objects of a training step die only in the cyclic garbage collector, and how much device memory they hold (tools/, not product).
Reference cycles around device tensors are freed by a FULL collection only — which CPython runs rarely in a process with a million
long-lived objects — so their memory piles up in the caching allocator (bench.py --steps 1000: +37 MB per step).
    python tools/cycles.py [--steps 12] [bench.py flags]"""
import collections
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402
import fcaf3d_amd.functional as Fn                              # noqa: E402
from fcaf3d_amd.runner import TrainStep                         # noqa: E402


def main():
    steps = 12
    if '--steps' in sys.argv:
        i = sys.argv.index('--steps')
        steps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    Fn.WGRAD_ASYNC = True
    tr = TrainStep.from_config(model, cfg)
    batches = bench.make_batches(args, 0, dev)
    for i in range(4):
        tr(batches[i % 2], next_batch=batches[(i + 1) % 2])
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()
    m0 = torch.cuda.memory_allocated()
    for i in range(steps):
        tr(batches[i % 2], next_batch=batches[(i + 1) % 2])
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    gc.set_debug(gc.DEBUG_SAVEALL)
    n = gc.collect()
    m2 = torch.cuda.memory_allocated()
    print(f'{steps} steps without the collector: allocated {m0 / 2**20:.0f} -> {m1 / 2**20:.0f} MB; a full collection finds {n} objects and frees '
          f'{(m1 - m2) / 2**20:.0f} MB = {(m1 - m2) / 2**20 / steps:.1f} MB per step')
    by = collections.Counter(type(o).__module__ + '.' + type(o).__qualname__ for o in gc.garbage)
    for k, v in by.most_common(25):
        print(f'  {v:7d}  {k}')
    tens = [o for o in gc.garbage if torch.is_tensor(o) and o.is_cuda]
    tens.sort(key=lambda t: -t.numel() * t.element_size())
    print('largest device tensors in cycles:', [(tuple(t.shape), str(t.dtype).split('.')[-1]) for t in tens[:12]])
    # the cycles themselves: strongly connected components of the referent graph among the collected objects
    ids = {id(o): o for o in gc.garbage}
    edges = {i: [id(r) for r in gc.get_referents(o) if id(r) in ids] for i, o in ids.items()}
    index, low, on, stack, comps, counter = {}, {}, set(), [], [], [0]
    sys.setrecursionlimit(100000)

    def strong(v):
        index[v] = low[v] = counter[0]; counter[0] += 1
        stack.append(v); on.add(v)
        for w in edges[v]:
            if w not in index:
                strong(w); low[v] = min(low[v], low[w])
            elif w in on:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = stack.pop(); on.discard(w); comp.append(w)
                if w == v:
                    break
            if len(comp) > 1 or v in edges[v]:
                comps.append(comp)
    for v in list(ids):
        if v not in index:
            strong(v)
    print(len(comps), 'cycles (strongly connected components); the first three:')

    def name(o):
        if isinstance(o, dict):
            return 'dict{' + ','.join(str(k)[:24] for k in list(o)[:8]) + '}'
        if isinstance(o, (tuple, list)):
            return type(o).__name__ + '[' + ','.join(type(x).__name__ for x in list(o)[:6]) + ']'
        return type(o).__qualname__
    for comp in comps[:3]:
        print('  ', ' | '.join(name(ids[i]) for i in comp[:14]))
    # who refers to the largest one?
    if tens:
        t = tens[0]
        refs = [r for r in gc.get_referrers(t) if r is not tens and r is not gc.garbage]
        print('referrers of the largest:', [type(r).__qualname__ + (':' + ','.join(list(r)[:6]) if isinstance(r, dict) and all(isinstance(k, str) for k in r) else '') for r in refs][:8])


if __name__ == '__main__':
    main()
