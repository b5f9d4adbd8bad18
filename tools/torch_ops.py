"""Which torch (aten) device ops does one training step issue, and from which source line?  (tools/, not product.)
Runs bench.py's model / batches / TrainStep under a TorchDispatchMode and prints calls per step per (aten op, innermost
fcaf3d_amd / bench frame) — the 'torch glue' share of profiles/r2_kernel_stats.csv (fills, copies, adds, cats)
attributed to the lines that cause it.  Backward-pass ops are attributed to the autograd node that issued them.

  python tools/torch_ops.py [--steps 2] [bench.py flags...]
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                    # noqa: E402
from fcaf3d_amd.runner import TrainStep                         # noqa: E402

SKIP = ('aten::view', 'aten::_unsafe_view', 'aten::reshape', 'aten::t', 'aten::transpose', 'aten::slice', 'aten::select',
        'aten::unsqueeze', 'aten::squeeze', 'aten::expand', 'aten::as_strided', 'aten::detach', 'aten::alias', 'aten::empty',
        'aten::empty_like', 'aten::empty_strided', 'aten::permute', 'aten::unbind', 'aten::split', 'aten::narrow',
        'aten::lift_fresh', 'aten::_reshape_alias', 'aten::new_empty', 'aten::record_stream', 'aten::is_pinned')


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func._schema.name
        if name.startswith(SKIP):
            return out
        flat = [a for a in (list(args) + [out]) if torch.is_tensor(a)]
        if not any(t.is_cuda for t in flat):
            return out
        where = 'autograd engine / torch internals'
        for fr in reversed(traceback.extract_stack(limit=40)):
            fn = fr.filename
            if fn.startswith(ROOT) and 'tools/torch_ops.py' not in fn:
                where = f'{os.path.relpath(fn, ROOT)}:{fr.lineno} {fr.name}'
                break
        self.n[(name, where)] += 1
        return out


def main():
    steps = 2
    if '--steps' in sys.argv:
        i = sys.argv.index('--steps')
        steps = int(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    trainer = TrainStep.from_config(model, cfg)
    batches = bench.make_batches(args, 0, dev)
    for i in range(3):
        trainer(batches[i % 2])
    torch.cuda.synchronize()
    with Count() as c:
        for i in range(steps):
            trainer(batches[i % 2])
        torch.cuda.synchronize()
    tot = sum(c.n.values()) / steps
    print(f'aten ops touching device tensors per step: {tot:.0f}')
    byop = collections.Counter()
    for (name, _), v in c.n.items():
        byop[name] += v
    print('by op:', ', '.join(f'{k.replace("aten::", "")} {v / steps:.0f}' for k, v in byop.most_common(25)))
    for (name, where), v in c.n.most_common(90):
        print(f'{v / steps:7.1f}/step  {name:26s} {where}')


if __name__ == '__main__':
    main()
