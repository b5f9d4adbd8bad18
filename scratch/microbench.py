import os, sys, time
import numpy as np, torch
sys.path.insert(0, '/root/repo')
import fcaf3d_amd as fa
import fcaf3d_amd.functional as Fn
from fcaf3d_amd import _lib as L
from fcaf3d_amd.sparse import CoordMap
dev = torch.device('cuda:0')
def tm(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
# col_stats on the stem shape
n, C, B = 580000, 64, 8
x = torch.randn(n, C, device=dev)
seg = torch.zeros((n, 4), dtype=torch.int32, device=dev)
seg[:, 0] = (torch.arange(n, device=dev) * B // n).to(torch.int32)
print('col_stats 8 segments us', tm(lambda: Fn.col_stats(x, seg, B)))
print('col_stats 1 segment  us', tm(lambda: Fn.col_stats(x, None, 1)))
# stem conv through the real maps of the bench batch
import bench
sys.argv = ['x']
args = bench.parse()
batches = bench.make_batches(args, 0, dev)
model, cfg = bench.build_model(args)
model = model.to(dev).train()
coords, feats = model.voxelize(batches[0]['points'])
from fcaf3d_amd.sparse import SparseTensor
xs = SparseTensor(feats, coordinates=coords, batch_size=8)
cm0 = xs.cmap
m1 = cm0.strided(2); km = cm0.kernel_map(m1, 3)
w = torch.randn(27, 3, 64, device=dev, requires_grad=True)
f = xs.F.detach()
go = torch.randn(m1.n, 64, device=dev)
print('rows in/out', f.shape[0], m1.n)
for col in (True, False):
    Fn.STEM_COL = col
    Fn.WGRAD_ASYNC = False
    out = Fn.sparse_conv(f, w, km, m1.n)
    print('STEM_COL', col, 'fwd us', tm(lambda: Fn.sparse_conv(f, w, km, m1.n)))
    def fb():
        o = Fn.sparse_conv(f, w, km, m1.n)
        o.backward(go)
        w.grad = None
    print('STEM_COL', col, 'fwd+wgrad us', tm(fb))
