"""Diagnostic (not a test): is a GPU-vs-oracle gradient gap real or fp32 noise?  Compares the HIP path and the
fp32 oracle against the SAME oracle run in fp64.   python tests/diag_grad.py [levels] [B] [n_points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_model import _build, _oracle_params, _scenes, _to_gpu_batch  # noqa: E402
from oracle import model_oracle as MO  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(1e-30, float(b.double().abs().max()))


def main():
    levels = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    npts = int(sys.argv[3]) if len(sys.argv) > 3 else 30000
    dev = torch.device('cuda:0')
    model, m = _build('fcaf3d_scannet-3d-18class', 0.02, levels)
    P32 = _oracle_params(model)
    P64 = {k: (v.detach().double().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in P32.items()}
    model = model.to(dev).train()
    pts, gts, labs = _scenes(range(10, 10 + B), n_points=npts)
    lg = model(return_loss=True, **_to_gpu_batch(pts, gts, labs, dev))
    sum(lg.values()).backward()
    l32 = MO.forward_train(P32, m, pts, gts, labs)
    sum(l32.values()).backward()
    l64 = MO.forward_train(P64, m, pts, gts, labs)
    sum(l64.values()).backward()
    print('losses gpu/o32/o64', {k: (float(lg[k]), float(l32[k]), float(l64[k])) for k in lg})
    rows = []
    for k, p in model.named_parameters():
        g64 = P64[k].grad
        rows.append((rel(p.grad.cpu(), g64), rel(P32[k].grad, g64), rel(p.grad.cpu(), P32[k].grad), k))
    rows.sort(reverse=True)
    print('worst 12 params:  gpu-vs-fp64   oracle32-vs-fp64   gpu-vs-oracle32')
    for r in rows[:12]:
        print(f'  {r[0]:.2e}  {r[1]:.2e}  {r[2]:.2e}  {r[3]}')


if __name__ == '__main__':
    main()
