"""Diagnostic (tools/): pipelined inference (simple_test_async, two batches in flight) against the synchronous loop, on the
default stream / a side stream / a high-priority side stream, optionally after a few training steps in the same process.

    python tools/pipeprof.py [--train-first N] [bench.py flags]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def loop(model, tb, n, pipelined):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    te = tf = 0.0
    pending = None
    with torch.no_grad():
        for i in range(n):
            a = time.perf_counter()
            if pipelined:
                h = model.simple_test_async(**tb[i % len(tb)])
                b = time.perf_counter()
                if pending is not None:
                    pending()
                pending = h
            else:
                h = model.simple_test_async(**tb[i % len(tb)])
                b = time.perf_counter()
                h()
            c = time.perf_counter()
            te += b - a
            tf += c - b
        if pending is not None:
            pending()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return 1e3 * dt / n, 1e3 * te / n, 1e3 * tf / n


def main():
    n_train = 0
    if '--train-first' in sys.argv:
        k = sys.argv.index('--train-first')
        n_train = int(sys.argv[k + 1])
        del sys.argv[k:k + 2]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev)
    model.async_maps = True
    model.inputs_resident = True
    batches = bench.make_batches(args, 0, dev)
    tb = [dict(points=b['points'], img_metas=b['img_metas']) for b in batches]
    for kind in ('default', 'side', 'priority'):
        s = None if kind == 'default' else torch.cuda.Stream(priority=-1 if kind == 'priority' else 0)
        ctx = torch.cuda.stream(s) if s is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            if n_train:
                from fcaf3d_amd.runner import TrainStep
                model.train()
                tr = TrainStep.from_config(model, cfg)
                for i in range(n_train):
                    tr(batches[i % len(batches)])
                torch.cuda.synchronize()
            model.eval()
            model.static_weights = True
            loop(model, tb, 4, True)
            for pipelined in (False, True, False, True):
                t, te, tf = loop(model, tb, 16, pipelined)
                print(f'{kind:9s} train_first={n_train} {"pipelined" if pipelined else "sync     "}: {t:6.2f} ms/batch (enqueue half {te:5.2f}, finish half {tf:5.2f})')


if __name__ == '__main__':
    main()
