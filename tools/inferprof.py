"""Diagnostic: where does one simple_test batch spend its time?  Synchronised phase timers, enqueue-only time and the
aten-op count of the decode / NMS part (tools/, not product).   python tools/inferprof.py [bench.py flags]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    # --train-first N: N training steps before the measurement (trained-looking scores: many more NMS candidates than at
    # initialisation); --save-state / --load-state PATH: hand that state to a second process (e.g. one under rocprofv3)
    opts = {}
    for flag in ('--train-first', '--save-state', '--load-state'):
        if flag in sys.argv:
            k = sys.argv.index(flag)
            opts[flag] = sys.argv[k + 1]
            del sys.argv[k:k + 2]
    args = bench.parse()
    dev = torch.device('cuda:0')
    model, cfg = bench.build_model(args)
    model = model.to(dev)
    if '--load-state' in opts:
        model.load_state_dict(torch.load(opts['--load-state'], map_location=dev))
    if '--train-first' in opts:
        from fcaf3d_amd.runner import TrainStep
        model.train()
        model.async_maps = True
        model.inputs_resident = True
        tr = TrainStep.from_config(model, cfg)
        bt = bench.make_batches(args, 0, dev)
        for i in range(int(opts['--train-first'])):
            tr(bt[i % len(bt)])
        torch.cuda.synchronize()
    if '--save-state' in opts:
        torch.save(model.state_dict(), opts['--save-state'])
    model = model.eval()
    model.static_weights = True
    model.async_maps = True
    model.inputs_resident = True
    batches = bench.make_batches(args, 0, dev)
    tb = [dict(points=b['points'], img_metas=b['img_metas']) for b in batches]

    def run(i, t=None):
        b = tb[i % 2]
        marks = [time.perf_counter()]

        def mark():
            if t is not None:
                torch.cuda.synchronize()
            marks.append(time.perf_counter())
        with torch.no_grad():
            x = model.extract_feat(b['points'], b['img_metas'])
            mark()
            res = model.neck_with_head.get_bboxes(*x, b['img_metas'])
            mark()
        if t is not None:
            for j in range(2):
                t[j] += marks[j + 1] - marks[j]
        return res

    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    t = [0.0, 0.0]
    for i in range(4):
        run(i, t)
    print('synced ms/batch: extract_feat %.2f | get_bboxes %.2f' % (1e3 * t[0] / 4, 1e3 * t[1] / 4))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(4):
        run(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('host-return ms/batch %.2f ; drained ms/batch %.2f' % (1e3 * (t1 - t0) / 4, 1e3 * (t2 - t0) / 4))
    # extract_feat only, un-synced inside
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(4):
            model.extract_feat(tb[i % 2]['points'], tb[i % 2]['img_metas'])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('extract_feat only: host %.2f ms/batch, drained %.2f ms/batch' % (1e3 * (t1 - t0) / 4, 1e3 * (t2 - t0) / 4))
    if os.environ.get('FC_CPROFILE'):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for i in range(3):
            run(i)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(70)
    n_det = [len(r[1]) for r in run(0)]
    print('detections per scene', n_det)


if __name__ == '__main__':
    main()
