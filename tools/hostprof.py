"""Diagnostic (tools/, not product): where does the HOST spend its time in one training step of bench.py?

For every --batches entry B: the step exactly as bench.py runs it (runner.TrainStep, coordinate work on its side stream,
weight gradients on theirs, priority main stream), measured three ways over the same steps:
  * enqueue-only  — host wall time until all work of a step is ENQUEUED (no synchronisation inside), per phase;
  * drained       — wall time per step including the final synchronise (= what bench.py reports);
  * serial phases — each phase followed by a synchronise (GPU time of a phase with nothing overlapping it);
plus the number of C-ABI calls per step and, with --cprofile B, the cProfile listing for that batch size.

    python tools/hostprof.py --batches 2,4,8 [--cprofile 2] [bench.py flags ...]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


LOOKAHEAD = False


def pop_flag(name, default):
    if name in sys.argv:
        i = sys.argv.index(name)
        v = sys.argv[i + 1]
        del sys.argv[i:i + 2]
        return v
    return default


def run_one(args, B, dev, do_cprofile, steps=8):
    import fcaf3d_amd._lib as L
    import fcaf3d_amd.executor as EX
    import fcaf3d_amd.functional as Fn
    from fcaf3d_amd.runner import TrainStep, parse_losses
    args.batch = B
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    Fn.WGRAD_ASYNC = not args.no_wgrad_overlap
    tr = TrainStep.from_config(model, cfg)
    batches = bench.make_batches(args, 0, dev)
    ncalls = [0]
    orig = L.call

    def counting(name, *a):
        ncalls[0] += 1
        return orig(name, *a)

    def step(i, marks=None, sync=False):
        b = batches[i % 2]

        def mark():
            if marks is not None:
                if sync:
                    torch.cuda.synchronize()
                marks.append(time.perf_counter())
        mark()
        tr._bound_run_ahead()
        if LOOKAHEAD:
            model.prefetch(batches[(i + 1) % 2]['points'], gt=True)
        tr.optimizer.zero_grad(set_to_none=True)
        prog = tr._program()
        if prog is not None:
            if not prog.weights_fresh:
                prog.refresh_weights()
        elif tr.images is not None and tr.images.n:
            if tr.images_version != sum(w._version for w in tr._image_ws):
                tr._build_images(side_stream=False)
            Fn.PREBUILT, Fn.PREBUILT_EVENT = tr.images.table, tr.images.event
        EX.TRUSTED = prog
        x = model.extract_feat(b['points'], b['img_metas'], (b['gt_bboxes_3d'], b['gt_labels_3d']))
        x = [list(v) for v in x]
        mark()
        losses = model.neck_with_head.loss(*x, b['gt_bboxes_3d'], b['gt_labels_3d'], b['img_metas'])
        loss = parse_losses(losses)
        mark()
        loss.backward()
        Fn.PREBUILT, Fn.PREBUILT_EVENT = {}, None
        EX.TRUSTED = None
        tr.averager.finish()
        mark()
        tr.optimizer.step(tr.max_norm)
        for pr in getattr(model, '_programs', {}).values():
            pr.weights_fresh = False
        if prog is not None:
            side, main = Fn.wgrad_stream(dev), torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                prog.refresh_weights()
        elif tr.images is not None and tr.images.n:
            tr._build_images(side_stream=True)
        mark()

    names = ['forward (extract_feat)', 'loss', 'backward', 'clip + AdamW + images']
    for i in range(4):
        step(i)
    torch.cuda.synchronize()
    out = {}
    for label, sync in (('enqueue-only', False), ('serial phases', True)):
        tot = [0.0] * 4
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            m = []
            step(i, m, sync)
            for k in range(4):
                tot[k] += m[k + 1] - m[k]
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out[label] = ([1e3 * v / steps for v in tot], 1e3 * (t1 - t0) / steps, 1e3 * (t2 - t0) / steps)
    if LOOKAHEAD:
        import fcaf3d_amd.plan as PL
        PL.TRACE = []
        t00 = time.perf_counter()
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        print('plan trace (ms since start): ' + ' '.join(f'{tag}@{(tt - t00) * 1e3:.2f}' for tag, tt in PL.TRACE))
        PL.TRACE = None
    L.call = counting
    step(0)
    L.call = orig
    torch.cuda.synchronize()
    print(f'=== B = {B} scenes per step ({args.workload}) ===')
    for label in ('enqueue-only', 'serial phases'):
        ph, enq, drained = out[label]
        print(f'{label:14s}: ' + ' | '.join(f'{n} {v:.2f}' for n, v in zip(names, ph)) + f'  || sum {sum(ph):.2f} ms/step')
        if label == 'enqueue-only':
            print(f'{"":14s}  host enqueue {enq:.2f} ms/step ; drained {drained:.2f} ms/step ({B / drained * 1e3:.1f} scenes/s)')
    print(f'C-ABI calls per step: {ncalls[0]}')
    if do_cprofile:
        pr = cProfile.Profile()
        pr.enable()
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats('cumulative').print_stats(60)
        st.sort_stats('tottime').print_stats(40)
    sys.stdout.flush()


def main():
    bs = [int(v) for v in pop_flag('--batches', '2,4,8').split(',')]
    cp = int(pop_flag('--cprofile', '0'))
    global LOOKAHEAD
    if '--lookahead' in sys.argv:
        sys.argv.remove('--lookahead')
        LOOKAHEAD = True
    args = bench.parse()
    dev = torch.device('cuda:0')
    if args.priority_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    for B in bs:
        run_one(args, B, dev, cp == B)


if __name__ == '__main__':
    main()
