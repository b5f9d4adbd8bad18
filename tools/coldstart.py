"""Diagnostic (tools/, not product): the step time of the benchmark loop over the first seconds of a process on a fresh box —
rocm-smi clocks / power before and after, average ms per 5 steps with the wall clock.  python tools/coldstart.py [--steps 120]"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def smi(tag):
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showtemp', '--showperflevel'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'mclk', 'fclk', 'Power', 'Temperature (Sensor junction', 'Performance Level'))]
        print(f'--- rocm-smi {tag} ---')
        print('\n'.join(keep[:12]), flush=True)
    except Exception as e:
        print('rocm-smi failed', e)


def main():
    n = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 120
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    dev = torch.device('cuda:0')
    smi('before')
    import fcaf3d_amd.functional as Fn
    from fcaf3d_amd.runner import TrainStep
    torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))
    model, cfg = bench.build_model(args)
    model = model.to(dev).train()
    model.async_maps = True
    model.inputs_resident = True
    Fn.WGRAD_ASYNC = True
    tr = TrainStep.from_config(model, cfg)
    batches = bench.make_batches(args, 0, dev)
    t_start = time.perf_counter()
    torch.cuda.synchronize()
    for g in range(n // 5):
        t0 = time.perf_counter()
        for i in range(5):
            tr(batches[i % 2])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        print(f't = {t1 - t_start:6.2f} s   steps {5 * g:3d}-{5 * g + 4:3d}: {(t1 - t0) / 5 * 1e3:7.2f} ms/step', flush=True)
        if g == 1:
            smi('after 10 steps')
    smi('after')


if __name__ == '__main__':
    main()
