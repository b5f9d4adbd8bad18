"""Busy / idle timeline of a training step from a rocprofv3 kernel trace (tools/, not product).

  rocprofv3 --kernel-trace --output-format csv -d <dir> -o t -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline \
      --infer-steps 0 --no-instrument
  python tools/timeline.py <dir>/<host>/t_kernel_trace.csv [--steps 6]

Splits the trace into steps at the optimizer's AdamW kernel (k_adamw), each step into forward (up to the loss kernel),
backward (up to the first optimizer kernel) and optimizer, and prints per phase: wall time, time with >= 1 kernel
running (union of intervals), idle time, the sum of kernel durations per stream/queue, and the kernels that follow the
largest gaps — i.e. where the dependent chain of small launches leaves the chip empty."""
import collections
import csv
import sys


def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0'),
                         r.get('Stream_Id', '0')))
    rows.sort()
    return rows


def union(iv):
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy


def short(name):
    name = name.replace('void ', '')
    return name[:70]


def main():
    path = sys.argv[1]
    rows = load(path)
    # step boundaries: the LAST optimizer kernel of a step (fused AdamW = multi_tensor_apply with FusedAdam functor)
    opt = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw') or 'FusedAdam' in r[2] or 'fused_adam' in r[2].lower()]
    if not opt:
        opt = [i for i, r in enumerate(rows) if 'multi_tensor_apply' in r[2]]
    bounds, prev = [], None
    for i in opt:
        if prev is not None and rows[i][0] - rows[prev][1] > 3_000_000:      # > 3 ms since the last optimizer kernel: a new step
            bounds.append(prev)
        prev = i
    bounds.append(prev)
    print(f'{len(rows)} kernels, {len(bounds)} optimizer phases')
    steps = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        steps.append(rows[a + 1:b + 1])
    # keep the steady-state steps of the timed region (similar kernel counts)
    if not steps:
        return
    med = sorted(len(s) for s in steps)[len(steps) // 2]
    steps = [s for s in steps if abs(len(s) - med) < 0.05 * med]
    print(f'{len(steps)} steady steps of ~{med} kernels')
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0])     # phase -> wall, busy, kernel-sum
    gaps = collections.Counter()
    gapn = collections.Counter()
    per_stream = collections.defaultdict(float)
    for st in steps:
        t_f = next((r[0] for r in st if 'k_focal_fwd' in r[2] or 'k_fcaf3d_loss_fwd' in r[2]), None)
        t_o = next((r[0] for r in st if r[2].startswith('k_sqsum_partial') or 'multi_tensor_apply' in r[2] or 'FusedAdam' in r[2]), None)
        t0, t1 = st[0][0], max(r[1] for r in st)
        phases = {'forward': (t0, t_f), 'backward': (t_f, t_o), 'optimizer': (t_o, t1)}
        for ph, (a, b) in phases.items():
            iv = [(max(s, a), min(e, b)) for s, e, *_ in st if e > a and s < b]
            agg[ph][0] += b - a
            agg[ph][1] += union(sorted(iv))
            agg[ph][2] += sum(e - s for s, e in iv)
        # gaps: time between the end of everything so far and the next kernel's start
        end = st[0][1]
        for s, e, name, q, sid in st[1:]:
            if s > end:
                ph = 'F' if s < t_f else ('B' if s < t_o else 'O')
                gaps[(ph, short(name))] += s - end
                gapn[(ph, short(name))] += 1
            end = max(end, e)
        for s, e, name, q, sid in st:
            per_stream[(q, sid)] += e - s
    n = len(steps)
    print(f'{"phase":10s} {"wall ms":>9s} {"busy ms":>9s} {"idle ms":>9s} {"kernel-sum ms":>14s}')
    tw = tb = tk = 0
    for ph in ('forward', 'backward', 'optimizer'):
        w, b, k = (x / n / 1e6 for x in agg[ph])
        tw, tb, tk = tw + w, tb + b, tk + k
        print(f'{ph:10s} {w:9.3f} {b:9.3f} {w - b:9.3f} {k:14.3f}')
    print(f'{"step":10s} {tw:9.3f} {tb:9.3f} {tw - tb:9.3f} {tk:14.3f}')
    print('kernel-sum per (queue, stream) ms/step:', {k: round(v / n / 1e6, 3) for k, v in sorted(per_stream.items())})
    print('--- idle time in front of a kernel (chip empty), ms/step, top 30 ---')
    for (ph, name), t in gaps.most_common(30):
        print(f'{ph} {name:70s} {t / n / 1e6:8.3f} ms  {gapn[(ph, name)] / n:6.1f} gaps  {t / gapn[(ph, name)] / 1e3:7.1f} us each')
    tot = collections.Counter()
    for (ph, name), t in gaps.items():
        tot[ph] += t
    print('idle by phase ms/step:', {k: round(v / n / 1e6, 3) for k, v in tot.items()})


if __name__ == '__main__':
    main()
