"""Samples what a thread of ANOTHER process is doing, from /proc (tools/, not product): state (R running, S sleeping, D disk wait) and,
when it sleeps, the kernel function it sleeps in (wchan) and the system call it is in — ~2 000 samples per second, from a separate
process so that the sampled interpreter is not disturbed.  bench.py --sample-main-thread runs it over the timed region: a step whose
host side is slow shows here as time asleep in a futex (a lock another thread holds), in a KFD event wait (the runtime waiting for the
GPU) or as plain running.
    python tools/wchan.py <pid> <tid> <out.json>      (SIGUSR1 starts the sampling, SIGTERM ends it and writes the histogram)"""
import collections
import json
import signal
import sys
import time


def main():
    pid, tid, out = sys.argv[1], sys.argv[2], sys.argv[3]
    base = f'/proc/{pid}/task/{tid}/'
    hist = collections.Counter()
    stop = []
    go = []
    signal.signal(signal.SIGTERM, lambda *a: stop.append(1))
    signal.signal(signal.SIGUSR1, lambda *a: go.append(1))
    while not go and not stop:               # the sampled process says when its region starts
        time.sleep(0.001)
    t0 = time.time()
    n = 0
    while not stop:
        try:
            st = open(base + 'stat').read()
            state = st[st.rindex(')') + 2]
            w = ''
            if state != 'R':
                try:
                    w = open(base + 'wchan').read().strip()
                except OSError:
                    w = '?'
                try:
                    sc = open(base + 'syscall').read().split()[0]
                except OSError:
                    sc = '?'
                w = f'{w} syscall {sc}'
            hist[f'{state} {w}'.strip()] += 1
            n += 1
        except (OSError, ValueError):
            break
        time.sleep(0.0003)
    dt = time.time() - t0
    json.dump(dict(samples=n, seconds=round(dt, 3), share={k: round(v / max(n, 1), 4) for k, v in hist.most_common(12)}), open(out, 'w'))


if __name__ == '__main__':
    main()
