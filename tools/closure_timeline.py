#!/usr/bin/env python
"""Timeline of ONE steady-state stage-3 closure evaluation from a rocprofv3 kernel trace of `closure_n.py N`: every dispatch of the
second-to-last evaluation with its start offset, duration, HW queue and the gap to the previous end on the same queue, and the time
during which two queues are busy at once (what the side-stream prior buys).
usage: rocprofv3 --kernel-trace --output-format csv -d OUT -- python tools/closure_n.py 10 ;  closure_timeline.py OUT"""
import csv
import glob
import os
import sys


def main():
    rows = []
    for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:48], r['Queue_Id']))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if 'rollout_persist_fwd' in r[2]]
    if len(starts) < 3:
        print('need at least three evaluations in the trace')
        return
    # one evaluation = from the end of a persistent adjoint to the end of the next one
    ends = [i for i, r in enumerate(rows) if 'rollout_persist_bwd' in r[2]]
    a, b = ends[-3], ends[-2]
    ev = rows[a + 1:b + 1]
    t0 = rows[a][1]
    print(f'evaluation: {len(ev)} dispatches, {(ev[-1][1] - t0) / 1e3:.1f} us from the end of the previous persistent adjoint to the end of this one')
    last_end = {}
    for s, e, n, q in ev:
        gap = (s - last_end[q]) / 1e3 if q in last_end else (s - t0) / 1e3
        last_end[q] = e
        print(f'{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  q{q:>3s}  gap {gap:7.1f}  {n}')
    # overlap: time covered by >= 2 kernels
    evs = sorted([(s, 1) for s, e, n, q in ev] + [(e, -1) for s, e, n, q in ev])
    depth, prev, busy1, busy2 = 0, evs[0][0], 0, 0
    for t, d in evs:
        if depth >= 1:
            busy1 += t - prev
        if depth >= 2:
            busy2 += t - prev
        depth += d
        prev = t
    print(f'GPU busy {busy1 / 1e3:.1f} us, of which >= 2 kernels in flight {busy2 / 1e3:.1f} us; idle {(ev[-1][1] - t0 - busy1) / 1e3:.1f} us')


if __name__ == '__main__':
    main()
