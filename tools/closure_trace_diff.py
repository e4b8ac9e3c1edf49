#!/usr/bin/env python
"""Per-closure kernel census from two rocprofv3 kernel traces of tools/closure_n.py (N1 < N2 evaluations): (count(N2) - count(N1)) /
(N2 - N1) per kernel, mean duration, and the busy / idle split of the last evaluations of the longer trace.
usage: closure_trace_diff.py trace_N1.csv N1 trace_N2.csv N2"""
import collections
import csv
import sys


def load(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    return rows


def main():
    a, n1, b, n2 = load(sys.argv[1]), int(sys.argv[2]), load(sys.argv[3]), int(sys.argv[4])
    ca, cb = collections.Counter(r['Kernel_Name'] for r in a), collections.Counter(r['Kernel_Name'] for r in b)
    dur = collections.defaultdict(list)
    for r in b:
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    per = {k: (cb[k] - ca.get(k, 0)) / (n2 - n1) for k in cb}
    tot_n = sum(v for v in per.values() if v > 0)
    tot_us = sum(per[k] * sum(dur[k]) / len(dur[k]) for k in per if per[k] > 0)
    print(f'launches per closure evaluation: {tot_n:.1f}; kernel time per evaluation {tot_us / 1e3:.3f} ms')
    print(f'{"per eval":>9} {"avg us":>8} {"ms/eval":>8}  kernel')
    for k, v in sorted(per.items(), key=lambda kv: -kv[1] * sum(dur[kv[0]]) / len(dur[kv[0]])):
        if v <= 0:
            continue
        m = sum(dur[k]) / len(dur[k])
        print(f'{v:9.1f} {m:8.2f} {v * m / 1e3:8.3f}  {k[:110]}')
    # busy / idle of the tail of the longer trace (the last (n2 - n1) evaluations ~ the last tot_n * (n2 - n1) dispatches)
    tail = b[-int(tot_n * (n2 - n1)):]
    span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / 1e3
    busy = sum((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tail)
    gaps = [(int(tail[i + 1]['Start_Timestamp']) - int(tail[i]['End_Timestamp'])) / 1e3 for i in range(len(tail) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f'tail of the trace: {len(tail)} dispatches over {span / 1e3:.3f} ms = {span / (n2 - n1) / 1e3:.3f} ms per evaluation; '
          f'kernel time {busy / 1e3:.3f} ms; positive gaps {sum(pos) / 1e3:.3f} ms (mean {sum(pos) / max(1, len(pos)):.2f} us, '
          f'{sum(1 for g in pos if g > 5)} gaps > 5 us totalling {sum(g for g in pos if g > 5) / 1e3:.3f} ms)')
    big = collections.Counter()
    for i, g in enumerate(gaps):
        if g > 5:
            big[(tail[i]['Kernel_Name'][:50], tail[i + 1]['Kernel_Name'][:50])] += g
    for (k0, k1), g in big.most_common(12):
        print(f'   {g / (n2 - n1):8.1f} us/eval of gaps > 5 us between  {k0}  ->  {k1}')


if __name__ == '__main__':
    main()
