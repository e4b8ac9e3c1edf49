#!/usr/bin/env python
"""Turns a rocprofv3 rocpd database (…_results.db) into the per-kernel summary committed under profiles/.
usage: python tools/rocprof_summary.py gpurun_out/prof1/r1_results.db profiles/r01_xxx.txt [note]"""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ''
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    with open(out, 'w') as f:
        f.write(f'# rocprofv3 --kernel-trace --stats summary (durations in microseconds)\n# source db: {db}\n# {note}\n')
        f.write(f'{"calls":>7} {"total_us":>12} {"avg_us":>10} {"pct":>6}  kernel\n')
        for name, calls, tot, avg, pct in rows:
            short = name if len(name) < 110 else name[:107] + '...'
            f.write(f'{calls:7d} {tot:12.2f} {avg:10.3f} {pct:6.2f}  {short}\n')
    print(open(out).read()[:1500])


if __name__ == '__main__':
    main()
