#!/usr/bin/env python
"""Groups a rocprofv3 kernel trace (…_kernel_trace.csv) by (kernel, grid size): count, mean / min / max duration in us."""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    acc = collections.OrderedDict()
    for r in rows:
        name = r['Kernel_Name']
        if pat and pat not in name:
            continue
        key = (name[:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''), r.get('Grid_Size_Z', ''))
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        acc.setdefault(key, []).append(d)
    for k, v in acc.items():
        print(f'{len(v):6d} mean {sum(v) / len(v):8.2f} min {min(v):8.2f} max {max(v):8.2f} us  grid {k[1]}x{k[2]}x{k[3]}  {k[0]}')


if __name__ == '__main__':
    main()
