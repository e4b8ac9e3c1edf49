#!/usr/bin/env python
"""Summary of the PMC passes of tools/pmc_lbs.sh: per-dispatch means of every counter for the LBS skinning kernel (and the device copy
launched next to it as the byte-count calibration).  usage: pmc_lbs_summary.py <outdir>"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
V, N = 6890, int(sys.argv[2]) if len(sys.argv) > 2 else 1920
ALG = N * (V * 12 * 2 + 52 * 48 + 12)        # v_posed in + verts out + A + transl, bytes per launch
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = r['Kernel_Name']
            name = 'lbs_skin' if 'lbs_skin' in k else ('copy' if 'copyBuffer' in k else None)
            if name:
                rows[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name in ('lbs_skin', 'copy'):
    if name not in rows:
        continue
    print(f'{name}: per-dispatch means')
    # the first launches of the rotating run touch cold pages; every launch is listed, the mean is over all of them
    for c, v in sorted(rows[name].items()):
        print(f'  {c:24s} n={len(v):3d}  mean {sum(v) / len(v):16.1f}   min {min(v):16.1f}   max {max(v):16.1f}')
r = rows.get('lbs_skin', {})
if 'FETCH_SIZE' in r and 'WRITE_SIZE' in r:
    fe, wr = sum(r['FETCH_SIZE']) / len(r['FETCH_SIZE']), sum(r['WRITE_SIZE']) / len(r['WRITE_SIZE'])
    cal = rows.get('copy', {})
    # machine-readable record for bench.py (roofline.traffic): bytes per launch + the fingerprint of the kernel source it was measured on
    try:
        import json
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        recf = os.path.join(out, 'traffic.json')
        rec = json.load(open(recf)) if os.path.exists(recf) else {'launches': []}
        rec['kernel_fingerprint'] = bench.lbs_kernel_fingerprint()
        rec['how'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only) over rotating operand sets; FETCH x 2 (gfx950 wide-read correction) + WRITE, KiB'
        rec['launches'] = [e for e in rec['launches'] if e.get('frames') != N] + [
            {'frames': N, 'fetch_kib': round(fe, 1), 'write_kib': round(wr, 1), 'hbm_bytes': int((2 * fe + wr) * 1024), 'algorithmic_bytes': ALG}]
        json.dump(rec, open(recf, 'w'), indent=1)
    except Exception as e:          # the summary itself must not fail on this
        print('traffic.json not written:', e)
    print(f'lbs_skin HBM traffic per launch: FETCH_SIZE {fe:.1f} KiB x 2 (gfx950 wide-read correction, MI355X_MICROARCH.md "HBM") + WRITE_SIZE {wr:.1f} KiB'
          f' = {(2 * fe + wr) * 1024 / 1e6:.1f} MB; algorithmic {ALG / 1e6:.1f} MB -> ratio {(2 * fe + wr) * 1024 / ALG:.3f}')
    if 'FETCH_SIZE' in cal and 'WRITE_SIZE' in cal:
        big = lambda v: [x for x in v if x > 0.5 * max(v)]      # the three v_posed-sized copies (the others are the small set-up copies)
        cf, cw = sum(big(cal['FETCH_SIZE'])) / len(big(cal['FETCH_SIZE'])), sum(big(cal['WRITE_SIZE'])) / len(big(cal['WRITE_SIZE']))
        b = N * V * 12
        print(f'calibration (device copy of {b / 1e6:.1f} MB): FETCH_SIZE x 2 = {2 * cf * 1024 / 1e6:.1f} MB, WRITE_SIZE = {cw * 1024 / 1e6:.1f} MB')
if 'SQ_LDS_BANK_CONFLICT' in r and 'SQ_LDS_IDX_ACTIVE' in r:
    bc, ia = sum(r['SQ_LDS_BANK_CONFLICT']) / len(r['SQ_LDS_BANK_CONFLICT']), sum(r['SQ_LDS_IDX_ACTIVE']) / len(r['SQ_LDS_IDX_ACTIVE'])
    print(f'LDS: bank-conflict cycles / LDS-active cycles = {bc:.0f} / {ia:.0f} = {bc / max(ia, 1):.3f}')
