#!/usr/bin/env python
"""Writes profiles/<run>/SUMMARY.txt from a run's bench.json + rocprofv3 bench_kernel_stats.csv.
usage: python tools/profile_summary.py profiles/r02_run19 ["note"]"""
import csv
import json
import os
import sys


def main():
    d = sys.argv[1]
    note = sys.argv[2] if len(sys.argv) > 2 else ''
    b = json.loads(open(os.path.join(d, 'bench.json')).read().strip().splitlines()[-1])
    rows = list(csv.DictReader(open(os.path.join(d, 'bench_kernel_stats.csv'))))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    ours = sum(float(r['TotalDurationNs']) for r in rows if 'ha::' in r['Name'])
    out = [f'{os.path.basename(d)} -- rocprofv3 --kernel-trace --stats of `python bench.py --no-cpu-baseline --no-lbfgs --no-rccl-check` on one MI355X', note,
           f"bench (un-profiled run, same box): {b['ms_per_step']:.3f} ms/closure = {b['value']:.1f} closure-evals/s; {b.get('closure_mode', '')}",
           f'total kernel time {tot / 1e6:.1f} ms over {sum(int(r["Calls"]) for r in rows)} dispatches; humor_amd kernels {100 * ours / tot:.1f} % of it', '',
           f'{"kernel":<80} {"calls":>7} {"total ms":>9} {"avg us":>9} {"min us":>8} {"%":>6}']
    for r in rows[:26]:
        out.append(f'{r["Name"][:80]:<80} {int(r["Calls"]):7d} {float(r["TotalDurationNs"]) / 1e6:9.2f} {float(r["AverageNs"]) / 1e3:9.2f} '
                   f'{float(r["MinNs"]) / 1e3:8.2f} {float(r["Percentage"]):6.2f}')
    rf = b['roofline']
    out += ['', f"roofline kernel {rf['kernel']} at N={rf['frames_per_launch']}: bench.py event-timed {rf['avg_launch_us']} us -> {rf['achieved']} GB/s algorithmic "
                f"= {rf['frac']} of 8 TB/s; PMC traffic {rf['traffic'] / 1e6:.1f} MB vs {rf['bytes_per_launch'] / 1e6:.1f} MB algorithmic"]
    for r in rows:
        if 'lbs_skin_wave_kernel' in r['Name']:
            out.append(f"  rocprofv3: {r['Name'][:60]} calls {r['Calls']} avg {float(r['AverageNs']) / 1e3:.1f} us min {float(r['MinNs']) / 1e3:.1f} max {float(r['MaxNs']) / 1e3:.1f}")
    if 'roofline_c5' in b:
        r5 = b['roofline_c5']
        out.append(f"roofline_c5 (N={r5['frames_per_launch']}, cache-free): {r5['avg_launch_us']} us -> {r5['achieved']} GB/s = {r5['frac']}")
    if 'c5_rooflines' in b:
        out.append('C5 (256x120): ' + json.dumps(b['c5_rooflines']))
    if 'lbfgs' in b:
        lb = b['lbfgs']
        out.append('L-BFGS outer iterations/s: ' + ', '.join(f"{k} {v['outer_iters_per_sec']:.1f}" for k, v in lb['phases'].items()) +
                   f"; whole 30/80/70 fit {lb['whole_fit_seconds_for_30_80_70_schedule']} s")
    if b.get('cpu_baseline'):
        c = b['cpu_baseline']
        out.append(f"cpu_baseline: {c['value']} {c['unit']} on {c['cores']} threads ({c['kind']}) -> GPU/CPU = {b['value'] / c['value']:.0f}x")
    out.append(f"dense SMPL forward (6890 verts, N=1920): {b['smpl_dense_fwd_ms']} ms = {b['smpl_verts_per_sec'] / 1e9:.1f} G verts/s")
    if 'smpl_dense_fwd_bwd_ms' in b:
        out.append(f"dense SMPL forward + backward (gradient on every vertex): {b['smpl_dense_fwd_bwd_ms']} ms = {b['smpl_verts_per_sec_fwd_bwd'] / 1e9:.1f} G verts/s")
    if 'rollout' in b:
        out.append('roll-out 32 x 59 alone: ' + json.dumps(b['rollout']))
    open(os.path.join(d, 'SUMMARY.txt'), 'w').write('\n'.join(out) + '\n')
    print('\n'.join(out))


if __name__ == '__main__':
    main()
