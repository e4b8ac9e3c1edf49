#!/usr/bin/env python
"""Summary of the MFMA PMC passes of tools/pmc_mfma.sh: per kernel, dispatch-mean MFMA instructions, MFMA-busy cycles, the
utilisation they imply and the FLOP rate from SQ_INSTS_VALU_MFMA_MOPS_F32.   usage: pmc_mfma_summary.py <dir> [<dir> ...]

MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x shader clock).  Counter values are summed over the
chip (all XCDs / SEs).  SQ_INSTS_VALU_MFMA_MOPS_F32 counts 512-FLOP units (MI355X_MICROARCH.md, "MFMA counters").

Effective clock (DVFS): SQ_BUSY_CYCLES is summed over the 32 shader engines; for a launch that keeps every SE busy from start to end
(the big GEMMs, the persistent kernels) SQ_BUSY_CYCLES / 32 / duration is the shader clock the launch actually ran at, and
MFMA-busy / SIMD / (SQ-busy / SE) is the share of ISSUED cycles the MFMA pipe was busy -- the utilisation a kernel can influence;
`util` (against 2.4 GHz) additionally carries the clock the power management granted (MI355X_MICROARCH.md, "DVFS give-back")."""
import collections
import csv
import glob
import os
import sys

CLK = 2.4e9
SIMDS = 1024
PEAK_TF = 157.3


def summarise(out):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    meta = {}
    for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r['Kernel_Name'].split('(')[0]
                if k.startswith('void '):
                    k = k[5:]
                rows[k][r['Counter_Name']].append(float(r['Counter_Value']))
                dur[k][r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
                meta[k] = (r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'])
    print(f'== {out}')
    for k in sorted(rows):
        c = rows[k]
        mean = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else 0.0
        insts, busy, mops = mean('SQ_INSTS_MFMA'), mean('SQ_VALU_MFMA_BUSY_CYCLES'), mean('SQ_INSTS_VALU_MFMA_MOPS_F32')
        if insts == 0 and '--all' not in sys.argv:
            continue
        d = sorted(dur[k].values())
        avg = sum(d) / len(d)
        util = busy / (SIMDS * avg * 1e-6 * CLK) if avg > 0 else 0.0
        tf = mops * 512 / (avg * 1e-6) / 1e12 if avg > 0 else 0.0
        g, w, v, a = meta[k]
        sqb = mean('SQ_BUSY_CYCLES') / 32
        clk = f'  eff clock {sqb / avg / 1e3:4.2f} GHz  MFMA pipe {100 * busy / SIMDS / sqb:5.1f} % of issued cycles' if sqb > 0 and avg > 100 else ''
        print(f'{k[:64]:64s} n={len(d):5d} avg {avg:9.1f} us  grid {g:>8s}/{w:>4s} vgpr {v}+{a}  MFMA insts {insts:12.0f}  busy cyc {busy:14.0f}'
              f'  busy/inst {busy / max(insts, 1):5.1f}  util {100 * util:5.1f} %  MOPS-rate {tf:6.1f} TF = {100 * tf / PEAK_TF:5.1f} % of {PEAK_TF}{clk}')


if __name__ == '__main__':
    for d in sys.argv[1:]:
        if not d.startswith('--'):
            summarise(d)
