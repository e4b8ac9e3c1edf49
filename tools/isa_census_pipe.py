#!/usr/bin/env python
"""Static check of the pipelined roll-out kernels (rollout_pipe.inc, compiled as part of rollout_persist.hip): every role of a kernel is a
loop nest of its own.  For every loop (backward branch) that contains MFMAs: instructions, MFMAs, AGPRs the MFMAs read as weights, AGPR
writes into those registers inside the loop (must be 0: a copy into an AGPR in front of an inline-asm MFMA is an unguarded hazard) and
scratch traffic (must be 0).  usage: isa_census_pipe.py [extra hipcc flags]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'humor_amd', 'csrc', 'rollout_persist.hip')


def main():
    out = os.path.join(tempfile.mkdtemp(), 'p.s')
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-S', '--cuda-device-only', SRC, '-o', out] + sys.argv[1:]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    text = open(out).read()
    lines = text.split('\n')
    funcs, cur = {}, None
    for ln in lines:
        m = re.match(r'^(_ZN2ha\w+):', ln)
        if m:
            cur = m.group(1)
            funcs[cur] = []
        elif ln.startswith('.Lfunc_end'):
            cur = None
        elif cur:
            funcs[cur].append(ln)
    for name, body in funcs.items():
        if 'rollout_pipe' not in name:
            continue
        labels, insts = {}, []
        for ln in body:
            t = ln.strip()
            m = re.match(r'^(\.LBB\w+):', t)
            if m:
                labels[m.group(1)] = len(insts)
                continue
            if not t or t.startswith(('.', ';', '//')):
                continue
            insts.append(t.split(';')[0].strip())
        short = ('fwd' if 'fwd' in name else 'bwd') + ('<true>' if 'Lb1' in name else '<false>')
        m = re.search(r'\.name:\s+' + re.escape(name) + r'\n(?:.*\n){0,12}?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', text)
        print(f'pipe {short}: {len(insts)} instructions, vgpr_count {m.group(1) if m else "?"}, spills {m.group(2) if m else "?"}')
        loops = []
        for i, t in enumerate(insts):
            m = re.match(r'^s_c?branch\w*\s+(\.LBB\w+)', t)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        # outermost MFMA loops only (a loop that is not inside another MFMA-containing loop)
        mfma_loops = [(a, b) for a, b in loops if any(t.startswith('v_mfma') for t in insts[a:b + 1])]
        outer = [l for l in mfma_loops if not any(o != l and o[0] <= l[0] and l[1] <= o[1] for o in mfma_loops)]
        for a, b in sorted(outer):
            loop = insts[a:b + 1]
            mix = collections.Counter(t.split()[0] for t in loop)
            wregs = set(m.group(1) for t in loop for m in [re.match(r'^v_mfma\S*\s+\S+,\s*\S+,\s*(a\d+),', t)] if m)
            written = set(m.group(1) for t in loop for m in [re.match(r'^v_accvgpr_write_b32\s+(a\d+),', t)] if m)
            clash = sorted(wregs & written)
            nop_cycles = sum(int(t.split()[1]) + 1 for t in loop if t.startswith('s_nop'))
            print(f'   role loop [{a}, {b}]: {len(loop)} instructions, v_mfma {sum(v for k, v in mix.items() if k.startswith("v_mfma"))}, weight AGPRs read by MFMAs {len(wregs)}, '
                  f'v_accvgpr_write INTO WEIGHT AGPRS IN THE LOOP {len(clash)}, v_accvgpr_write (any) {mix.get("v_accvgpr_write_b32", 0)}, v_accvgpr_read {mix.get("v_accvgpr_read_b32", 0)}, '
                  f'scratch ops {sum(v for k, v in mix.items() if k.startswith("scratch_"))}, s_nop wait states {nop_cycles}, ds ops {sum(v for k, v in mix.items() if k.startswith("ds_"))}, '
                  f'global/buffer {sum(v for k, v in mix.items() if k.startswith(("global_", "buffer_")))}, v_mov {mix.get("v_mov_b32", 0)}')


if __name__ == '__main__':
    main()
