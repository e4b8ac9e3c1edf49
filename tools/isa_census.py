#!/usr/bin/env python
"""Static instruction census of the per-step loops of the persistent roll-out kernels: compiles rollout_persist.hip to assembly,
finds in each kernel the outermost backward branch (the step loop) and counts the instructions between its target and the branch.
usage: isa_census.py [extra hipcc flags]      (prints per kernel: instructions per step and wave, opcode mix, counts of interest)"""
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
    lines = open(out).read().split('\n')
    # split into functions
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
        if 'rollout_persist' not in name:
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
        # backward branches: (target index, branch index); the step loop = the one spanning the most instructions
        best = (0, 0)
        for i, t in enumerate(insts):
            m = re.match(r'^s_c?branch\w*\s+(\.LBB\w+)', t)
            if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > best[1] - best[0]:
                best = (labels[m.group(1)], i)
        loop = insts[best[0]:best[1] + 1]
        mix = collections.Counter(t.split()[0] for t in loop)
        # AGPRs the inline-asm MFMAs read as their B operand vs AGPRs the loop writes (a spill INTO a weight register would be an
        # unguarded hazard; spills into other AGPRs are ordinary compiler-managed copies)
        wregs = set(m.group(1) for t in loop for m in [re.match(r'^v_mfma\S*\s+\S+,\s*\S+,\s*(a\d+),', t)] if m)
        written = set(m.group(1) for t in loop for m in [re.match(r'^v_accvgpr_write_b32\s+(a\d+),', t)] if m)
        clash = sorted(wregs & written)
        short = 'fwd' if 'fwd' in name else 'bwd'
        sc1 = 'Lb1' in name
        print(f'{short}<{"true" if sc1 else "false"}>: {len(loop)} instructions in the step loop (of {len(insts)} in the kernel)')
        print('   ' + ', '.join(f'{k} {v}' for k, v in mix.most_common(24)))
        nop_cycles = sum(int(t.split()[1]) + 1 for t in loop if t.startswith('s_nop'))
        print(f'   s_nop wait states {nop_cycles}, v_mfma {sum(v for k, v in mix.items() if k.startswith("v_mfma"))}, '
              f'permlane swaps {sum(v for k, v in mix.items() if "permlane" in k)}, dpp adds {mix.get("v_add_f32_dpp", 0)}, '
              f'v_accvgpr_write INTO WEIGHT AGPRS IN THE LOOP {len(clash)} (must be 0: a copy into an AGPR in front of an inline-asm MFMA is an unguarded hazard), '
              f'weight AGPRs read by MFMAs {len(wregs)}, v_accvgpr_write (spills) {mix.get("v_accvgpr_write_b32", 0)}, v_accvgpr_read {mix.get("v_accvgpr_read_b32", 0)}, '
              f'scratch ops {sum(v for k, v in mix.items() if k.startswith("scratch_"))}, ds ops {sum(v for k, v in mix.items() if k.startswith("ds_"))}, global/buffer {sum(v for k, v in mix.items() if k.startswith(("global_", "buffer_")))}')


if __name__ == '__main__':
    main()
