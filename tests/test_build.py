"""CPU tier: static checks of the compiled persistent roll-out kernels (hipcc cross-compiles gfx950 without a GPU).

The inline-asm MFMAs of rollout_persist.hip are opaque to the compiler's hazard recogniser.  What keeps them correct is the register plan:
weights that are "a" operands must LIVE in AGPRs for the whole launch.  When the plan over-subscribes a register half the allocator
keeps a weight elsewhere and copies it into an AGPR right in front of the MFMA that reads it (v_accvgpr_write -> MFMA read without the
wait states the hardware needs): wrong gradients, measured in round 4 with 232 AGPR weights in the adjoint.  The kernels only run on a
GPU box, so this tier holds the compiled code to: no AGPR writes and no scratch traffic inside the step loops."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_persistent_step_loops_have_no_agpr_copies_or_scratch():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_census.py')], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels = [l for l in out.stdout.split('\n') if l.startswith(('fwd<', 'bwd<'))]
    assert len(kernels) == 4, out.stdout
    details = [l for l in out.stdout.split('\n') if 'IN THE LOOP' in l]
    assert all('weight AGPRs read by MFMAs' in d and int(d.split('weight AGPRs read by MFMAs')[1].split()[0].rstrip(',')) >= 200 for d in details), details
    assert len(details) == 4
    for head, d in zip(kernels, details):
        n_write = int(d.split('IN THE LOOP')[1].split()[0])
        n_scratch = int(d.split('scratch ops')[1].split()[0].rstrip(','))
        assert n_write == 0, (head, 'v_accvgpr_write in the step loop', n_write)
        assert n_scratch == 0, (head, 'scratch traffic in the step loop', n_scratch)
        n_inst = int(head.split(':')[1].split()[0])
        assert 2000 < n_inst < 6000, head          # the loop was found (a changed code shape would make the census meaningless)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason='needs hipcc')
def test_pipelined_kernels_have_no_agpr_traffic_or_scratch():
    """rollout_pipe.inc (the layer-parallel kernels for more than 32 sequences): every role loop of both kernels keeps all 256 AGPRs as
    MFMA weight operands and never touches them otherwise -- no v_accvgpr_write (a copy into an AGPR in front of an inline-asm MFMA is an
    unguarded hazard), no v_accvgpr_read, no v_mov (the VGPR-class weights carry no wait states in front of their MFMAs, which is only
    correct while nothing rewrites them), no scratch, no spills."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'isa_census_pipe.py')], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    heads = [l for l in out.stdout.split('\n') if l.startswith('pipe ')]
    assert len(heads) == 4 and all('spills 0' in h for h in heads), out.stdout
    loops = [l for l in out.stdout.split('\n') if 'role loop' in l]
    assert len(loops) >= 8, out.stdout
    num = lambda l, key: int(l.split(key)[1].split()[0].rstrip(','))
    for l in loops:
        assert num(l, 'v_accvgpr_write INTO WEIGHT AGPRS IN THE LOOP') == 0 and num(l, 'v_accvgpr_write (any)') == 0, l
        assert num(l, 'v_accvgpr_read') == 0 and num(l, 'scratch ops') == 0 and num(l, 'v_mov') == 0, l
        assert num(l, 'v_mfma') >= 250, l
    assert any(num(l, 'weight AGPRs read by MFMAs') == 256 for l in loops)
