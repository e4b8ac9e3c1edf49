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
