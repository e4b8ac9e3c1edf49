import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: long-running CPU test')


def pytest_collection_modifyitems(config, items):
    if os.environ.get('HUMOR_AMD_SLOW', '0') == '1':
        return
    skip = pytest.mark.skip(reason='slow SIMT-emulator test: set HUMOR_AMD_SLOW=1')
    for item in items:
        if 'slow' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def smplh_npz(tmp_path_factory):
    """Seed-0 synthetic SMPL+H model file (same bytes on every machine)."""
    from humor_amd import synth
    return synth.write_smplh_npz(str(tmp_path_factory.mktemp('smplh') / 'model.npz'), seed=0)


@pytest.fixture(scope='session')
def smplh_struct(smplh_npz):
    data = np.load(smplh_npz)

    class DS:
        pass
    ds = DS()
    for k in data.files:
        setattr(ds, k, data[k])
    return ds


@pytest.fixture(scope='session')
def emu_lib():
    """The kernel sources built for the host SIMT emulator (tests/simt_emu) -- CPU test tier only."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'simt_emu'))
    import build as emu_build
    from humor_amd import _lib
    return _lib.load(emu_build.build(), emulator=True)


_real_empty = torch.empty


# HUMOR_AMD_TEST_POISON_VALUE=big: 1e30 instead of NaN.  The hardware's max / min drop NaN operands (ReLU(NaN) = 0), so a NaN that is read by
# mistake can be squashed before it reaches an output; a huge finite value is not.  Both passes are run at the end of a round (profiles/r06_final).
_POISON_BIG = os.environ.get('HUMOR_AMD_TEST_POISON_VALUE', 'nan') == 'big'
_POISON_FILL = 1.0e30 if _POISON_BIG else float('nan')


def _poisoned_empty(*args, **kwargs):
    t = _real_empty(*args, **kwargs)
    if t.is_cuda and t.is_floating_point():
        t.fill_(_POISON_FILL)
    return t


@pytest.fixture(autouse=True)
def _poisoned_allocations(request, monkeypatch):
    """GPU tier: every torch.empty of a floating-point device tensor comes back filled with NaN, so that a kernel that reads a word of a
    workspace / stash / output it did not write first fails on EVERY box, not only on the one whose allocator hands out unlucky memory
    (round 5: green on the builder's lease, NaN on the driver's).  The library's own state gets the same treatment through
    HUMOR_AMD_CU_POISON (gpu_lib).  HUMOR_AMD_TEST_POISON=0 switches both off."""
    if 'gpu' in request.keywords and os.environ.get('HUMOR_AMD_TEST_POISON', '1') != '0':
        monkeypatch.setattr(torch, 'empty', _poisoned_empty)
    yield


@pytest.fixture(scope='session')
def gpu_lib():
    from humor_amd import _lib
    assert torch.cuda.is_available(), 'GPU tests need a visible MI355X'
    if os.environ.get('HUMOR_AMD_TEST_POISON', '1') != '0':
        # LDS and the vector registers of every CU hold NaN patterns when a compute entry point (and every persistent roll-out launch inside
        # one) starts: humor_amd/csrc/debug.hip.  Child processes of the tests (bench ranks) inherit it.
        os.environ.setdefault('HUMOR_AMD_CU_POISON', '0x7149f2ca' if _POISON_BIG else '1')      # (0x7149f2ca = 1e30f)
    lib = _lib.get_lib()
    arch = lib.device_arch(0)
    assert arch.startswith('gfx950'), f'expected gfx950, found {arch}'
    return lib


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
