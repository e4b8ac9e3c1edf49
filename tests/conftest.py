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


@pytest.fixture(scope='session')
def gpu_lib():
    from humor_amd import _lib
    assert torch.cuda.is_available(), 'GPU tests need a visible MI355X'
    lib = _lib.get_lib()
    arch = lib.device_arch(0)
    assert arch.startswith('gfx950'), f'expected gfx950, found {arch}'
    return lib


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
