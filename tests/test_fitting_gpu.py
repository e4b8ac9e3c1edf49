"""GPU tier: the fitting objectives (stage 1/2/3 closures) and short MotionOptimizer runs against fixtures produced by
the reference MotionOptimizer."""
import pytest
import torch

import fitting_checks as FC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    return torch.device('cuda:0')


@pytest.mark.parametrize('kind', ['amass', 'rgb'])
def test_stage_objectives_match_reference(gpu_lib, dev, smplh_npz, kind):
    FC.check_objectives(gpu_lib, dev, smplh_npz, kind)


@pytest.mark.parametrize('kind', ['amass', 'rgb'])
def test_short_run_matches_reference(gpu_lib, dev, smplh_npz, kind):
    FC.check_short_run(gpu_lib, dev, smplh_npz, kind)
