"""GPU tier: HumorLoss (training path; SMPL terms on the HIP body-model kernels, dense adjoint for the mesh term) against the
reference HumorLoss fixture."""
import pytest
import torch

import humor_loss_checks as HL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('name', ['a', 'b', 'c'])
def test_humor_loss_matches_reference(gpu_lib, tmp_path_factory, name):
    root = HL.write_models(str(tmp_path_factory.mktemp('smplh_gender')))
    worst = HL.check_case(name, torch.device('cuda:0'), root)
    print(name, {k: f'{v:.1e}' for k, v in worst.items()})


def test_humor_loss_batch_bound_and_missing_inputs(gpu_lib, tmp_path_factory):
    """The reference's error behaviour: more rows of one gender than smpl_batch_size, and SMPL terms without gender / betas."""
    from humor_amd.humor_loss import HumorLoss
    from oracle import humor_loss_cases as HC
    root = HL.write_models(str(tmp_path_factory.mktemp('smplh_gender')))
    dev = torch.device('cuda:0')
    case = HC.make_case(7, 1)
    pred = {k: (tuple(t.to(dev) for t in v) if isinstance(v, tuple) else v.to(dev)) for k, v in case['pred'].items()}
    gt = {k: v.to(dev) for k, v in case['gt'].items()}
    mod = HumorLoss(smpl_batch_size=2, smplh_path=root, **HC.WEIGHTS)
    with pytest.raises(Exception, match='batch size not large enough'):
        mod(pred, gt, 0, gender=case['gender'], betas=case['betas'].to(dev))
    with pytest.raises(Exception, match='gender and betas'):
        mod(pred, gt, 0)
