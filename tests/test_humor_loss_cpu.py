"""CPU tier: HumorLoss host logic.  (1) the non-SMPL terms bit-for-bit against the live reference HumorLoss (build container only);
(2) the SMPL terms with the kernel sources running on the host SIMT emulator, against the reference fixture (case 'c': joints +
key-vertex terms, subset kernels; the dense mesh cases run in the GPU tier)."""
import importlib

import pytest
import torch

import humor_loss_checks as HL
from oracle import humor_loss_cases as HC
from oracle import ref_loader

needs_ref = pytest.mark.skipif(not ref_loader.available(), reason='reference tree only exists in the build container')


@needs_ref
@pytest.mark.parametrize('cycle', [False, True])
def test_non_smpl_terms_bit_exact_vs_live_reference(cycle):
    from humor_amd.humor_loss import HumorLoss
    ref_loader.load()
    ref_mod = importlib.import_module('losses.humor_loss')
    w = dict(HC.WEIGHTS, smpl_joint_loss=0.0, smpl_mesh_loss=0.0, smpl_joint_consistency_loss=0.0, smpl_vert_consistency_loss=0.0)
    if cycle:
        w.update(kl_loss_anneal_end=0, kl_loss_cycle_len=30)
    case = HC.make_case(11, 5)
    cpu = torch.device('cpu')
    a = HC.evaluate(ref_mod.HumorLoss(**w), case, cpu)
    b = HC.evaluate(HumorLoss(**w), case, cpu)
    assert sorted(a) == sorted(b)
    for k in a:
        if k.startswith('grad_'):
            assert (a[k] == b[k]).all(), k
        else:
            assert a[k] == b[k], (k, a[k], b[k])


def test_smpl_terms_on_the_emulator_match_reference_fixture(emu_lib, tmp_path_factory):
    root = HL.write_models(str(tmp_path_factory.mktemp('smplh_gender')))
    worst = HL.check_case('c', torch.device('cpu'), root, lib=emu_lib)
    print({k: f'{v:.1e}' for k, v in worst.items()})
