"""ORACLE (test infrastructure only).  Generates tests/golden/*.npz by running the UNMODIFIED reference Python
(/root/reference/humor, imported through oracle/ref_loader.py) on seeded synthetic inputs.  Runs only inside the
build container (the reference tree is absent on the GPU box); the resulting small fixtures are committed.

  python -m oracle.make_golden

Fixtures
  smpl_bodymodel.npz   reference BodyModel.forward (body_model.py:72-115; smplx arithmetic = oracle/lbs_restated.py)
                       on the seed-0 synthetic SMPL+H model: inputs, selected outputs, input gradients
  rollout.npz          reference HumorModel.roll_out (humor_model.py:785-1017) with seed-0 synthetic weights:
                       initial state, z sequence, world-frame outputs, prior (mean, var), input gradients
  rotations.npz        reference batch_rodrigues / rotation_matrix_to_angle_axis (transforms.py:139-170, 243-389)
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from humor_amd import synth            # noqa: E402
from oracle import ref_loader          # noqa: E402
from oracle.lbs_restated import batch_rodrigues   # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
KEEP_VERTS = list(range(0, 6890, 53))    # 130 vertices kept from the dense output to keep the fixture small


def canonical_state(B, gen):
    """A plausible canonical-frame initial input state [B,339] (root at xy origin)."""
    r = lambda *s: torch.randn(*s, generator=gen)
    trans = torch.cat([torch.zeros(B, 2), 0.9 + 0.1 * r(B, 1)], 1)
    R_root = batch_rodrigues(0.3 * r(B, 3)).reshape(B, 9)
    R_body = batch_rodrigues(0.3 * r(B * 21, 3)).reshape(B, 189)
    joints = 0.3 * r(B, 66)
    joints[:, :2] = 0
    return torch.cat([trans, 0.3 * r(B, 3), R_root, 0.3 * r(B, 3), R_body, joints, 0.3 * r(B, 66)], 1)


def main():
    R = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)

    # ---- SMPL through the reference BodyModel wrapper ---------------------------------------------
    with tempfile.TemporaryDirectory() as td:
        path = synth.write_smplh_npz(os.path.join(td, 'model.npz'), seed=0)
        N = 4
        g = torch.Generator().manual_seed(11)
        root = (0.5 * torch.randn(N, 3, generator=g)).requires_grad_(True)
        body = (0.4 * torch.randn(N, 63, generator=g)).requires_grad_(True)
        betas = torch.randn(N, 16, generator=g).requires_grad_(True)
        trans = torch.randn(N, 3, generator=g).requires_grad_(True)
        bm = R.body_model.BodyModel(path, num_betas=16, batch_size=N, use_vtx_selector=True)
        out = bm(root_orient=root, pose_body=body, betas=betas, trans=trans)
        gJ = torch.randn(out.Jtr.shape, generator=g)
        gV = torch.randn(N, len(R.bm_utils.KEYPT_VERTS), 3, generator=g)
        loss = (out.Jtr * gJ).sum() + (out.v[:, R.bm_utils.KEYPT_VERTS] * gV).sum()
        grads = torch.autograd.grad(loss, [root, body, betas, trans])
        np.savez_compressed(os.path.join(OUT, 'smpl_bodymodel.npz'),
                            root_orient=root.detach().numpy(), pose_body=body.detach().numpy(),
                            betas=betas.detach().numpy(), trans=trans.detach().numpy(),
                            Jtr=out.Jtr.detach().numpy(), keep_verts=np.array(KEEP_VERTS),
                            v_keep=out.v[:, KEEP_VERTS].detach().numpy(),
                            v_keypt=out.v[:, R.bm_utils.KEYPT_VERTS].detach().numpy(),
                            keypt_verts=np.array(R.bm_utils.KEYPT_VERTS), gJ=gJ.numpy(), gV=gV.numpy(),
                            g_root=grads[0].numpy(), g_body=grads[1].numpy(), g_betas=grads[2].numpy(),
                            g_trans=grads[3].numpy(), model_seed=0)

    # ---- roll-out through the reference HumorModel ------------------------------------------------
    sd = synth.humor_state_dict(seed=0)
    hm = R.humor_model.HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48,
                                  model_data_config='smpl+joints+contacts', steps_in=1)
    hm.load_state_dict(sd)
    hm.eval()
    B, S = 4, 10
    g = torch.Generator().manual_seed(5)
    past0 = canonical_state(B, g).requires_grad_(True)
    z = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    names = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel']
    dims = [3, 3, 9, 3, 189, 66, 66]
    d, o = {}, 0
    for k, n in zip(names, dims):
        d[k] = past0[:, o:o + n].unsqueeze(1)
        o += n
    pred, (pm, pv) = hm.roll_out(None, d, S, z_seq=z, return_prior=True)
    world = torch.cat([pred[k] for k in names + ['contacts']], 2)
    gw = torch.randn(world.shape, generator=g)
    gm = torch.randn(pm.shape, generator=g)
    gvv = torch.randn(pv.shape, generator=g)
    grads = torch.autograd.grad((world * gw).sum() + (pm * gm).sum() + (pv * gvv).sum(), [past0, z])
    # what rollout_latent_motion derives (motion_optimizer.py:959-998)
    aa_root = R.transforms.rotation_matrix_to_angle_axis(pred['root_orient'].reshape(-1, 3, 3)).reshape(B, S, 3)
    aa_body = R.transforms.rotation_matrix_to_angle_axis(pred['pose_body'].reshape(-1, 3, 3)).reshape(B, S, 63)
    np.savez_compressed(os.path.join(OUT, 'rollout.npz'), past0=past0.detach().numpy(), z=z.detach().numpy(),
                        world=world.detach().numpy(), prior_mu=pm.detach().numpy(), prior_var=pv.detach().numpy(),
                        gw=gw.numpy(), gm=gm.numpy(), gv=gvv.numpy(), g_past0=grads[0].numpy(), g_z=grads[1].numpy(),
                        aa_root=aa_root.detach().numpy(), aa_body=aa_body.detach().numpy(), weight_seed=0)

    # ---- rotation conversions ----------------------------------------------------------------------
    g = torch.Generator().manual_seed(7)
    aa = torch.cat([1.2 * torch.randn(200, 3, generator=g), torch.zeros(4, 3),
                    torch.tensor([[np.pi, 0, 0], [0, 3.1, 0], [1e-4, 0, 0], [0, 0, 2.9]], dtype=torch.float32)], 0)
    aa = aa.requires_grad_(True)
    Rm = R.transforms.batch_rodrigues(aa)
    gR = torch.randn(Rm.shape, generator=g)
    g_aa = torch.autograd.grad((Rm * gR).sum(), aa)[0]
    Rin = Rm.detach().clone().requires_grad_(True)
    back = R.transforms.rotation_matrix_to_angle_axis(Rin)
    gb = torch.randn(back.shape, generator=g)
    g_Rin = torch.autograd.grad((back * gb).sum(), Rin)[0]
    np.savez_compressed(os.path.join(OUT, 'rotations.npz'), aa=aa.detach().numpy(), R=Rm.detach().numpy(),
                        gR=gR.numpy(), g_aa=g_aa.numpy(), aa_back=back.detach().numpy(), gb=gb.numpy(),
                        g_Rin=g_Rin.numpy())
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == '__main__':
    main()
