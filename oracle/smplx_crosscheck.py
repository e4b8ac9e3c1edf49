"""ORACLE (test infrastructure only).  One-shot cross-check for a machine that HAS the real `smplx` package (pinned ==0.1.28 by
the reference's requirements.txt:11) and, optionally, a licensed SMPL+H model.npz -- neither is available where this repo is
built, which is why the smplx arithmetic of oracle/lbs_restated.py is marked "parity unpinned".

    python -m oracle.smplx_crosscheck [path/to/model.npz] [--gpu]

Without a path it writes the seed-0 synthetic model (humor_amd.synth) and uses that: the arithmetic being checked does not depend
on the model being a real body.  Compares, on the same seeded inputs: smplx.SMPLH (the reference's dependency) vs
oracle/lbs_restated.SMPLHLayer vs oracle/smpl_paper_numpy (fp64) and, with --gpu, humor_amd.BodyModel on cuda:0.
Prints the max abs differences of vertices / joints / input gradients and exits non-zero above 1e-4."""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from humor_amd import synth
    from oracle import lbs_restated as L
    from oracle import smpl_paper_numpy as SP
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    path = args[0] if args else synth.write_smplh_npz(os.path.join(tempfile.mkdtemp(), 'model.npz'), seed=0)
    data = np.load(path, encoding='latin1', allow_pickle=True)

    class DS:
        pass
    ds = DS()
    for k in data.files:
        setattr(ds, k, data[k])
    N, NB = 4, 16
    g = torch.Generator().manual_seed(0)
    r = lambda *s, sc=1.0: sc * torch.randn(*s, generator=g)
    inp = dict(global_orient=r(N, 3, sc=0.5), body_pose=r(N, 63, sc=0.4), betas=r(N, NB), transl=r(N, 3))
    ours = L.SMPLHLayer(data_struct=ds, num_betas=NB, batch_size=N, vertex_ids=L.VERTEX_IDS_SMPLH)
    a = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    out = ours(**a)
    worst = 0.0
    report = lambda name, d: print(f'{name:58s} max|diff| = {d:.3e}') or d
    # independent fp64 derivation
    kin = np.asarray(data['kintree_table'])[0].astype(np.int64)
    model = dict(v_template=data['v_template'], shapedirs=data['shapedirs'], posedirs=data['posedirs'], J_regressor=data['J_regressor'],
                 weights=data['weights'], parents=[-1] + [int(p) for p in kin[1:]])
    for i in range(N):
        pose = np.concatenate([inp['global_orient'][i].numpy(), inp['body_pose'][i].numpy(), np.zeros(90)])
        v, j = SP.smpl_frame(model, pose, inp['betas'][i].numpy(), inp['transl'][i].numpy())
        worst = max(worst, report(f'frame {i}: restatement (fp32) vs SMPL-paper numpy (fp64), vertices', np.abs(out.vertices[i].detach().numpy() - v).max()))
        worst = max(worst, report(f'frame {i}: ... joints', np.abs(out.joints[i, :52].detach().numpy() - j).max()))
    try:
        import smplx
    except ImportError:
        print('smplx is not installed here: the comparison against the real package was SKIPPED (parity stays unpinned)')
        smplx = None
    if smplx is not None:
        ref = smplx.SMPLH(model_path=path, num_betas=NB, batch_size=N, use_pca=False, flat_hand_mean=True, ext='npz')
        b = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
        ro = ref(**b)
        worst = max(worst, report('smplx.SMPLH vs restatement: vertices', (ro.vertices - out.vertices).abs().max().item()))
        worst = max(worst, report('smplx.SMPLH vs restatement: joints', (ro.joints[:, :out.joints.shape[1]] - out.joints).abs().max().item()))
        gw = torch.randn(out.joints.shape, generator=g)
        ga = torch.autograd.grad((out.joints * gw).sum(), list(a.values()))
        gb = torch.autograd.grad((ro.joints[:, :out.joints.shape[1]] * gw).sum(), list(b.values()))
        for k, x, y in zip(a.keys(), ga, gb):
            worst = max(worst, report(f'smplx.SMPLH vs restatement: d/d{k}', ((x - y).abs().max() / y.abs().max().clamp(min=1.0)).item()))
    if '--gpu' in sys.argv:
        from humor_amd.body_model import BodyModel
        dev = torch.device('cuda:0')
        bm = BodyModel(path, num_betas=NB, batch_size=N, use_vtx_selector=True)
        o = bm(root_orient=inp['global_orient'].to(dev), pose_body=inp['body_pose'].to(dev), betas=inp['betas'].to(dev), trans=inp['transl'].to(dev))
        worst = max(worst, report('humor_amd.BodyModel (HIP) vs restatement: vertices', (o.v.cpu() - out.vertices.detach()).abs().max().item()))
        worst = max(worst, report('humor_amd.BodyModel (HIP) vs restatement: joints', (o.Jtr.cpu() - out.joints.detach()).abs().max().item()))
    print('worst:', worst)
    sys.exit(0 if worst < 1e-4 else 1)


if __name__ == '__main__':
    main()
