"""Shared roll-out / rotation parity checks (emulator tier on CPU, gfx950 build on the GPU)."""
import numpy as np
import torch

from conftest import golden
from humor_amd import ops, synth
from humor_amd.humor_model import HumorModel
from oracle import humor_restated as H
from oracle import lbs_restated as L

KEYS = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel', 'contacts']
FWD_TOL = 1e-4      # north_star: fp32 latents / states within 1e-4
GRAD_RTOL = 1e-3    # relative to the largest reference gradient entry (chain of S steps)


def canonical_state(B, gen):
    r = lambda *s: torch.randn(*s, generator=gen)
    trans = torch.cat([torch.zeros(B, 2), 0.9 + 0.1 * r(B, 1)], 1)
    R_root = L.batch_rodrigues(0.3 * r(B, 3)).reshape(B, 9)
    R_body = L.batch_rodrigues(0.3 * r(B * 21, 3)).reshape(B, 189)
    joints = 0.3 * r(B, 66)
    joints[:, :2] = 0
    return torch.cat([trans, 0.3 * r(B, 3), R_root, 0.3 * r(B, 3), R_body, joints, 0.3 * r(B, 66)], 1)


def make_model(lib, device, seed=0, weight_scale=1.0, contractive=False):
    sd = synth.contractive_state_dict(seed) if contractive else synth.humor_state_dict(seed=seed, weight_scale=weight_scale)
    hm = HumorModel(in_rot_rep='mat', out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1,
                    _lib_override=lib)
    hm.load_state_dict(sd)
    return hm.to(device).eval(), sd


def world_of(out):
    return torch.cat([out[k] for k in KEYS], 2)


def oracle_grads(sd, past, z, objective, dtype=torch.float64, probe=None):
    """(world, pm, pv, [dL/dpast, dL/dz]) of the oracle in `dtype`, optionally under a KinkProbe (recorded / forced ReLU branches)."""
    p = past.detach().to(dtype).clone().requires_grad_(True)
    zz = z.detach().to(dtype).clone().requires_grad_(True)
    sdd = {k: v.to(dtype) for k, v in sd.items()}
    if probe is None:
        w, (pm, pv) = H.roll_out(sdd, p, zz)
    else:
        with probe:
            w, (pm, pv) = H.roll_out(sdd, p, zz)
    g = torch.autograd.grad(objective(w, pm, pv), [p, zz])
    return w.detach(), pm.detach(), pv.detach(), [x.detach() for x in g]


def _perturb_rows(v, amp, gen):
    """v (float64, [B, ...]) with every entry moved by a uniform relative amount in [-amp[row], amp[row]]."""
    a = amp.reshape(-1, *([1] * (v.dim() - 1)))
    return v * (1.0 + (torch.rand(v.shape, generator=gen, dtype=torch.float64) * 2 - 1) * a)


def kink_aware_grad_check(sd, past, z, objective, world_gpu, g_gpu, grad_rtol=GRAD_RTOL, max_units=6, npert=6):
    """The gradients of a roll-out against the fp64 oracle at the FLAT bar `grad_rtol` (relative to the largest reference entry), sequence by
    sequence.  A sequence that misses the bar on the oracle's own ReLU branch is examined from the REFERENCE side, never from the kernel's:
      (a) kinks: the units of that sequence whose GroupNorm output lies closer to zero than the kernel's forward deviation can resolve
          (tau = max(4e-6, 8 x the sequence's largest state deviation from fp64)) are the ones an fp32 evaluation may legitimately put on
          the other side; the sequence passes if the kernel's gradient equals the oracle's -- at the same flat bar -- for ONE on / off
          assignment of those units (<= 2^max_units batched oracle evaluations; the sequences are independent);
      (b) conditioning: the gradient of a sequence can follow the oracle's no better than the oracle's own gradient holds still when its
          inputs move by as much as the kernel's forward values are off (a deviation the forward bar allows): relative input perturbations
          of amplitude max(1 ulp, 2 x the sequence's state deviation) are applied to the fp64 oracle (npert draws); where ITS gradient moves
          by >= grad_rtol / 4 the sequence is ill-conditioned (heading within 1e-2 rad of the acos singularity: the gradient moves 1e4 x as
          much as the states) -- reported as `unstable`, it must stay within 30 x the bar, and at most max(1, 10 %) of the sequences may be.
    Anything else fails.  Returns the report."""
    B = past.shape[0]
    w64, _, _, g64 = oracle_grads(sd, past, z, objective)
    scale = [max(1.0, a.abs().max().item()) for a in g64]

    def row_err(ref, got):
        return torch.stack([(a.double() - b.double()).abs().reshape(B, -1).amax(1) / sc for a, b, sc in zip(ref, got, scale)]).amax(0)
    gg = [x.detach().cpu().double() for x in g_gpu]
    e_nat = row_err(g64, gg)
    failing = (e_nat >= grad_rtol).nonzero().flatten().tolist()
    report = {'natural_max': e_nat.max().item(), 'rows_on_other_branch': [], 'unstable': [], 'unresolved': []}
    if not failing:
        return report
    dev = (world_gpu.detach().cpu().double() - w64).abs().reshape(B, -1).amax(1)
    tau = torch.zeros(B, dtype=torch.float64)
    tau[failing] = torch.clamp(8.0 * dev[failing], min=4e-6)
    probe = H.KinkProbe(tau=tau)
    oracle_grads(sd, past, z, objective, probe=probe)
    units = {b: [] for b in failing}
    for site, r, c, y in probe.near:
        units[r].append((site, c, y))
    best = {b: e_nat[b].item() for b in failing}
    n_eval = max((1 << min(len(u), max_units)) for u in units.values())
    widths = {}
    for k in range(1, n_eval):
        force = {}
        for b in failing:
            u = sorted(units[b], key=lambda t: abs(t[2]))[:max_units]        # (the units nearest to zero first)
            kb = k % (1 << len(u)) if u else 0
            for j, (site, c, y) in enumerate(u):
                if (kb >> j) & 1:                                          # flip this unit against its natural branch
                    if site not in force:
                        if site not in widths:
                            li = site[2]
                            widths[site] = sd[f"{'decoder' if site[0] == 'dec' else 'prior_net'}.net.{3 * li}.weight"].shape[0]
                        force[site] = torch.full((B, widths[site]), -1, dtype=torch.int8)
                    force[site][b, c] = 0 if y > 0 else 1
        _, _, _, gk = oracle_grads(sd, past, z, objective, probe=H.KinkProbe(force=force))
        ek = row_err(gk, gg)
        for b in failing:
            best[b] = min(best[b], ek[b].item())
    left = [b for b in failing if best[b] >= grad_rtol]
    for b in failing:
        if b not in left:
            report['rows_on_other_branch'].append((b, round(e_nat[b].item(), 6), round(best[b], 6), len(units[b]), float(tau[b])))
    if left:
        # (b) how far does the ORACLE's gradient move when its inputs move as much as the kernel's forward values are off?
        wmax = w64.abs().reshape(B, -1).amax(1).clamp(min=1.0)
        amp = torch.clamp(2.0 * dev / wmax, min=2.0 ** -23)
        moved = torch.zeros(B, dtype=torch.float64)
        gen = torch.Generator().manual_seed(12345)
        for _ in range(npert):
            _, _, _, gp = oracle_grads(sd, _perturb_rows(past.detach().double(), amp, gen), _perturb_rows(z.detach().double(), amp, gen), objective)
            moved = torch.maximum(moved, row_err(g64, gp))
        for b in left:
            rec = (b, round(e_nat[b].item(), 6), round(best[b], 6), len(units[b]), float(tau[b]), round(moved[b].item(), 6))
            if moved[b].item() >= grad_rtol / 4 and best[b] < 30 * grad_rtol:
                report['unstable'].append(rec)
            else:
                report['unresolved'].append(rec)
    assert not report['unresolved'], ('gradient differs from the fp64 oracle on every reachable ReLU branch of a sequence whose oracle gradient '
                                      'holds still (row, natural error, best branch error, near-kink units, tau, oracle movement under input perturbation)', report)
    assert len(report['unstable']) <= max(1, B // 10), ('too many sequences beyond the flat bar, even if ill-conditioned', report)
    return report


def check_rollout(lib, device, B, S, seed=0, with_prior=True, weight_scale=1.0, fwd_tol=FWD_TOL, grad_rtol=GRAD_RTOL, cond_aware=False,
                  contractive=False):
    """One roll-out forward + backward against the oracle at FLAT bars: every state / prior output within fwd_tol (north_star: 1e-4) of the
    fp64 oracle and of the fp32 oracle, every gradient entry within grad_rtol x the largest reference entry of the fp64 oracle's gradient on a
    ReLU branch the reference side cannot tell from the kernel's (kink_aware_grad_check: the natural branch unless the sequence has a unit
    within rounding of its kink).  cond_aware (long chains of the random, non-contractive network only): the fp32 oracle's own distance from
    fp64 is the yardstick instead."""
    if contractive:
        hm, sd = make_model(lib, device, seed=seed, contractive=True)
    else:
        hm, sd = make_model(lib, device, seed=seed, weight_scale=weight_scale)
    g = torch.Generator().manual_seed(seed + 5)
    past_c = canonical_state(B, g).requires_grad_(True)
    z_c = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    past = past_c.detach().to(device).requires_grad_(True)
    z = z_c.detach().to(device).requires_grad_(True)
    res = hm.roll_out(past, None, S, z_seq=z, return_prior=with_prior)
    out, (pm, pv) = res if with_prior else (res, (None, None))
    world = world_of(out)
    w_ref, (pm_r, pv_r) = H.roll_out(sd, past_c, z_c)
    err = (world.detach().cpu() - w_ref).abs().max().item()
    w64, (pm64, pv64) = H.roll_out({k: v.double() for k, v in sd.items()}, past_c.detach().double(), z_c.detach().double())
    e_gpu = (world.detach().cpu().double() - w64).abs().max().item()
    if cond_aware:
        # long chains amplify fp32 rounding (any two fp32 implementations drift apart): judge both against fp64
        e_cpu = (w_ref.detach().double() - w64).abs().max().item()
        assert e_gpu < max(fwd_tol, 3.0 * e_cpu), (e_gpu, e_cpu)
        fwd_tol = max(fwd_tol, 3.0 * e_cpu)
    else:
        # flat bar per sequence against fp64; a sequence whose fp32 ORACLE evaluation already uses up more than a third of the bar (the
        # random network amplifies rounding by ~1.2 x per step: at 12 steps the oracle's own fp32 run sits at 0.95e-4 for seed 32, and old
        # and new kernels land on either side of 1e-4 by turns -- profiles/r06_accuracy) gets 3 x the oracle's own distance instead
        e_rows = (world.detach().cpu().double() - w64).abs().reshape(B, -1).amax(1)
        o_rows = (w_ref.detach().double() - w64).abs().reshape(B, -1).amax(1)
        bar = torch.where(o_rows > fwd_tol / 3, 3.0 * o_rows, torch.full_like(o_rows, fwd_tol))
        assert (e_rows < bar).all(), ('forward', e_rows.max().item(), (e_rows / bar).max().item(), o_rows.max().item())
        if (o_rows > fwd_tol / 3).any():
            print(f'check_rollout {B}x{S} seed {seed}: {int((o_rows > fwd_tol / 3).sum())} sequence(s) judged at 3 x the fp32 oracle\'s own distance '
                  f'from fp64 (largest {o_rows.max().item():.1e}); kernel {e_rows.max().item():.1e}')
        fwd_tol = max(fwd_tol, bar.max().item())
    gw = torch.randn(w_ref.shape, generator=g)
    gm, gv = torch.randn(pm_r.shape, generator=g), torch.randn(pv_r.shape, generator=g)

    def objective(w, m, v):
        loss = (w * gw.to(w)).sum()
        if with_prior:
            loss = loss + (m * gm.to(m)).sum() + (v * gv.to(v)).sum()
        return loss
    if with_prior:
        assert (pm.detach().cpu().double() - pm64).abs().max().item() < fwd_tol
        assert ((pv.detach().cpu().double() - pv64).abs() / pv64.abs().clamp(min=1.0)).max().item() < fwd_tol
    g_our = torch.autograd.grad(objective(world, pm, pv), [past, z])
    if not cond_aware:
        rep = kink_aware_grad_check(sd, past_c, z_c, objective, world, g_our, grad_rtol=grad_rtol)
        if rep['rows_on_other_branch'] or rep['unstable']:
            print(f'check_rollout {B}x{S} seed {seed}: sequences on another ReLU branch than the fp64 oracle '
                  f'(row, natural error, error on the matching branch, near-kink units, tau): {rep["rows_on_other_branch"]}; '
                  f'ill-conditioned sequences (.., the oracle\'s own movement under input perturbations of the kernel\'s forward deviation): {rep["unstable"]}')
    else:
        # the adjoint of a long chain is amplified like the forward error: judge both fp32 gradients against fp64
        g_ref = torch.autograd.grad(objective(w_ref, pm_r, pv_r), [past_c, z_c])
        _, _, _, g64 = oracle_grads(sd, past_c, z_c, objective)
        for i, (name, a, b) in enumerate(zip(('g_past_in0', 'g_z'), g_ref, g_our)):
            scale = max(1.0, a.abs().max().item())
            e_g = (g64[i] - b.cpu().double()).abs().max().item()
            e_c = (g64[i] - a.double()).abs().max().item()
            assert e_g < max(grad_rtol * scale, 3.0 * e_c), (name, e_g, e_c, scale)
    # contact labels (the only thresholded output) must be bit-exact away from the decision boundary (G11)
    logits_ref = w_ref[:, :, 339:348].detach()
    lab = (torch.sigmoid(world[:, :, 339:348].detach().cpu()) > 0.5)
    lab_ref = torch.sigmoid(logits_ref) > 0.5
    far = logits_ref.abs() > 1e-4
    assert (lab[far] == lab_ref[far]).all()
    return err


def check_rollout_conditioned(lib, device, B=32, S=59, seed=3):
    """Full BASELINE length (59 steps).  With random-init weights the autoregressive chain amplifies rounding error by
    ~1e4 (measured: the reference-style fp32 CPU evaluation drifts 6e-2 from an fp64 evaluation of the same chain at
    step 58, tests/diagnostics/rollout_diag.py), so a fixed 1e-4 bound against the fp32 oracle is meaningless there.  Criterion:
    at every step the HIP result must be as close to the fp64 evaluation as the fp32 oracle is (factor 4 + a 1e-5
    relative floor), and within 1e-4 absolute over the first steps where the chain is still well conditioned."""
    hm, sd = make_model(lib, device, seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    past, z = canonical_state(B, g), torch.randn(B, S, 48, generator=g)
    out, (pm, pv) = hm.roll_out(past.to(device), None, S, z_seq=z.to(device), return_prior=True)
    world = world_of(out).detach().cpu().double()
    w32, _ = H.roll_out(sd, past, z)
    w64, _ = H.roll_out({k: v.double() for k, v in sd.items()}, past.double(), z.double())
    e_gpu = (world - w64).abs().amax(dim=(0, 2))
    e_cpu = (w32.double() - w64).abs().amax(dim=(0, 2))
    scale = w64.abs().amax(dim=(0, 2))
    assert e_gpu[:5].max().item() < 1e-4
    running = torch.cummax(e_cpu, dim=0)[0]
    bound = 4.0 * running + 1e-5 * scale
    assert (e_gpu <= bound).all(), (e_gpu / bound).max().item()
    assert torch.isfinite(world).all() and torch.isfinite(pv).all()


def per_seq_rel(a, b):
    """max |a-b| per sequence (leading index), relative to max(1, max|b|) of the whole tensor."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).reshape(a.shape[0], -1).max(axis=1) / max(1.0, np.abs(b).max())


KINK_RTOL = 5e-2     # sequences whose REFERENCE gradient is flagged unstable under fp32-level changes (ReLU kink crossing)


def assert_grad(name, got, ref, stable, rtol=GRAD_RTOL, allow=0):
    """Flat relative bar on every sequence the fixture marks stable; the flagged ones (the reference's own gradient moves by
    more than 2e-4 under an fp64 re-evaluation / 1-ulp input perturbations, oracle/make_golden_long.py) get the kink bar.
    allow: sequences flagged stable that may still sit on a kink the finite set of perturbations did not reach (only used at batch
    sizes where the flags are computed from a handful of perturbations of hundreds of sequences); they stay under the kink bar."""
    e = per_seq_rel(got, ref)
    stable = np.asarray(stable, dtype=bool)
    over = int((e[stable] > rtol).sum())
    assert over <= allow, (name, over, allow, e.tolist(), stable.tolist())
    assert (e <= KINK_RTOL).all(), (name, e.tolist())
    ok = stable & (e <= rtol)
    return float(e[ok].max()) if ok.any() else 0.0


def check_rollout_long(lib, device, name):
    """Reference HumorModel.roll_out at a BASELINE length ('c4': 59 steps, 'c3': 89, 'c5': 119) with the well-conditioned
    synthetic weights: EVERY step's state and prior output within the flat 1e-4, gradients within 1e-3 (relative)."""
    from oracle import closure_cases as CC
    gd = golden('rollout_long.npz')
    pre = f'ro_{name}_'
    hm, _ = make_model(lib, device, seed=int(gd['weight_seed']), contractive=True)
    past = torch.tensor(gd[pre + 'past0']).to(device).requires_grad_(True)
    z = torch.tensor(gd[pre + 'z']).to(device).requires_grad_(True)
    S = z.shape[1]
    out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
    world = world_of(out)
    e_w = np.abs(world.detach().cpu().numpy() - gd[pre + 'world']).max(axis=(0, 2))       # per step
    e_m = np.abs(pm.detach().cpu().numpy() - gd[pre + 'prior_mu']).max(axis=(0, 2))
    e_v = np.abs(pv.detach().cpu().numpy() - gd[pre + 'prior_var']).max(axis=(0, 2))
    assert e_w.max() < FWD_TOL and e_m.max() < FWD_TOL and e_v.max() < FWD_TOL, (name, e_w.max(), e_m.max(), e_v.max())
    gw, gm, gv = (CC.det_weights(t.shape, ph).to(device) for t, ph in ((world, 0.1), (pm, 0.2), (pv, 0.3)))
    g0, gz = torch.autograd.grad((world * gw).sum() + (pm * gm).sum() + (pv * gv).sum(), [past, z])
    stable = gd[pre + 'stable']
    eg0 = assert_grad('g_past0', g0.cpu().numpy(), gd[pre + 'g_past0'], stable)
    egz = assert_grad('g_z', gz.cpu().numpy(), gd[pre + 'g_z'], stable)
    lab = torch.sigmoid(world[:, :, 339:348].detach().cpu()) > 0.5
    logits_ref = torch.tensor(gd[pre + 'world'][:, :, 339:348])
    far = logits_ref.abs() > 1e-4
    assert (lab[far] == (torch.sigmoid(logits_ref) > 0.5)[far]).all()
    return dict(world=float(e_w.max()), prior_mu=float(e_m.max()), prior_var=float(e_v.max()), g_past0=eg0, g_z=egz)


def check_rollout_full_tiles(lib, device, B, S, seed=0):
    """Full row tiles at the BASELINE sizes (32 x 59 = the metric's batch, one complete 32-row tile; 256 x 119 = C5, eight tiles and
    the finishing-pass policy) against the restated oracle (humor_restated.roll_out, bit-pinned to the reference's roll_out in the CPU
    tier) with the well-conditioned synthetic weights: flat 1e-4 on every step's state / prior output, 1e-3 relative on the gradients
    of every sequence whose ORACLE gradient is stable under an fp64 re-evaluation (kink flags as in oracle/make_golden_long.py)."""
    from oracle import closure_cases as CC
    hm, sd = make_model(lib, device, seed=seed, contractive=True)
    g = torch.Generator().manual_seed(1000 + B + S)
    past_c = canonical_state(B, g).requires_grad_(True)
    z_c = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    w_ref, (pm_r, pv_r) = H.roll_out(sd, past_c, z_c)
    gw, gm, gv = CC.det_weights(w_ref.shape, 0.1), CC.det_weights(pm_r.shape, 0.2), CC.det_weights(pv_r.shape, 0.3)
    g_ref = torch.autograd.grad((w_ref * gw).sum() + (pm_r * gm).sum() + (pv_r * gv).sum(), [past_c, z_c])
    # Kink flags: which sequences have a ReLU unit within fp32 rounding of its kink is a property of the inputs and weights, and
    # finding them takes many oracle evaluations (the flagged set grows with the number of 1-ulp perturbations tried and saturates:
    # 34 -> 56 -> 68 -> 69 sequences of 256 x 119 after fp64 / 4 / 8 / 12 perturbations).  oracle/make_kink_flags.py computes the union
    # over an fp64 re-evaluation and 24 perturbations once in the build container -> tests/golden/rollout_kink_flags.npz; the oracle
    # gradient itself is recomputed here.  2 % of the unflagged sequences may still be such cases (they stay under the kink bar).
    flags = golden('rollout_kink_flags.npz')
    assert seed == 0 and f'{B}x{S}' in flags.files, 'kink flags exist for the committed cases only (oracle/make_kink_flags.py)'
    stable = flags[f'{B}x{S}'].astype(bool)
    with torch.no_grad():
        w64 = H.roll_out({k: v.double() for k, v in sd.items()}, past_c.detach().double(), z_c.detach().double())[0]
    drift = (w_ref.detach().double() - w64).abs().max().item()
    assert drift < 2e-5, drift            # the chain itself is well conditioned
    past = past_c.detach().to(device).requires_grad_(True)
    z = z_c.detach().to(device).requires_grad_(True)
    out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
    world = world_of(out)
    e_w = (world.detach().cpu() - w_ref.detach()).abs().amax(dim=(0, 2))
    e_m = (pm.detach().cpu() - pm_r.detach()).abs().max().item()
    e_v = (pv.detach().cpu() - pv_r.detach()).abs().max().item()
    assert e_w.max().item() < FWD_TOL and e_m < FWD_TOL and e_v < FWD_TOL, (e_w.max().item(), e_m, e_v)
    g0, gz = torch.autograd.grad((world * gw.to(device)).sum() + (pm * gm.to(device)).sum() + (pv * gv.to(device)).sum(), [past, z])
    allow = int(np.ceil(0.02 * B))
    eg0 = assert_grad('g_past0', g0.cpu().numpy(), g_ref[0].numpy(), stable, allow=allow)
    egz = assert_grad('g_z', gz.cpu().numpy(), g_ref[1].numpy(), stable, allow=allow)
    e_all = np.maximum(per_seq_rel(g0.cpu().numpy(), g_ref[0].numpy()), per_seq_rel(gz.cpu().numpy(), g_ref[1].numpy()))
    return dict(world=float(e_w.max()), prior_mu=e_m, prior_var=e_v, g_past0=eg0, g_z=egz, flagged=int((~stable).sum()),
                unflagged_over_bar=int((e_all[stable] > GRAD_RTOL).sum()), median_grad_err=float(np.median(e_all)), drift64=drift)


def check_rollout_golden(lib, device):
    gd = golden('rollout.npz')
    hm, _ = make_model(lib, device, seed=int(gd['weight_seed']))
    past = torch.tensor(gd['past0']).to(device).requires_grad_(True)
    z = torch.tensor(gd['z']).to(device).requires_grad_(True)
    out, (pm, pv) = hm.roll_out(past, None, z.shape[1], z_seq=z, return_prior=True)
    world = world_of(out)
    assert np.abs(world.detach().cpu().numpy() - gd['world']).max() < FWD_TOL
    assert np.abs(pm.detach().cpu().numpy() - gd['prior_mu']).max() < FWD_TOL
    assert np.abs(pv.detach().cpu().numpy() - gd['prior_var']).max() < FWD_TOL
    t = lambda k: torch.tensor(gd[k]).to(device)
    loss = (world * t('gw')).sum() + (pm * t('gm')).sum() + (pv * t('gv')).sum()
    g0, gz = torch.autograd.grad(loss, [past, z])
    assert np.abs(g0.cpu().numpy() - gd['g_past0']).max() < 2e-3 * max(1.0, np.abs(gd['g_past0']).max())
    assert np.abs(gz.cpu().numpy() - gd['g_z']).max() < 2e-3 * max(1.0, np.abs(gd['g_z']).max())
    aa_root = ops.rotation_matrix_to_angle_axis(out['root_orient'].reshape(-1, 3, 3), _lib_override=lib)
    assert np.abs(aa_root.detach().cpu().numpy().reshape(gd['aa_root'].shape) - gd['aa_root']).max() < FWD_TOL


def rotrep_model(lib, rep, seed):
    """The fixture's model for `rep`: out_rot_rep '6d' / '9d', or 'nd' = axis-angle outputs with output_delta=False."""
    delta = rep != 'nd'
    sd = synth.rotrep_state_dict(rep, seed=seed) if delta else synth.nodelta_state_dict(seed=seed)
    hm = HumorModel(in_rot_rep='mat', out_rot_rep=rep if delta else 'aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=1,
                    output_delta=delta, _lib_override=lib)
    hm.load_state_dict(sd)
    return hm, sd, delta


def check_rollout_rotrep_short(lib, device, rep, B=2, S=2):
    """The first S steps of B sequences of the rotation-representation fixture (emulator-sized): forward against the reference's
    outputs, gradients against the restated oracle's autograd on the same inputs."""
    gd = golden('rollout_rotrep.npz')
    p = 'r' + rep + '_'
    hm, sd, delta = rotrep_model(lib, rep, int(gd['weight_seed']))
    hm = hm.to(device).eval()
    pc = torch.tensor(gd[p + 'past0'][:B]).requires_grad_(True)
    zc = torch.tensor(gd[p + 'z'][:B, :S]).requires_grad_(True)
    past, z = pc.detach().clone().to(device).requires_grad_(True), zc.detach().clone().to(device).requires_grad_(True)
    out, (pm, pv) = hm.roll_out(past, None, S, z_seq=z, return_prior=True)
    world = world_of(out)
    assert np.abs(world.detach().cpu().numpy() - gd[p + 'world'][:B, :S]).max() < FWD_TOL
    assert np.abs(pm.detach().cpu().numpy() - gd[p + 'prior_mu'][:B, :S]).max() < FWD_TOL
    t = lambda k: torch.tensor(gd[p + k][:B, :S])
    g0, gz = torch.autograd.grad((world * t('gw').to(device)).sum() + (pm * t('gm').to(device)).sum() + (pv * t('gv').to(device)).sum(), [past, z])
    wo, (pmo, pvo) = H.roll_out(sd, pc, zc, output_delta=delta)
    r0, rz = torch.autograd.grad((wo * t('gw')).sum() + (pmo * t('gm')).sum() + (pvo * t('gv')).sum(), [pc, zc])
    e0 = (g0.cpu() - r0).abs().max().item() / max(1.0, r0.abs().max().item())
    ez = (gz.cpu() - rz).abs().max().item() / max(1.0, rz.abs().max().item())
    assert e0 < GRAD_RTOL and ez < GRAD_RTOL, (e0, ez)
    return e0, ez


def check_rollout_rotrep_golden(lib, device, rep):
    """HumorModel(out_rot_rep='6d' | '9d') roll-out (the launch-chain kernels with the 6-D / SVD residual-rotation glue) against the
    reference-generated fixture (oracle/make_golden_rotrep.py), and the model's plain-PyTorch single step (decode) against the
    oracle's on the first step."""
    gd = golden('rollout_rotrep.npz')
    p = 'r' + rep + '_'
    hm, sd, delta = rotrep_model(lib, rep, int(gd['weight_seed']))
    hm = hm.to(device).eval()
    past = torch.tensor(gd[p + 'past0']).to(device).requires_grad_(True)
    z = torch.tensor(gd[p + 'z']).to(device).requires_grad_(True)
    out, (pm, pv) = hm.roll_out(past, None, z.shape[1], z_seq=z, return_prior=True)
    world = world_of(out)
    ew = np.abs(world.detach().cpu().numpy() - gd[p + 'world']).max()
    assert ew < FWD_TOL, ew
    assert np.abs(pm.detach().cpu().numpy() - gd[p + 'prior_mu']).max() < FWD_TOL
    assert np.abs(pv.detach().cpu().numpy() - gd[p + 'prior_var']).max() < FWD_TOL
    t = lambda k: torch.tensor(gd[p + k]).to(device)
    loss = (world * t('gw')).sum() + (pm * t('gm')).sum() + (pv * t('gv')).sum()
    g0, gz = torch.autograd.grad(loss, [past, z])
    e0 = np.abs(g0.cpu().numpy() - gd[p + 'g_past0']).max() / max(1.0, np.abs(gd[p + 'g_past0']).max())
    ez = np.abs(gz.cpu().numpy() - gd[p + 'g_z']).max() / max(1.0, np.abs(gd[p + 'g_z']).max())
    assert e0 < GRAD_RTOL and ez < GRAD_RTOL, (e0, ez)
    # single step in plain PyTorch (HumorModel.decode) against the oracle's residual composition
    with torch.no_grad():
        pc, zc = torch.tensor(gd[p + 'past0']), torch.tensor(gd[p + 'z'])[:, 0]
        hc = rotrep_model(lib, rep, int(gd['weight_seed']))[0].eval()
        dec_lin, dec_gn = H.mlp_params(sd, 'decoder')
        raw = H.mlp_forward(torch.cat([pc, zc], 1), dec_lin, dec_gn, skip=zc)
        got = hc.split_output(hc.decode(zc, pc).reshape(pc.shape[0], 1, -1))
        got = torch.cat([got[k] for k in KEYS], 2).reshape(pc.shape[0], -1)
        assert (got - H.decode_compose(pc, raw, output_delta=delta)).abs().max().item() < 1e-5
    return ew, e0, ez


INREP_CASES = [('aa', 1), ('6d', 1), ('mat', 2), ('aa', 2)]


def check_rollout_inrep(lib, device, rep, steps_in):
    """HumorModel's input variants (in_rot_rep 'aa' / '6d', steps_in = 2) against the reference-generated fixture tests/golden/rollout_inrep.npz
    (oracle/make_golden_inrep.py: world states, prior outputs, gradients w.r.t. the latent sequence and the initial window), driven like the
    reference (x_past in the input representation + init_input_dict with matrix rotations).  Flat bars 1e-4 / 1e-3."""
    from humor_amd import synth as SY
    gold = golden('rollout_inrep.npz')
    tag = f'{rep}_{steps_in}'
    rot_w = {'aa': 3, '6d': 6, 'mat': 9}[rep]
    in_dim = 3 + 3 + rot_w + 3 + 21 * rot_w + 66 + 66
    sd = SY.humor_state_dict(seed=0, in_dim=in_dim, past_steps=steps_in, **SY.CONTRACTIVE)
    hm = HumorModel(in_rot_rep=rep, out_rot_rep='aa', latent_size=48, model_data_config='smpl+joints+contacts', steps_in=steps_in, _lib_override=lib)
    hm.load_state_dict(sd)
    hm = hm.to(device).eval()
    # on the device the frozen prior goes through the fused MLP kernels (ha_mlp_*); on host tensors the parameters stay trainable, which keeps
    # the prior on its PyTorch module (the emulator build of the GEMM kernels would take minutes per case here)
    for p in hm.parameters():
        p.requires_grad_(device.type != 'cuda')
    t = lambda k: torch.from_numpy(gold[f'{tag}_{k}']).to(device)
    win0, z = t('win0').requires_grad_(True), t('z').requires_grad_(True)
    B, S = z.shape[0], z.shape[1]
    names, dims = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel'], [3, 3, 9, 3, 189, 66, 66]
    d, o = {}, 0
    for k, n in zip(names, dims):
        d[k] = win0[:, :, o:o + n]
        o += n
    parts = []
    for k in names:
        v = d[k]
        if k in ('root_orient', 'pose_body') and rep != 'mat':
            nj = v.shape[2] // 9
            if rep == 'aa':
                v = ops.rotation_matrix_to_angle_axis(v.reshape(B * steps_in * nj, 3, 3), _lib_override=lib).reshape(B, steps_in, nj * 3)
            else:
                v = v.reshape(B, steps_in, nj, 9)[:, :, :, :6].reshape(B, steps_in, nj * 6)
        parts.append(v)
    x_past = torch.cat(parts, dim=2)
    pred, (pm, pv) = hm.roll_out(x_past, d, S, z_seq=z, return_prior=True)
    world = torch.cat([pred[k] for k in names + ['contacts']], 2)
    e = {'world': (world - t('world')).abs().max().item(), 'pm': (pm - t('pm')).abs().max().item(),
         'pv': ((pv - t('pv')).abs() / t('pv').abs().clamp(min=1.0)).max().item()}
    assert max(e.values()) < FWD_TOL, e
    g = torch.autograd.grad((world * t('gw')).sum() + (pm * t('gm')).sum() + (pv * t('gv')).sum(), [win0, z])
    for name, got, ref in (('g_win', g[0], t('g_win')), ('g_z', g[1], t('g_z'))):
        e[name] = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert e[name] < GRAD_RTOL, (name, e)
    # the roll-out continues where the reference's cannot (second step with in_rot_rep != 'mat'): finite, and its first step is the one-step result
    if rep != 'mat':
        z2 = torch.cat([z.detach(), 0.5 * z.detach()], dim=1)
        two = hm.roll_out(x_past.detach(), {k: v.detach() for k, v in d.items()}, 2, z_seq=z2)
        assert all(torch.isfinite(v).all() for v in two.values()) and two['trans'].shape[1] == 2
        assert (two['trans'][:, :1].detach() - pred['trans'].detach()).abs().max().item() < 1e-6
    return e


def check_persistent_kernels_whole_team(emu_lib, B, S, seed=0):
    """The B <= 32 persistent roll-out kernels AS A WHOLE on the host emulator (tests/simt_emu: resident teams): the 32 blocks of one XCD team run
    at the same time, one OS thread per work-item, every block on its own NaN-filled LDS, every buffer the kernels write NaN-filled.  Forward:
    team formation, the step loop, the four exchange hand-offs per step through tagged granules, consumer-side GroupNorm, the glue chains, copy_out;
    then the one-launch adjoint over the forward's stash (reverse scan, GroupNorm / glue adjoints, dL/dz partials + reduction).  Against the
    oracle at the flat bars; the state slabs the prior reads and the decoder outputs must be finite for every live row -- the pad channel of the
    ODD steps is where round 5's uninitialised LDS buffer showed (removing that fix makes this check fail: 3 non-finite entries per odd step)."""
    import ctypes as C
    dll = emu_lib._dll
    sd = synth.humor_state_dict(seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    past = canonical_state(B, g).requires_grad_(True)
    z = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    gw = torch.randn(B, S, 348, generator=g)
    f = lambda t: np.ascontiguousarray(t.detach().float().numpy())
    arrs = [f(sd[f'decoder.net.{i}.weight']) for i in (0, 3, 6, 9)] + [f(sd[f'decoder.net.{i}.bias']) for i in (0, 3, 6, 9)]
    for i in (1, 4, 7):
        arrs += [f(sd[f'decoder.net.{i}.weight']), f(sd[f'decoder.net.{i}.bias'])]
    pin, zin, gwn = f(past), f(z), f(gw)
    world, xT, raw = np.zeros((B, S, 348), np.float32), np.zeros((S + 1, 85, 32, 4), np.float32), np.zeros((S, 56, 32, 4), np.float32)
    g_past, g_z = np.zeros((B, 339), np.float32), np.zeros((B, S, 48), np.float32)
    err, errb = C.c_uint(0), C.c_uint(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    fn = dll.ha_emu_persist_team
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 19 + [C.POINTER(C.c_uint), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
    assert fn(B, S, *[P(a) for a in arrs], P(pin), P(zin), P(world), P(xT), P(raw), C.byref(err), P(gwn), P(g_past), P(g_z), C.byref(errb)) == 0
    assert err.value == 0 and errb.value == 0, (hex(err.value), hex(errb.value))
    w_ref, _ = H.roll_out(sd, past, z)
    (w_ref * gw).sum().backward()
    e = {'world': float(np.abs(world - w_ref.detach().numpy()).max())}
    assert np.isfinite(world).all() and e['world'] < FWD_TOL, e
    live = xT[:, :, :B, :]
    bad = [int((~np.isfinite(live[t])).sum()) for t in range(S + 1)]
    assert not any(bad), ('non-finite entries in the state slabs of the live rows, by step', bad)
    assert np.isfinite(raw[:, :54, :B, :]).all()
    for name, got, ref in (('g_past', g_past, past.grad.numpy()), ('g_z', g_z, z.grad.numpy())):
        assert np.isfinite(got).all(), name
        e[name] = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
        assert e[name] < GRAD_RTOL, e
    return e


def check_persistent_failure_protocol_whole_team(emu_lib, B=6, S=1, seed=0):
    """The failure protocol of the persistent forward on the host emulator: with the injection of ha_tune_set("rollout_persist_inject") member 3 of
    team 0 leaves at once; the team's bounded waits run out (the bound lowered to 30 s of polls), every output row of team 0 is NaN, the error word is
    0x200 | team, and the OTHER resident team (rows 4 ..) finishes with correct results -- teams never talk to one another."""
    import ctypes as C
    dll = emu_lib._dll
    sd = synth.humor_state_dict(seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    past, z = canonical_state(B, g), torch.randn(B, S, 48, generator=g)
    f = lambda t: np.ascontiguousarray(t.detach().float().numpy())
    arrs = [f(sd[f'decoder.net.{i}.weight']) for i in (0, 3, 6, 9)] + [f(sd[f'decoder.net.{i}.bias']) for i in (0, 3, 6, 9)]
    for i in (1, 4, 7):
        arrs += [f(sd[f'decoder.net.{i}.weight']), f(sd[f'decoder.net.{i}.bias'])]
    pin, zin = f(past), f(z)
    world, xT, raw = np.zeros((B, S, 348), np.float32), np.zeros((S + 1, 85, 32, 4), np.float32), np.zeros((S, 56, 32, 4), np.float32)
    err, errb = C.c_uint(0), C.c_uint(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    fn = dll.ha_emu_persist_team
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 19 + [C.POINTER(C.c_uint), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
    dll.ha_emu_persist_knobs(1, 1500)
    try:
        assert fn(B, S, *[P(a) for a in arrs], P(pin), P(zin), P(world), P(xT), P(raw), C.byref(err), None, None, None, C.byref(errb)) == 0
    finally:
        dll.ha_emu_persist_knobs(0, 0)
    assert err.value == 0x200, hex(err.value)
    assert np.isnan(world[:4]).all(), 'the incomplete team\'s rows must be NaN'
    w_ref, _ = H.roll_out(sd, past, z)
    e = float(np.abs(world[4:] - w_ref.numpy()[4:]).max())
    assert np.isfinite(world[4:]).all() and e < FWD_TOL, e
    return e


def check_pipelined_failure_protocol_whole_team(emu_lib, B=33, S=1, seed=0):
    """The same protocol in the pipelined forward (one resident team): a layer-0 CU that leaves at once -> every row the team owns, in EVERY tile, is NaN
    and the error word is 0x600 | team."""
    import ctypes as C
    dll = emu_lib._dll
    NG = (B + 31) // 32
    rows = [32 * g + i for g in range(NG) for i in range(4) if 32 * g + i < B]
    sd = synth.humor_state_dict(seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    past, z = canonical_state(B, g), torch.randn(B, S, 48, generator=g)
    f = lambda t: np.ascontiguousarray(t.detach().float().numpy())
    arrs = [f(sd[f'decoder.net.{i}.weight']) for i in (0, 3, 6, 9)] + [f(sd[f'decoder.net.{i}.bias']) for i in (0, 3, 6, 9)]
    for i in (1, 4, 7):
        arrs += [f(sd[f'decoder.net.{i}.weight']), f(sd[f'decoder.net.{i}.bias'])]
    pin, zin = f(past), f(z)
    world, xT = np.zeros((B, S, 348), np.float32), np.zeros((S + 1, NG, 85, 32, 4), np.float32)
    err, errb = C.c_uint(0), C.c_uint(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    fn = dll.ha_emu_pipe_team
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 18 + [C.POINTER(C.c_uint), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
    dll.ha_emu_persist_knobs(1, 1500)
    try:
        assert fn(B, S, *[P(a) for a in arrs], P(pin), P(zin), P(world), P(xT), C.byref(err), None, None, None, C.byref(errb)) == 0
    finally:
        dll.ha_emu_persist_knobs(0, 0)
    assert err.value == 0x600, hex(err.value)
    assert np.isnan(world[rows]).all(), 'the incomplete team\'s rows must be NaN in every tile'
    return rows


def check_pipelined_kernels_whole_team(emu_lib, B, S, seed=0):
    """The pipelined roll-out kernels (rollout_pipe.inc, 32 < B <= 256) as a whole on the host emulator, for ONE resident team: layer roles on
    5 / 16 / 8 / 2 CUs + the glue CU, NG = ceil(B / 32) groups of four sequences flowing through them, forward and one-launch adjoint, on NaN-filled
    LDS and buffers.  Team 0 owns sequences 32 g + 0 .. 3 of every tile g: those rows against the oracle at the flat bars (the other teams are not
    resident; their rows stay NaN)."""
    import ctypes as C
    dll = emu_lib._dll
    NG = (B + 31) // 32
    rows = [32 * g + i for g in range(NG) for i in range(4) if 32 * g + i < B]
    sd = synth.humor_state_dict(seed=seed)
    g = torch.Generator().manual_seed(seed + 5)
    past = canonical_state(B, g).requires_grad_(True)
    z = torch.randn(B, S, 48, generator=g).requires_grad_(True)
    gw = torch.randn(B, S, 348, generator=g)
    f = lambda t: np.ascontiguousarray(t.detach().float().numpy())
    arrs = [f(sd[f'decoder.net.{i}.weight']) for i in (0, 3, 6, 9)] + [f(sd[f'decoder.net.{i}.bias']) for i in (0, 3, 6, 9)]
    for i in (1, 4, 7):
        arrs += [f(sd[f'decoder.net.{i}.weight']), f(sd[f'decoder.net.{i}.bias'])]
    pin, zin, gwn = f(past), f(z), f(gw)
    world, xT = np.zeros((B, S, 348), np.float32), np.zeros((S + 1, NG, 85, 32, 4), np.float32)
    g_past, g_z = np.zeros((B, 339), np.float32), np.zeros((B, S, 48), np.float32)
    err, errb = C.c_uint(0), C.c_uint(0)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    fn = dll.ha_emu_pipe_team
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 18 + [C.POINTER(C.c_uint), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint)]
    assert fn(B, S, *[P(a) for a in arrs], P(pin), P(zin), P(world), P(xT), C.byref(err), P(gwn), P(g_past), P(g_z), C.byref(errb)) == 0
    assert err.value == 0 and errb.value == 0, (hex(err.value), hex(errb.value))
    w_ref, _ = H.roll_out(sd, past, z)
    (w_ref * gw).sum().backward()
    e = {'world': float(np.abs(world[rows] - w_ref.detach().numpy()[rows]).max())}
    assert np.isfinite(world[rows]).all() and e['world'] < FWD_TOL, e
    nl = [min(4, B - 32 * gg) for gg in range(NG)]
    bad = [int(sum((~np.isfinite(xT[t, gg, :, :nl[gg], :])).sum() for gg in range(NG))) for t in range(S + 1)]
    assert not any(bad), ('non-finite entries in the state slabs of the live rows, by step', bad)
    for name, got, ref in (('g_past', g_past, past.grad.numpy()), ('g_z', g_z, z.grad.numpy())):
        assert np.isfinite(got[rows]).all(), name
        e[name] = float(np.abs(got[rows] - ref[rows]).max() / max(1.0, np.abs(ref).max()))
        assert e[name] < GRAD_RTOL, e
    return e


def check_rotations_golden(lib, device):
    gd = golden('rotations.npz')
    aa = torch.tensor(gd['aa']).to(device).requires_grad_(True)
    Rm = ops.batch_rodrigues(aa, _lib_override=lib)
    assert np.abs(Rm.detach().cpu().numpy() - gd['R']).max() < 1e-6
    g_aa = torch.autograd.grad((Rm * torch.tensor(gd['gR']).to(device)).sum(), aa)[0].cpu().numpy()
    ok = np.isfinite(gd['g_aa']).all(axis=1) & (np.abs(gd['aa']).sum(axis=1) > 0)   # skip the exact-zero rows (1/1.7e-8 scaling)
    assert np.abs(g_aa[ok] - gd['g_aa'][ok]).max() < 1e-3 * max(1.0, np.abs(gd['g_aa'][ok]).max())
    Rin = torch.tensor(gd['R']).to(device).requires_grad_(True)
    back = ops.rotation_matrix_to_angle_axis(Rin, _lib_override=lib)
    assert np.abs(back.detach().cpu().numpy() - gd['aa_back']).max() < 1e-5
    g_R = torch.autograd.grad((back * torch.tensor(gd['gb']).to(device)).sum(), Rin)[0].cpu().numpy()
    # the reference gradient is ill-conditioned at theta ~ 0 and theta ~ pi (G3): compare well-conditioned rows
    ang = np.linalg.norm(gd['aa_back'], axis=1)
    good = (ang > 0.2) & (ang < 2.6) & np.isfinite(gd['g_Rin']).all(axis=(1, 2))
    assert np.abs(g_R[good] - gd['g_Rin'][good]).max() < 1e-3 * max(1.0, np.abs(gd['g_Rin'][good]).max())


def check_rot_random(lib, device, n=4096, seed=0):
    """rotation kernels against the oracle on random, well-conditioned inputs (fwd + bwd)."""
    g = torch.Generator().manual_seed(seed)
    aa_c = (1.0 * torch.randn(n, 3, generator=g)).requires_grad_(True)
    aa = aa_c.detach().to(device).requires_grad_(True)
    R_ref = L.batch_rodrigues(aa_c)
    Rm = ops.batch_rodrigues(aa, _lib_override=lib)
    assert (Rm.detach().cpu() - R_ref).abs().max().item() < 1e-6
    gR = torch.randn(n, 3, 3, generator=g)
    ga = torch.autograd.grad((Rm * gR.to(device)).sum(), aa)[0].cpu()
    ga_ref = torch.autograd.grad((R_ref * gR).sum(), aa_c)[0]
    assert (ga - ga_ref).abs().max().item() < 1e-4 * max(1.0, ga_ref.abs().max().item())
    Rc = R_ref.detach().clone().requires_grad_(True)
    Rd = Rc.detach().to(device).requires_grad_(True)
    b_ref = H.rot_to_aa(Rc)
    b = ops.rotation_matrix_to_angle_axis(Rd, _lib_override=lib)
    assert (b.detach().cpu() - b_ref).abs().max().item() < 1e-5
    gb = torch.randn(n, 3, generator=g)
    gr = torch.autograd.grad((b * gb.to(device)).sum(), Rd)[0].cpu()
    gr_ref = torch.autograd.grad((b_ref * gb).sum(), Rc)[0]
    ang = b_ref.detach().norm(dim=1)
    good = (ang > 0.2) & (ang < 2.6)
    assert (gr[good] - gr_ref[good]).abs().max().item() < 1e-3 * max(1.0, gr_ref[good].abs().max().item())


def check_sampling_rollout(lib, device, B=2, S=30, seed=0, n_mean=6, n_canon=5):
    """BASELINE config C1 shape (test_humor_sampling_debug: batch 2, 30 steps, z sampled from the prior each step) and the
    canonicalize_input / uncanonicalize_output options of HumorModel.roll_out, against the oracle with the same noise."""
    hm, sd = make_model(lib, device, seed=seed)
    g = torch.Generator().manual_seed(seed + 11)
    past = canonical_state(B, g)
    eps = torch.randn(B, S, 48, generator=g)
    out, (pm, pv) = hm.roll_out(past.to(device), None, S, z_seq=None, return_prior=True, return_z=True, eps_seq=eps.to(device))
    w_ref, (pm_r, pv_r) = H.roll_out(sd, past, None, eps_seq=eps)
    n = min(S, 12)          # compare where the chain is still well conditioned
    assert (world_of(out)[:, :n].cpu() - w_ref[:, :n]).abs().max().item() < FWD_TOL
    assert (pm[:, :n].cpu() - pm_r[:, :n]).abs().max().item() < FWD_TOL
    z_ref = pm_r + eps * torch.sqrt(pv_r)
    assert (out['z'][:, :n].cpu() - z_ref[:, :n]).abs().max().item() < FWD_TOL
    assert torch.isfinite(world_of(out)).all()
    # use_mean
    out_m = hm.roll_out(past.to(device), None, n_mean, z_seq=None, use_mean=True)
    w_m, _ = H.roll_out(sd, past, None, eps_seq=torch.zeros(B, n_mean, 48))
    assert (world_of(out_m).cpu() - w_m).abs().max().item() < FWD_TOL
    # canonicalize_input (+ uncanonicalize_output): a state rotated/translated in the world
    from humor_amd import frames
    ang = torch.tensor([0.7, -1.2])[:B]
    Rz = L.batch_rodrigues(torch.stack([torch.zeros_like(ang), torch.zeros_like(ang), ang], 1))
    shift = torch.tensor([[1.5, -0.7, 0.0], [-2.0, 0.3, 0.0]])[:B]
    rot = lambda v: torch.einsum('bij,bj->bi', Rz, v)
    j = torch.einsum('bij,bkj->bki', Rz, past[:, 207:273].reshape(B, 22, 3)) + shift.unsqueeze(1)
    jv = torch.einsum('bij,bkj->bki', Rz, past[:, 273:339].reshape(B, 22, 3))
    moved = torch.cat([rot(past[:, 0:3]) + shift, rot(past[:, 3:6]), torch.matmul(Rz, past[:, 6:15].reshape(B, 3, 3)).reshape(B, 9),
                       rot(past[:, 15:18]), past[:, 18:207], j.reshape(B, 66), jv.reshape(B, 66)], 1)
    z = torch.randn(B, n_canon, 48, generator=g)
    keys = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel']
    dims = [3, 3, 9, 3, 189, 66, 66]
    d, o = {}, 0
    for k, nn in zip(keys, dims):
        d[k] = moved[:, o:o + nn].unsqueeze(1).to(device)
        o += nn
    out_c = hm.roll_out(None, d, n_canon, z_seq=z.to(device), canonicalize_input=True, uncanonicalize_output=True)
    local, (R0, t0, t2j) = frames.canonicalize_state(moved)
    w_c, _ = H.roll_out(sd, local, z, G0=R0, gt0=t0, t2j=t2j)
    assert (world_of(out_c).cpu() - w_c).abs().max().item() < 2e-4
    out_cc = hm.roll_out(None, d, n_canon, z_seq=z.to(device), canonicalize_input=True)
    w_cc, _ = H.roll_out(sd, local, z, t2j=t2j)
    assert (world_of(out_cc).cpu() - w_cc).abs().max().item() < 2e-4


def check_sampling_golden(lib, device):
    """Reference-generated vectors (oracle/make_golden_sampling.py) for BASELINE config C1's sampling roll-out, the prior-mean
    roll-out and the two canonicalize_input forms.  Sampling compounds rounding through the drawn latents, so the 30-step
    chain is compared over its first 10 steps at the stated 1e-4 and checked finite/bounded beyond."""
    gd = golden('rollout_sampling.npz')
    hm, _ = make_model(lib, device, seed=int(gd['weight_seed']))
    past = torch.from_numpy(gd['past']).to(device)
    eps = torch.from_numpy(gd['eps']).to(device)
    S = eps.shape[1]
    out = hm.roll_out(past, None, S, z_seq=None, return_z=True, eps_seq=eps)
    w = world_of(out).cpu().numpy()
    n = 10
    assert np.abs(w[:, :n] - gd['world_sampled'][:, :n]).max() < FWD_TOL
    assert np.abs(out['z'].cpu().numpy()[:, :n] - gd['z_sampled'][:, :n]).max() < FWD_TOL
    assert np.isfinite(w).all() and np.abs(w - gd['world_sampled']).max() < 0.5
    out_m = hm.roll_out(past, None, gd['world_mean'].shape[1], z_seq=None, use_mean=True)
    assert np.abs(world_of(out_m).cpu().numpy() - gd['world_mean']).max() < FWD_TOL
    moved = torch.from_numpy(gd['moved'])
    keys = ['trans', 'trans_vel', 'root_orient', 'root_orient_vel', 'pose_body', 'joints', 'joints_vel']
    dims = [3, 3, 9, 3, 189, 66, 66]
    d, o = {}, 0
    for k, nn in zip(keys, dims):
        d[k] = moved[:, o:o + nn].unsqueeze(1).to(device)
        o += nn
    z = torch.from_numpy(gd['z_canon']).to(device)
    for unc, key in ((True, 'world_canon_uncanon'), (False, 'world_canon')):
        oc = hm.roll_out(None, d, z.shape[1], z_seq=z, canonicalize_input=True, uncanonicalize_output=unc)
        assert np.abs(world_of(oc).cpu().numpy() - gd[key]).max() < 2e-4


def check_rot6d(lib, device, n=2000, seed=0):
    """rot6d_to_rotmat kernel (fwd + adjoint) against the restated formula of transforms.py:201-220 evaluated by PyTorch autograd."""
    g = torch.Generator().manual_seed(seed)
    xc = torch.randn(n, 6, generator=g)
    xc[0] = torch.tensor([1.0, 0.0, 0.0, 1.0, 0.0, 0.0])
    xc = xc.requires_grad_(True)
    x = xc.detach().clone().to(device).requires_grad_(True)
    v = xc.view(-1, 3, 2)
    a1, a2 = v[:, :, 0], v[:, :, 1]
    b1 = torch.nn.functional.normalize(a1)
    b2 = torch.nn.functional.normalize(a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1)
    R_ref = torch.stack((b1, b2, torch.cross(b1, b2, dim=1)), dim=-1)
    Rm = ops.rot6d_to_rotmat(x, _lib_override=lib)
    # Gram-Schmidt amplifies rounding by 1 / sin(angle(a1, a2)) (and the gradient by its square): rows are judged at their conditioning
    sin = (a2 - torch.einsum('bi,bi->b', b1, a2).unsqueeze(-1) * b1).norm(dim=1).detach() / a2.norm(dim=1).detach()
    cond = (1.0 / sin.clamp(min=1e-6))
    e = (Rm.detach().cpu() - R_ref).abs().amax(dim=(1, 2))
    assert (e <= 1e-6 * cond).all(), (e / cond).max().item()
    gR = torch.randn(n, 3, 3, generator=g)
    ga = torch.autograd.grad((Rm * gR.to(device)).sum(), x)[0].cpu()
    ga_ref = torch.autograd.grad((R_ref * gR).sum(), xc)[0]
    well = sin > 0.1
    eg = (ga - ga_ref).abs().amax(dim=1) / ga_ref.abs().amax(dim=1).clamp(min=1.0)
    assert eg[well].max().item() < 1e-4, eg[well].max().item()
    assert (eg <= 1e-4 * cond * cond).all()
    return R_ref


def rot9d_reference_formula(x):
    """rot9d_to_rotmat as humor/utils/transforms.py:222-241 writes it (torch.svd; the live test in test_oracle.py pins it bit for bit)."""
    B = x.size(0)
    x = x.reshape((B, 3, 3))
    u, s, v = torch.svd(x)
    v_T = v.transpose(-2, -1)
    s_p = torch.eye(3).to(x).reshape((1, 3, 3)).expand_as(x).clone()
    s_p[:, 2, 2] = torch.det(torch.matmul(u, v_T))
    return torch.matmul(torch.matmul(u, s_p), v_T).reshape((B, 9))


def check_rot9d(lib, device, n=2000, seed=0):
    """rot9d_to_rotmat kernel (Jacobi SVD projection + its adjoint) against the reference formula evaluated by PyTorch (torch.svd and
    its autograd) in float64: perturbed rotations (what a residual decoder emits), generic matrices of both determinant signs,
    the identity.  torch.svd's autograd divides by s_k^2 - s_l^2, so gradients are compared on rows with separated singular values."""
    g = torch.Generator().manual_seed(seed)
    n3 = n // 3
    Rr = L.batch_rodrigues(1.5 * torch.randn(n3, 3, generator=g)).reshape(n3, 9)
    xc = torch.cat([Rr + 0.2 * torch.randn(n3, 9, generator=g), torch.randn(n - n3, 9, generator=g)], 0)
    xc[0] = torch.eye(3).reshape(9)
    xd = xc.double().requires_grad_(True)
    x = xc.clone().to(device).requires_grad_(True)
    R_ref = rot9d_reference_formula(xd)
    Rm = ops.rot9d_to_rotmat(x, _lib_override=lib).reshape(n, 9)
    sv = torch.linalg.svdvals(xc.double().reshape(n, 3, 3))
    det = torch.det(xc.double().reshape(n, 3, 3))
    # conditioning of the projection: 1 / (s_2 + s_3) for det > 0, 1 / (s_2 - s_3) for det < 0
    gap = torch.where(det > 0, sv[:, 1] + sv[:, 2], sv[:, 1] - sv[:, 2]) / sv[:, 0]
    e = (Rm.detach().cpu().double() - R_ref.detach()).abs().amax(dim=1)
    assert (e <= 2e-6 / gap.clamp(min=1e-6)).all(), (e * gap).max().item()
    ortho = (torch.matmul(Rm.detach().cpu().reshape(n, 3, 3), Rm.detach().cpu().reshape(n, 3, 3).transpose(1, 2)) - torch.eye(3)).abs().max().item()
    assert ortho < 1e-5, ortho
    assert (torch.det(Rm.detach().cpu().reshape(n, 3, 3)) > 0.999).all()
    gR = torch.randn(n, 9, generator=g)
    ga = torch.autograd.grad((Rm * gR.to(device)).sum(), x)[0].cpu().double()
    ga_ref = torch.autograd.grad((R_ref * gR.double()).sum(), xd)[0]
    sep = torch.minimum(sv[:, 0] - sv[:, 1], sv[:, 1] - sv[:, 2]) / sv[:, 0]
    well = (sep > 0.05) & (gap > 0.05)
    assert well.sum().item() > n // 4
    eg = (ga - ga_ref).abs().amax(dim=1) / ga_ref.abs().amax(dim=1).clamp(min=1.0)
    assert eg[well].max().item() < 1e-4, eg[well].max().item()
    # rows with (nearly) equal singular values: finite and equal to a central finite difference of the projection itself
    idx = [0] + torch.nonzero(~well).flatten()[:8].tolist()
    eps = 1e-4
    for i in idx:
        if gap[i] < 0.05:
            continue
        fd = torch.zeros(9, dtype=torch.float64)
        for k in range(9):
            xp, xm = xc[i].double().clone(), xc[i].double().clone()
            xp[k] += eps; xm[k] -= eps
            fd[k] = ((rot9d_reference_formula(xp[None]) - rot9d_reference_formula(xm[None]))[0] * gR[i].double()).sum() / (2 * eps)
        assert (ga[i] - fd).abs().max().item() < 2e-3 * max(1.0, fd.abs().max().item()), (i, ga[i], fd)
    return eg[well].max().item()


def check_rot_to_aa_near_pi(lib, device, n=3000, seed=0):
    """R -> axis-angle forward AND gradient right at the seam the optimiser starts from (root_orient = (pi, 0, 0)):
    rotation angles in [2.6, pi - 1e-4] plus exact pi-rotations about the coordinate axes, against the oracle (bit-pinned to
    transforms.py:243-389).  The 4-branch quaternion extraction is well conditioned there: flat 1e-5 / 1e-4-relative bars."""
    g = torch.Generator().manual_seed(seed)
    axis = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    th = torch.cat([2.6 + (3.1415 - 2.6) * torch.rand(n - 600, 1, generator=g), 3.13 + 0.0115 * torch.rand(600, 1, generator=g)], 0)
    aa = torch.cat([axis * th, torch.tensor([[3.14159265, 0.0, 0.0], [0.0, 3.14159265, 0.0], [0.0, 0.0, 3.14159265], [3.1415, 1e-3, -1e-3]])], 0)
    R = L.batch_rodrigues(aa)
    Rc = R.clone().requires_grad_(True)
    Rd = R.clone().to(device).requires_grad_(True)
    b_ref = H.rot_to_aa(Rc)
    b = ops.rotation_matrix_to_angle_axis(Rd, _lib_override=lib)
    assert (b.detach().cpu() - b_ref).abs().max().item() < 1e-5
    gb = torch.randn(b_ref.shape, generator=g)
    gr = torch.autograd.grad((b * gb.to(device)).sum(), Rd)[0].cpu()
    gr_ref = torch.autograd.grad((b_ref * gb).sum(), Rc)[0]
    ok = torch.isfinite(gr_ref).all(dim=(1, 2))
    sc = gr_ref[ok].abs().amax(dim=(1, 2)).clamp(min=1.0)
    e = (gr[ok] - gr_ref[ok]).abs().amax(dim=(1, 2)) / sc
    assert e.max().item() < 1e-4, e.max().item()
    return e.max().item()


def persist_status(lib, hm, device):
    """(available, error word, launches) of the persistent forward of `hm`'s network handle on `device` (synchronises first)."""
    import ctypes as C
    if device.type == 'cuda':
        torch.cuda.synchronize(device)
    h = hm._net_handle(device)
    av, err, n = C.c_int(), C.c_uint(), C.c_int64()
    lib.call('ha_humor_persist_status', h.ptr, C.byref(av), C.byref(err), C.byref(n))
    return av.value, err.value, n.value


def check_persistent_vs_chain(lib, device, B, S, seed=0, variant=1, contractive=True, launches_per_call=1):
    """The one-launch persistent forward and adjoint (ha_tune_set "rollout_persist" / "rollout_persist_bwd") against the launch chain on
    the same inputs: world states, prior outputs and every gradient, for (i) persistent forward + launch-chain adjoint reading its
    stash and (ii) persistent forward + persistent adjoint."""
    hm, _ = make_model(lib, device, seed=seed, contractive=contractive)
    g = torch.Generator().manual_seed(77 + B + S)
    past, z = canonical_state(B, g).to(device), torch.randn(B, S, 48, generator=g).to(device)
    gw = torch.randn(B, S, 348, generator=g).to(device)
    gm, gv = torch.randn(B, S, 48, generator=g).to(device), torch.randn(B, S, 48, generator=g).to(device)
    res = []
    n0 = persist_status(lib, hm, device)[2]

    def run(pz, zz):
        p, zq = pz.clone().requires_grad_(True), zz.clone().requires_grad_(True)
        out, (pm, pv) = hm.roll_out(p, None, S, z_seq=zq, return_prior=True)
        w = world_of(out)
        ((w * gw).sum() + (pm * gm).sum() + (pv * gv).sum()).backward()
        return w.detach(), pm.detach(), pv.detach(), p.grad.clone(), zq.grad.clone()
    try:
        for knob, bwd in ((0, 0), (variant, 0), (variant, 1)):
            lib.call('ha_tune_set', b'rollout_persist', knob)
            lib.call('ha_tune_set', b'rollout_persist_bwd', bwd)
            res.append(run(past, z))
        # (gradient bar 3e-4: typical sequences agree to 1e-6; the worst-conditioned unflagged one -- heading within 1e-2 rad of the acos
        # singularity -- is 0.8e-4 .. 1.4e-4 from the float64 gradient on EVERY path, tools/persist_grad_accuracy.py, profiles/r04_persist)
        # Kink flags from the REFERENCE side of this comparison (the launch chain): a sequence whose launch-chain gradient itself moves
        # under 1-ulp changes of the latents has a ReLU unit within rounding of its kink; any other correct fp32 evaluation may land on
        # the other side.  Only those sequences get the kink bar, every other one the flat tolerance (no count allowance).
        lib.call('ha_tune_set', b'rollout_persist', 0)
        lib.call('ha_tune_set', b'rollout_persist_bwd', 0)
        flagged = torch.zeros(B, dtype=torch.bool)
        gp = torch.Generator().manual_seed(5)
        for _ in range(32):          # (the set found grows with the number of perturbations tried: DESIGN section 6)
            sign = (torch.rand(z.shape, generator=gp) > 0.5).float().to(device) * 2 - 1
            sgp = (torch.rand(past.shape, generator=gp) > 0.5).float().to(device) * 2 - 1
            r = run(past * (1.0 + sgp * 2.0 ** -23), z * (1.0 + sign * 2.0 ** -23))
            for a, b in zip(res[0][3:], r[3:]):
                mv = (a - b).abs().reshape(B, -1).amax(dim=1) / max(1.0, a.abs().max().item())
                flagged |= (mv > 1e-4).cpu()
    finally:
        lib.call('ha_tune_set', b'rollout_persist', 1)
        lib.call('ha_tune_set', b'rollout_persist_bwd', 1)
    av, err, n1 = persist_status(lib, hm, device)
    assert err == 0, hex(err)
    # the persistent kernels really ran (no silent fall-back): two forwards (low word), one adjoint (high word)
    assert av == 1 and (n1 & 0xffffffff) == (n0 & 0xffffffff) + 2 * launches_per_call and (n1 >> 32) == (n0 >> 32) + launches_per_call, (av, n0, n1)
    errs = []
    for k in (1, 2):
        for name, a, b, tol in zip(('world', 'prior_mu', 'prior_var', 'g_past', 'g_z'), res[0], res[k], (2e-5, 2e-5, 2e-5, 3e-4, 3e-4)):
            assert torch.isfinite(b).all(), (name, k)
            e = (a - b).abs().reshape(B, -1).amax(dim=1) / max(1.0, a.abs().max().item())       # per sequence
            errs.append(e.max().item())
            if name.startswith('g_'):
                e = e.cpu()
                over = [(i, float(f'{x:.2g}'), bool(flagged[i])) for i, x in enumerate(e.tolist()) if x > tol]
                print(f'persistent (path {k}) vs launch chain, {name}: sequences over {tol:g} (index, error, kink-flagged):', over, '| flagged:', int(flagged.sum()))
                assert e[~flagged].numel() == 0 or e[~flagged].max().item() <= tol, (name, k, over, B, S)
                assert e.max().item() <= KINK_RTOL, (name, k, over, B, S)
            else:
                assert e.max().item() <= tol, (name, k, e.max().item(), B, S)
    return errs
