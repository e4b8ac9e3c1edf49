"""Reads the persistent kernels' stash regions back (debug build: tools/build_variant.sh pdebug -DHA_PERSIST_DEBUG, HUMOR_AMD_LIB=...) and
compares GroupNorm statistics, glue records and the per-slot dL/dz partials of a one-step roll-out with PyTorch on the decoder module."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from humor_amd import _lib
import rollout_checks as RC

dev = torch.device('cuda:0')
lib = _lib.get_lib()
hm, sd = RC.make_model(lib, dev, seed=0, contractive=True)
B, S = 4, 1
g = torch.Generator().manual_seed(3)
past, z = RC.canonical_state(B, g).to(dev), torch.randn(B, S, 48, generator=g).to(dev)
gw = torch.zeros(B, S, 348, device=dev)
gw[:, :, 339:348] = torch.randn(B, S, 9, generator=g).to(dev)
p, zz = past.clone().requires_grad_(True), z.clone().requires_grad_(True)
o = hm.roll_out(p, None, S, z_seq=zz, return_prior=False)
w = RC.world_of(o)
node = w.grad_fn
seen, stack, fn = set(), [node], None
while stack:
    n = stack.pop()
    if n is None or id(n) in seen:
        continue
    seen.add(id(n))
    if hasattr(n, 'stash'):
        fn = n
        break
    stack.extend(x[0] for x in n.next_functions)
assert fn is not None
stash = fn.stash
(w * gw).sum().backward()
torch.cuda.synchronize()
h = hm._net_handle(dev)
lay = (C.c_int64 * 15)()
f = lib._dll.ha_debug_persist_layout
f.restype = C.c_int
assert f(h.ptr, B, S, lay) == 0
xT, steps, per_step, off_G, d0, d1, d2, d3, gn0, gn1, gn2, off_gl, dz_part, single, pws = list(lay)
print('layout', list(lay))
st = stash.cpu()
# ---- torch decoder on the same input -------------------------------------------------------------------------------------
dec = hm.decoder
x0 = torch.cat([past, z[:, 0]], 1).detach().cpu().requires_grad_(True)
mods = list(dec.net)
hs, a = [], x0
zc = z[:, 0].cpu()
k = 0
for mod in mods:
    if isinstance(mod, torch.nn.Linear):
        if k > 0:
            a = torch.cat([a, zc], 1)
        a = mod.cpu()(a)
        hs.append(a)
        k += 1
    else:
        a = mod.cpu()(a)
raw = a
print('raw vs world contacts', (raw[:, 207:216].detach() - w[:, 0, 339:348].detach().cpu()).abs().max().item())
step0 = st[steps: steps + per_step]
def slab(off, C):        # [C/4][32][4] -> [32][C]
    return step0[off: off + C * 32].reshape(C // 4, 32, 4).permute(1, 0, 2).reshape(32, C)
for l, (off, Cn) in enumerate(((d0, 1024), (d1, 1024), (d2, 512))):
    hk = slab(off, Cn)[:B]
    print('h%d slab err' % l, (hk - hs[l].detach()).abs().max().item())
    ng = 16
    gs = Cn // ng
    stats = step0[(gn0, gn1, gn2)[l]: (gn0, gn1, gn2)[l] + 16 * 32 * 2].reshape(16, 32, 2)[:, :B]
    hh = hs[l].detach().reshape(B, ng, gs)
    mu = hh.mean(2)
    var = ((hh - mu.unsqueeze(2)) ** 2).mean(2)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    print('  GN stats: mean err %.2e rstd err %.2e' % ((stats[:, :, 0].T - mu).abs().max().item(), (stats[:, :, 1].T - rstd).abs().max().item()))
gl = step0[off_gl: off_gl + 32 * 32].reshape(32, 32)[:B]
print('glue record W row0', gl[0, :9].tolist(), 'angle', gl[0, 27].item())
# ---- adjoints by autograd: dL/dh_l and the dz contributions per layer ------------------------------------------------------------
g_raw = torch.zeros(B, 216)
g_raw[:, 207:216] = gw[:, 0, 339:348].cpu()
grads = torch.autograd.grad((raw * g_raw).sum(), hs[:3] + [x0], retain_graph=True)
dh = list(grads[:3])
lin = [m for m in mods if isinstance(m, torch.nn.Linear)]
cin = [339, 1024, 1024, 512]
dz_ref = []
dlist = dh + [g_raw]       # dL/dh0, dh1, dh2, d3
for l in range(4):
    Wz = lin[l].weight.detach().cpu()[:, cin[l]:]       # [out, 48]
    dz_ref.append(dlist[l] @ Wz)
dzp = st[dz_part: dz_part + S * 31 * 32 * 48].reshape(S, 31, 32, 48)[0][:, :B]
sl = [(0, 8), (8, 16), (16, 24), (24, 31)]
for l in range(4):
    got = dzp[sl[l][0]:sl[l][1]].sum(0)
    print('dz layer %d: err %.3e scale %.3e' % (l, (got - dz_ref[l]).abs().max().item(), dz_ref[l].abs().max().item()))
print('g_z total err', (zz.grad[:, 0].cpu() - sum(dz_ref)).abs().max().item(), 'gx err', (p.grad.cpu() - grads[3][:, :339]).abs().max().item())

# ---- the adjoint's exchange buffers of team 0 after the launch (granules [channel][4 rows]{value, tag}) ---------------------------
xch = st[pws + 64: pws + 64 + 93184 // 4]
def gran(off_bytes, C):
    gq = xch[off_bytes // 4: off_bytes // 4 + C * 8].reshape(C, 4, 2)
    return gq[:, :, 0].T, gq[:, :, 1].view(torch.int32).T       # [4 rows][C] values, tags
W3 = lin[3].weight.detach().cpu(); W2 = lin[2].weight.detach().cpu(); W1 = lin[1].weight.detach().cpu(); W0 = lin[0].weight.detach().cpu()
ga3_ref = g_raw @ W3[:, :512]
ga3, t3 = gran(0, 512)
print('ga3 err %.3e scale %.3e tags %s' % ((ga3 - ga3_ref).abs().max().item(), ga3_ref.abs().max().item(), t3.unique().tolist()))
ga2_ref = dh[2] @ W2[:, :1024]
ga2, t2 = gran(512 * 32, 1024)
print('ga2 err %.3e scale %.3e (given the true dh2) tags %s' % ((ga2 - ga2_ref).abs().max().item(), ga2_ref.abs().max().item(), t2.unique().tolist()))
ga1_ref = dh[1] @ W1[:, :1024]
ga1, t1 = gran(512 * 32 + 1024 * 32, 1024)
print('ga1 err %.3e scale %.3e tags %s' % ((ga1 - ga1_ref).abs().max().item(), ga1_ref.abs().max().item(), t1.unique().tolist()))
gx_ref = dh[0] @ W0[:, :339]
gx0, t0 = gran(512 * 32 + 2048 * 32, 352)
print('gx0 err %.3e scale %.3e tags %s' % ((gx0[:, :339] - gx_ref).abs().max().item(), gx_ref.abs().max().item(), t0.unique().tolist()))
# which channels of ga3 are off
bad = ((ga3 - ga3_ref).abs() > 1e-5).nonzero()
print('ga3 mismatches', bad.shape[0], bad[:20].tolist())
