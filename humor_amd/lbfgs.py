"""``LBFGS``: drop-in for ``torch.optim.LBFGS`` on the fitting path (constructor arguments, ``step(closure)`` semantics, state
carried across ``step`` calls, strong-Wolfe line search -- torch/optim/lbfgs.py, which the reference drives from
humor/fitting/motion_optimizer.py:233-254, 284-310, 461-512), restructured for the GPU:

  * the parameters are views of ONE flat buffer (zero-copy when they already lie back to back, as MotionOptimizer allocates
    them), so trying a step is one ``x = x0 + t d`` launch instead of an add per parameter + a copy per parameter back;
  * the two-loop recursion runs in coefficient form (humor_amd/csrc/lbfgs.hip): Gram matrix of the stored pairs kept on the
    device, one GEMV for [S;Y] g, one single-wave kernel for the 2k coefficients, one GEMV for the direction -- ~6 launches
    where torch issues ~4 per stored pair (400 at history 100);
  * every scalar the line search branches on (loss, g.d, max|g|) reaches the host in ONE read per closure evaluation, and the
    interpolation arithmetic is plain Python floats (torch runs it as 0-dim GPU tensor ops with a host sync per comparison).

Measured at C4 (32 x 60): 4.9 ms per closure evaluation inside torch.optim.LBFGS.step for a 0.64 ms stage-1 closure.
Same algorithm, same decisions in exact arithmetic; fp32 summation order differs, so iterates agree with torch's to rounding."""
import ctypes as C

import torch

from . import _lib


def _cubic_interpolate(x1, f1, g1, x2, f2, g2, bounds=None):
    # torch/optim/lbfgs.py:_cubic_interpolate on Python floats
    if bounds is not None:
        xmin_bound, xmax_bound = bounds
    else:
        xmin_bound, xmax_bound = (x1, x2) if x1 <= x2 else (x2, x1)
    d1 = g1 + g2 - 3 * (f1 - f2) / (x1 - x2)
    d2_square = d1 ** 2 - g1 * g2
    if d2_square >= 0:
        d2 = d2_square ** 0.5
        if x1 <= x2:
            min_pos = x2 - (x2 - x1) * ((g2 + d2 - d1) / (g2 - g1 + 2 * d2))
        else:
            min_pos = x1 - (x1 - x2) * ((g1 + d2 - d1) / (g1 - g2 + 2 * d2))
        return min(max(min_pos, xmin_bound), xmax_bound)
    return (xmin_bound + xmax_bound) / 2.0


def _strong_wolfe(obj_func, t, d_norm, f, g, gtd, c1=1e-4, c2=0.9, tolerance_change=1e-9, max_ls=25):
    """torch/optim/lbfgs.py:_strong_wolfe; obj_func(t) -> (f_new: float, g_new: tensor (owned by the caller), gtd_new: float)."""
    f_new, g_new, gtd_new = obj_func(t)
    ls_func_evals = 1
    t_prev, f_prev, g_prev, gtd_prev = 0, f, g, gtd
    done = False
    ls_iter = 0
    while ls_iter < max_ls:
        if f_new > (f + c1 * t * gtd) or (ls_iter > 1 and f_new >= f_prev):
            bracket, bracket_f, bracket_g, bracket_gtd = [t_prev, t], [f_prev, f_new], [g_prev, g_new], [gtd_prev, gtd_new]
            break
        if abs(gtd_new) <= -c2 * gtd:
            bracket, bracket_f, bracket_g = [t], [f_new], [g_new]
            done = True
            break
        if gtd_new >= 0:
            bracket, bracket_f, bracket_g, bracket_gtd = [t_prev, t], [f_prev, f_new], [g_prev, g_new], [gtd_prev, gtd_new]
            break
        min_step = t + 0.01 * (t - t_prev)
        max_step = t * 10
        tmp = t
        t = _cubic_interpolate(t_prev, f_prev, gtd_prev, t, f_new, gtd_new, bounds=(min_step, max_step))
        t_prev, f_prev, g_prev, gtd_prev = tmp, f_new, g_new, gtd_new
        f_new, g_new, gtd_new = obj_func(t)
        ls_func_evals += 1
        ls_iter += 1
    if ls_iter == max_ls:
        bracket, bracket_f, bracket_g = [0, t], [f, f_new], [g, g_new]
    insuf_progress = False
    low_pos, high_pos = (0, 1) if bracket_f[0] <= bracket_f[-1] else (1, 0)
    while not done and ls_iter < max_ls:
        if abs(bracket[1] - bracket[0]) * d_norm < tolerance_change:
            break
        t = _cubic_interpolate(bracket[0], bracket_f[0], bracket_gtd[0], bracket[1], bracket_f[1], bracket_gtd[1])
        eps = 0.1 * (max(bracket) - min(bracket))
        if min(max(bracket) - t, t - min(bracket)) < eps:
            if insuf_progress or t >= max(bracket) or t <= min(bracket):
                t = max(bracket) - eps if abs(t - max(bracket)) < abs(t - min(bracket)) else min(bracket) + eps
                insuf_progress = False
            else:
                insuf_progress = True
        else:
            insuf_progress = False
        f_new, g_new, gtd_new = obj_func(t)
        ls_func_evals += 1
        ls_iter += 1
        if f_new > (f + c1 * t * gtd) or f_new >= bracket_f[low_pos]:
            bracket[high_pos], bracket_f[high_pos], bracket_g[high_pos], bracket_gtd[high_pos] = t, f_new, g_new, gtd_new
            low_pos, high_pos = (0, 1) if bracket_f[0] <= bracket_f[1] else (1, 0)
        else:
            if abs(gtd_new) <= -c2 * gtd:
                done = True
            elif gtd_new * (bracket[high_pos] - bracket[low_pos]) >= 0:
                bracket[high_pos], bracket_f[high_pos] = bracket[low_pos], bracket_f[low_pos]
                bracket_g[high_pos], bracket_gtd[high_pos] = bracket_g[low_pos], bracket_gtd[low_pos]
            bracket[low_pos], bracket_f[low_pos], bracket_g[low_pos], bracket_gtd[low_pos] = t, f_new, g_new, gtd_new
    t = bracket[low_pos]
    return bracket_f[low_pos], bracket_g[low_pos], t, ls_func_evals


def flat_arena(shapes, device, dtype=torch.float32):
    """One flat buffer and views of the given shapes laid out back to back (what LBFGS binds without copying)."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = torch.zeros(sum(sizes), dtype=dtype, device=device)
    views, o = [], 0
    for s, n in zip(shapes, sizes):
        views.append(flat[o:o + n].view(s))
        o += n
    return flat, views


class LBFGS:
    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100,
                 line_search_fn=None, _lib_override=None):
        self._params = list(params)
        if not self._params:
            raise ValueError('optimizer got an empty parameter list')
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        if line_search_fn not in (None, 'strong_wolfe'):
            raise RuntimeError("only 'strong_wolfe' is supported")
        if history_size > 128:
            raise ValueError('history_size must be <= 128 (ha_lbfgs_coeffs)')
        self.param_groups = [dict(params=self._params, lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                                  tolerance_change=tolerance_change, history_size=history_size, line_search_fn=line_search_fn)]
        self.state = {'func_evals': 0, 'n_iter': 0}
        self._lib = _lib_override
        self._flat = None
        self._hist = None
        # optional host-side timeline (tools/lbfgs_eval_breakdown.py): set to a dict to accumulate seconds per phase
        self.profile = None

    def _tick(self, key, t0):
        import time
        t1 = time.perf_counter()
        if self.profile is not None:
            self.profile[key] = self.profile.get(key, 0.0) + (t1 - t0)
        return t1

    # ---- flat parameter buffer ------------------------------------------------------------------------------------------------
    def _bind(self):
        """Makes every parameter a view of one flat buffer.  No copy when they already lie back to back in one allocation."""
        ps = self._params
        if self._flat is not None:
            o, ok = 0, True
            for p in ps:
                ok = ok and p.data_ptr() == self._flat.data_ptr() + 4 * o and p.is_contiguous()
                o += p.numel()
            if ok:
                return
        o, consecutive = ps[0].data_ptr(), all(p.is_contiguous() and p.dtype == torch.float32 for p in ps)
        for p in ps:
            consecutive = consecutive and p.data_ptr() == o
            o += 4 * p.numel()
        n = sum(p.numel() for p in ps)
        if consecutive and ps[0].untyped_storage().data_ptr() == ps[-1].untyped_storage().data_ptr():
            off = (ps[0].data_ptr() - ps[0].untyped_storage().data_ptr()) // 4
            self._flat = torch.empty(0, dtype=torch.float32, device=ps[0].device).set_(ps[0].untyped_storage(), off, (n,), (1,))
        else:
            self._flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
            o = 0
            for p in ps:
                p.data = self._flat[o:o + p.numel()].view(p.shape)
                o += p.numel()

    def _gather_flat_grad(self):
        views = [p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), dtype=torch.float32, device=p.device) for p in self._params]
        return torch.cat(views, 0)

    # ---- history (device Gram matrix) ------------------------------------------------------------------------------------------
    def _init_history(self, n, device):
        h = self.param_groups[0]['history_size']
        self._hist = {'M': torch.zeros(2 * h, n, dtype=torch.float32, device=device), 'G': torch.zeros(2 * h, 2 * h, dtype=torch.float32, device=device),
                      'order': [], 'coef': torch.zeros(2 * h, dtype=torch.float32, device=device), 'h': h}

    def _pop_pair(self, undo):
        """Drops the pair stored by the matching _push_pair (its slot's Gram entries become dead: slots outside `order` are ignored).
        An evicted oldest pair is not restored -- torch would have kept it; this only happens when the curvature test fails with a
        full history, and only shortens the memory by one pair."""
        self._hist['order'].remove(undo)

    def _push_pair(self, s, y):
        H = self._hist
        h, order = H['h'], H['order']
        slot = order.pop(0) if len(order) == h else next(i for i in range(h) if i not in order)
        order.append(slot)
        M, G = H['M'], H['G']
        M[slot].copy_(s)
        M[h + slot].copy_(y)
        vs, vy = torch.mv(M, M[slot]), torch.mv(M, M[h + slot])
        G[slot], G[:, slot] = vs, vs
        G[h + slot], G[:, h + slot] = vy, vy
        return slot

    def _direction(self, g, h_diag):
        """d = [S;Y]^T coef - h_diag g.  h_diag: Python float, or a 0-dim device tensor (then it is read on the device: no host sync)."""
        H = self._hist
        lib = self._lib if self._lib is not None else _lib.get_lib()
        Mg = torch.mv(H['M'], g)
        order = (C.c_int32 * max(1, len(H['order'])))(*H['order'])
        on_dev = torch.is_tensor(h_diag)
        hd = h_diag.reshape(1).float().contiguous() if on_dev else None
        lib.call('ha_lbfgs_coeffs', H['h'], len(H['order']), order, _lib.ptr(H['G']), _lib.ptr(Mg), 0.0 if on_dev else float(h_diag),
                 _lib.ptr(hd), _lib.ptr(H['coef']), _lib.stream_ptr(g))
        if on_dev:
            return torch.addmv(g * (-hd), H['M'].t(), H['coef'])
        return torch.addmv(g, H['M'].t(), H['coef'], beta=-float(h_diag), alpha=1.0)

    # ---- step ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure):
        group = self.param_groups[0]
        lr, max_iter, max_eval = float(group['lr']), group['max_iter'], group['max_eval']
        tolerance_grad, tolerance_change = group['tolerance_grad'], group['tolerance_change']
        line_search_fn = group['line_search_fn']
        closure = torch.enable_grad()(closure)
        state = self.state
        self._bind()
        x = self._flat

        orig_loss = closure()
        flat_grad = self._gather_flat_grad()
        loss, gmax = torch.stack([orig_loss.detach().reshape(()).float(), flat_grad.abs().max()]).tolist()
        current_evals = 1
        state['func_evals'] += 1
        if gmax <= tolerance_grad:
            return orig_loss
        if self._hist is None or self._hist['M'].shape[1] != x.numel() or self._hist['M'].device != x.device:
            self._init_history(x.numel(), x.device)
        d, t = state.get('d'), state.get('t')
        H_diag = state.get('H_diag', 1.0)
        prev_flat_grad, prev_loss = state.get('prev_flat_grad'), state.get('prev_loss')

        n_iter = 0
        import time
        tp = time.perf_counter()
        while n_iter < max_iter:
            n_iter += 1
            state['n_iter'] += 1
            pushed = None
            tp = self._tick('other', tp)
            if state['n_iter'] == 1:
                d = flat_grad.neg()
                self._hist['order'] = []
                H_diag = 1.0
                scal = [flat_grad.dot(d), d.abs().max(), flat_grad.abs().sum()]
            else:
                # Speculative update: the pair is stored and the direction built with H = ys / yy computed on the device, and the
                # curvature test ys > 1e-10 is read back together with g.d and max|d| -- ONE host read per iteration.  When the
                # test fails (rare) the pair is dropped and the direction rebuilt with the previous scaling, as torch does.
                y = flat_grad.sub(prev_flat_grad)
                s = d.mul(t)
                ysyy = torch.stack([y.dot(s), y.dot(y)])
                pushed = self._push_pair(s, y)
                d = self._direction(flat_grad, ysyy[0] / ysyy[1])
                scal = [flat_grad.dot(d), d.abs().max(), ysyy[0], ysyy[1]]
            if prev_flat_grad is None:
                prev_flat_grad = flat_grad.clone()
            else:
                prev_flat_grad.copy_(flat_grad)
            prev_loss = loss

            tp = self._tick('direction_issue', tp)
            vals = torch.stack(scal).tolist()
            tp = self._tick('direction_wait', tp)
            gtd, d_norm = vals[0], vals[1]
            if state['n_iter'] == 1:
                t = min(1.0, 1.0 / vals[2]) * lr
            else:
                ys, yy = vals[2], vals[3]
                if ys > 1e-10:
                    H_diag = ys / yy
                else:
                    self._pop_pair(pushed)
                    d = self._direction(flat_grad, H_diag)
                    gtd, d_norm = torch.stack([flat_grad.dot(d), d.abs().max()]).tolist()
                t = lr
            if gtd > -tolerance_change:
                break

            ls_func_evals = 0
            if line_search_fn is not None:
                x_init = x.clone()
                gmax_of = {}

                def obj_func(tt):
                    t0 = self._tick('other', time.perf_counter()) if self.profile is None else self._tick('other', self._tp)
                    torch.add(x_init, d, alpha=tt, out=x)
                    l = closure()
                    g_new = self._gather_flat_grad()
                    t0 = self._tick('closure_issue', t0)
                    f_new, gtd_new, gm = torch.stack([l.detach().reshape(()).float(), g_new.dot(d), g_new.abs().max()]).tolist()
                    self._tp = self._tick('closure_wait', t0)
                    gmax_of[id(g_new)] = gm
                    return f_new, g_new, gtd_new
                self._tp = tp
                g_in = flat_grad
                loss, flat_grad, t, ls_func_evals = _strong_wolfe(obj_func, t, d_norm, loss, flat_grad, gtd, max_ls=max_eval - current_evals)
                tp = self._tp
                torch.add(x_init, d, alpha=t, out=x)
                gmax = gmax_of[id(flat_grad)] if id(flat_grad) in gmax_of else (gmax if flat_grad is g_in else flat_grad.abs().max().item())
            else:
                x.add_(d, alpha=t)
                if n_iter != max_iter:
                    l = closure()
                    flat_grad = self._gather_flat_grad()
                    loss, gmax = torch.stack([l.detach().reshape(()).float(), flat_grad.abs().max()]).tolist()
                    ls_func_evals = 1
            current_evals += ls_func_evals
            state['func_evals'] += ls_func_evals

            if n_iter == max_iter:
                break
            if current_evals >= max_eval:
                break
            if gmax <= tolerance_grad:
                break
            if d_norm * abs(t) <= tolerance_change:
                break
            if abs(loss - prev_loss) < tolerance_change:
                break

        state['d'], state['t'], state['H_diag'] = d, t, H_diag
        state['prev_flat_grad'], state['prev_loss'] = prev_flat_grad, prev_loss
        return orig_loss

    def zero_grad(self, set_to_none=True):
        for p in self._params:
            p.grad = None
