"""``LBFGS``: drop-in for ``torch.optim.LBFGS`` on the fitting path (constructor arguments, ``step(closure)`` semantics, state
carried across ``step`` calls, strong-Wolfe line search -- torch/optim/lbfgs.py, which the reference drives from
humor/fitting/motion_optimizer.py:233-254, 284-310, 461-512), restructured for the GPU:

  * the parameters are views of ONE flat buffer (zero-copy when they already lie back to back, as MotionOptimizer allocates
    them), so trying a step is one ``x = x0 + t d`` launch instead of an add per parameter + a copy per parameter back;
  * everything between two closure evaluations runs in a handful of launches of humor_amd/csrc/lbfgs.hip: the new curvature pair is
    written straight into the history M = [S; Y; g], ONE pass over M gives M s, M y, M g (fixed summation order -- the replicated
    multi-GPU optimiser needs bit-identical directions on every rank), one launch installs the pair in the device Gram matrix and runs
    the two-loop recursion in coefficient form, and the direction is one GEMV d = M^T coef -- ~8 launches where torch issues ~4 per
    stored pair (400 at history 100);
  * every scalar the line search branches on (loss, g.d, max|g|; g.d, max|d|, y.s, y.y) comes from one small kernel per evaluation /
    direction, and the first trial evaluation of an iteration is issued right behind the direction kernels: ONE host read per inner
    iteration; the interpolation arithmetic is plain Python floats (torch runs it as 0-dim GPU tensor ops with a host sync per
    comparison).  A closure whose owner wants exact evaluation counters may carry a `discard_last` attribute: it is called in the
    rare cases where torch would have stopped before the trial evaluation that was already issued;
  * optionally (`speculate`, default OFF) the host read itself is taken off the critical path: the common outcome of an inner iteration is "first
    trial step t = lr accepted", and everything the NEXT iteration issues in that case -- pair (y = g_trial - g, s = lr d), direction,
    first trial evaluation -- needs no host value.  It is issued BEFORE the current iteration's scalars are read, so the GPU works on
    iteration k + 1 while the host decides on k.  Any other outcome (line search continues, curvature test fails, a stopping test
    fires) rolls the speculative iteration back: its pair is dropped from the history, the retired oldest pair restored, the gradient
    row reset, its evaluation reported through `discard_last`; the iterates are bit-identical to the non-speculative run.  Measured on
    MI355X at C4 (tools/lbfgs_phase_profile.py, profiles/r05_lbfgs): no gain -- 2.13-2.17 ms per stage-3 evaluation against 2.02-2.06
    without (94 speculative iterations, none rolled back): since round 4 an evaluation inside step() is GPU-bound (1.73 ms closure +
    ~0.3 ms of the optimiser's own kernels: two passes over the 78 MB history, the coefficient kernel, ten small launches), the host's
    decision latency is no longer on the critical path -- hence off by default.

Measured at C4 (32 x 60): 4.9 ms per closure evaluation inside torch.optim.LBFGS.step for a 0.64 ms stage-1 closure; here 0.37 ms
(stage 1) / 0.46 ms (stage 2) per evaluation including the closure.
Same algorithm, same decisions in exact arithmetic; fp32 summation order differs, so iterates agree with torch's to rounding."""
import ctypes as C
import math
import time

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


def _strong_wolfe(obj_func, t, d_norm, f, g, gtd, c1=1e-4, c2=0.9, tolerance_change=1e-9, max_ls=25, first=None):
    """torch/optim/lbfgs.py:_strong_wolfe; obj_func(t) -> (f_new: float, g_new: tensor (owned by the caller), gtd_new: float).
    `first`: the result of obj_func(t) when the caller has already evaluated the first trial point."""
    f_new, g_new, gtd_new = obj_func(t) if first is None else first
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
    _SPARE = 2          # slots beyond the history size (two allocations can be open at a time)

    def __init__(self, params, lr=1, max_iter=20, max_eval=None, tolerance_grad=1e-7, tolerance_change=1e-9, history_size=100,
                 line_search_fn=None, _lib_override=None):
        self._params = list(params)
        if not self._params:
            raise ValueError('optimizer got an empty parameter list')
        if max_eval is None:
            max_eval = max_iter * 5 // 4
        if line_search_fn not in (None, 'strong_wolfe'):
            raise RuntimeError("only 'strong_wolfe' is supported")
        if history_size > 126:
            raise ValueError('history_size must be <= 126 (ha_lbfgs_coeffs holds history_size + 2 slots, at most 128)')
        self.param_groups = [dict(params=self._params, lr=lr, max_iter=max_iter, max_eval=max_eval, tolerance_grad=tolerance_grad,
                                  tolerance_change=tolerance_change, history_size=history_size, line_search_fn=line_search_fn)]
        self.state = {'func_evals': 0, 'n_iter': 0}
        self._lib = _lib_override
        self._flat = None
        self._hist = None
        # True: issue the next inner iteration before the current one's scalars are read (see the module docstring: no gain measured)
        self.speculate = False
        self.spec_stats = {'issued': 0, 'rolled_back': 0}
        # optional host-side timeline (tools/lbfgs_eval_breakdown.py): set to a dict to accumulate seconds per phase
        self.profile = None

    def _tick(self, key, t0):
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
        # two slots more than the history holds: with a full history a new pair goes to a spare slot and the oldest pair it retires stays
        # intact until that allocation can no longer be undone -- a failed curvature test restores it (torch.optim.LBFGS keeps it in that
        # case), and so does the roll-back of a speculatively issued iteration: two allocations can be open at a time
        hmax = self.param_groups[0]['history_size']
        h = hmax + self._SPARE
        lib = self._lib if self._lib is not None else _lib.get_lib()
        npart = C.c_int64()
        lib.call('ha_lbfgs_gram_workspace', n, 2 * h, C.byref(npart))
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
        # M: rows 0..h-1 the s slots, h..2h-1 the y slots, row 2h the current gradient (it doubles as torch's prev_flat_grad)
        self._hist = {'M': z(2 * h + 1, n), 'G': z(2 * h, 2 * h), 'order': [], 'coef': z(2 * h + 1), 'h': h, 'Mg': z(2 * h),
                      'P': z(2 * h, 3), 'part': z(npart.value), 'scal': z(2, 12), 'hmax': hmax, 'evicted': []}

    def _pop_pair(self, undo):
        """Drops the pair stored by the matching _alloc_slot -- the most recent one still open -- (its slot's Gram entries become dead:
        slots outside `order` are ignored) and puts back the oldest pair that allocation had retired from a full history (its slot was
        not touched), as torch.optim.LBFGS keeps it when the curvature test fails."""
        H = self._hist
        slot, evicted = H['evicted'].pop()
        assert slot == undo, 'allocations are undone in reverse order'
        H['order'].remove(undo)
        if evicted is not None:
            H['order'].insert(0, evicted)

    def _alloc_slot(self):
        H = self._hist
        h, order, open_ = H['h'], H['order'], H['evicted']
        if len(open_) == 2:            # the older of two open allocations can no longer be undone: the pair it retired is gone for good
            open_.pop(0)
        evicted = order.pop(0) if len(order) == H['hmax'] else None       # (its rows stay intact while the allocation is open)
        reserved = {e for _, e in open_ if e is not None}
        slot = next(i for i in range(h) if i not in order and i != evicted and i not in reserved)
        order.append(slot)
        open_.append((slot, evicted))
        return slot

    def _direction(self, g, h_diag):
        """d = [S;Y]^T coef - h_diag g.  h_diag: Python float, or a 0-dim device tensor (then it is read on the device: no host sync)."""
        H = self._hist
        lib = self._lib if self._lib is not None else _lib.get_lib()
        M2 = H['M'][:2 * H['h']]
        Mg = torch.mv(M2, g)
        order = (C.c_int32 * max(1, len(H['order'])))(*H['order'])
        on_dev = torch.is_tensor(h_diag)
        hd = h_diag.reshape(1).float().contiguous() if on_dev else None
        lib.call('ha_lbfgs_coeffs', H['h'], len(H['order']), order, _lib.ptr(H['G']), _lib.ptr(Mg), 0.0 if on_dev else float(h_diag),
                 _lib.ptr(hd), _lib.ptr(H['coef']), _lib.stream_ptr(g))
        if on_dev:
            return torch.addmv(g * (-hd), M2.t(), H['coef'][:2 * H['h']])
        return torch.addmv(g, M2.t(), H['coef'][:2 * H['h']], beta=-float(h_diag), alpha=1.0)

    def _scalars(self, a, b, extra=None, out=None):
        """[a.b, max|a|, sum|a|, extra] (ha_lbfgs_scalars) as a device buffer of 4 floats; the caller reads it with ONE .tolist()."""
        lib = self._lib if self._lib is not None else _lib.get_lib()
        out = torch.empty(4, dtype=torch.float32, device=a.device) if out is None else out
        lib.call('ha_lbfgs_scalars', a.numel(), _lib.ptr(a), _lib.ptr(b), _lib.ptr(extra) if extra is not None else None, _lib.ptr(out),
                 _lib.stream_ptr(a))
        return out

    def _pair_direction(self, slot, scal):
        """Installs the pair already written to rows slot / h + slot of M (gradient in row 2h) and returns the new direction
        d = M^T coef (coef[2h] = -y.s / y.y); scal[4:6] = (y.s, y.y).  Four launches: Gram pass, its reduction, coefficients, GEMV."""
        H = self._hist
        lib = self._lib if self._lib is not None else _lib.get_lib()
        h, M = H['h'], H['M']
        st = _lib.stream_ptr(M)
        lib.call('ha_lbfgs_gram', M.shape[1], 2 * h, _lib.ptr(M), slot, h + slot, 2 * h, _lib.ptr(H['part']), _lib.ptr(H['P']), st)
        order = (C.c_int32 * len(H['order']))(*H['order'])
        lib.call('ha_lbfgs_pair_coeffs', h, len(H['order']), order, slot, _lib.ptr(H['P']), _lib.ptr(H['G']), _lib.ptr(H['Mg']),
                 _lib.ptr(H['coef']), _lib.ptr(scal[4:]), st)
        return torch.mv(M.t(), H['coef'])

    # ---- step ------------------------------------------------------------------------------------------------------------------
    def _issue_iteration(self, closure, x, flat_grad, d_prev, t_prev, lr, scal):
        """Everything an inner iteration (from the second one on, with a line search) issues before it needs a host value: the pair
        y = g - g_prev, s = t d of the step that led here, the new direction with its scalars (scal[0:2] = g.d, max|d|; scal[4:6] =
        y.s, y.y) and the first trial evaluation at t = lr (scal[8:12] = g_new.d, max|g_new|, sum|g_new|, loss)."""
        Hh = self._hist
        M, hsz = Hh['M'], Hh['h']
        g_row = M[2 * hsz]
        pushed = self._alloc_slot()
        torch.sub(flat_grad, g_row, out=M[hsz + pushed])             # y = g - g_prev
        torch.mul(d_prev, t_prev, out=M[pushed])                     # s = t d
        g_row.copy_(flat_grad)
        d = self._pair_direction(pushed, scal)
        self._scalars(d, flat_grad, None, scal[:4])
        x_init = x.clone()
        torch.add(x_init, d, alpha=lr, out=x)
        l = closure()
        g_new = self._gather_flat_grad()
        self._scalars(g_new, d, l.detach().reshape(1).float(), scal[8:12])
        return {'pushed': pushed, 'd': d, 'x_init': x_init, 'g_new': g_new, 'scal': scal, 'g_in': flat_grad}

    def _roll_back(self, it, g_row_value, discard):
        """Undoes a speculatively issued iteration (see the module docstring): its pair leaves the history (the oldest pair it retired
        returns), the gradient row gets the gradient it held before, the evaluation is reported as discarded.  The caller restores x
        (every continuation writes x from the current iteration's x_init)."""
        self._pop_pair(it['pushed'])
        H = self._hist
        H['M'][2 * H['h']].copy_(g_row_value)
        # (a speculative iteration may have been built on a direction the host then rejects -- curvature test failed: H = y.s / y.y of a
        # vanishing pair -- and hold non-finite rows; dead rows are multiplied by zero coefficients in d = M^T coef, so they must be finite)
        slot, h = it['pushed'], H['h']
        H['M'][slot].zero_()
        H['M'][h + slot].zero_()
        # (... and so must everything ha_lbfgs_gram wrote for the dead slot: its Gram rows / columns, its projections and the coefficient
        # vector built with it -- ha_lbfgs_coeffs multiplies entries of slots outside `order` by zero, and 0 x NaN poisons every later
        # direction (advisor, round 5: repeated step() calls at a converged point, y = 0 -> H = 0 / 0))
        dead = [slot, h + slot]
        H['G'][dead, :] = 0.0
        H['G'][:, dead] = 0.0
        H['P'][dead, :] = 0.0
        H['Mg'][dead] = 0.0
        H['coef'].zero_()
        if discard is not None:
            discard()
        self.spec_stats['rolled_back'] += 1

    @torch.no_grad()
    def step(self, closure):
        group = self.param_groups[0]
        lr, max_iter, max_eval = float(group['lr']), group['max_iter'], group['max_eval']
        tolerance_grad, tolerance_change = group['tolerance_grad'], group['tolerance_change']
        line_search_fn = group['line_search_fn']
        discard = getattr(closure, 'discard_last', None)     # told when a speculatively issued evaluation is thrown away
        closure = torch.enable_grad()(closure)
        state = self.state
        self._bind()
        x = self._flat

        orig_loss = closure()
        flat_grad = self._gather_flat_grad()
        if self._hist is None or self._hist['M'].shape[1] != x.numel() or self._hist['M'].device != x.device:
            self._init_history(x.numel(), x.device)
            state['n_iter'] = 0          # a new variable vector: the stored pairs (and torch's persistent direction) no longer apply
        Hh = self._hist
        M, hsz = Hh['M'], Hh['h']
        g_row = M[2 * hsz]               # the previous gradient (torch's prev_flat_grad) until it is overwritten below
        n_issued = 0                     # iterations issued through _issue_iteration so far (they alternate between the two scalar buffers)
        scal = Hh['scal'][0]
        loss_t = orig_loss.detach().reshape(1).float()
        gg, gmax, gsum, loss = self._scalars(flat_grad, flat_grad, loss_t, scal[:4]).tolist()
        # host copies of what this call has read anyway (no extra synchronisation): the objective and the largest gradient entry at the
        # call's first evaluation -- MotionOptimizer.run raises on a non-finite one instead of returning NaN results
        self.last_loss, self.last_gmax = loss, gmax
        current_evals = 1
        state['func_evals'] += 1
        if not math.isfinite(loss):
            # (ha_lbfgs_scalars' maximum ignores NaN: a NaN gradient reads as max|g| = 0 -- "converged".  Nothing useful can follow a
            # non-finite objective; the caller is told through last_loss)
            self.last_gmax = float('nan')
            return orig_loss
        if gmax <= tolerance_grad:
            return orig_loss
        d, t = state.get('d'), state.get('t')
        H_diag = state.get('H_diag', 1.0)
        prev_loss = state.get('prev_loss')

        n_iter = 0
        pending = None                   # the next iteration, issued speculatively
        tp = time.perf_counter()
        while n_iter < max_iter:
            n_iter += 1
            state['n_iter'] += 1
            pushed = None
            it = None
            tp = self._tick('other', tp)
            # From the second iteration on the first trial step is always t = lr, and the direction's scalars are only needed for two
            # rare decisions (curvature test failed / directional derivative ~ 0): with a line search the first trial evaluation is
            # issued right behind the direction kernels and ONE read returns both sets of scalars -- one host round trip per iteration.
            deferred = state['n_iter'] > 1 and line_search_fn is not None
            if state['n_iter'] == 1:
                d = flat_grad.neg()
                Hh['order'] = []
                Hh['evicted'] = []
                H_diag = 1.0
                g_row.copy_(flat_grad)
            elif deferred:
                if pending is not None:
                    it, pending = pending, None          # issued while the previous iteration's scalars were on their way
                else:
                    it = self._issue_iteration(closure, x, flat_grad, d, t, lr, Hh['scal'][n_issued & 1])
                    n_issued += 1
                pushed, d, scal = it['pushed'], it['d'], it['scal']
            else:
                # (no line search) speculative update: the pair is stored and the direction built with H = ys / yy computed on the device
                scal = Hh['scal'][0]
                pushed = self._alloc_slot()
                torch.sub(flat_grad, g_row, out=M[hsz + pushed])             # y = g - g_prev
                torch.mul(d, t, out=M[pushed])                                # s = t d
                g_row.copy_(flat_grad)
                d = self._pair_direction(pushed, scal)
                self._scalars(d, flat_grad, None, scal[:4])                   # g.d, max|d|
            prev_loss = loss

            tp = self._tick('direction_issue', tp)
            if state['n_iter'] == 1:
                gtd, d_norm = -gg, gmax              # d = -g: g.d, max|d| and sum|g| came with the first evaluation's read
                t = min(1.0, 1.0 / gsum) * lr
            elif not deferred:
                vals = scal.tolist()
                tp = self._tick('direction_wait', tp)
                gtd, d_norm = vals[0], vals[1]
                ys, yy = vals[4], vals[5]
                if ys > 1e-10:
                    H_diag = ys / yy
                else:
                    self._pop_pair(pushed)
                    d = self._direction(flat_grad, H_diag)
                    gtd, d_norm = self._scalars(d, flat_grad).tolist()[:2]
                t = lr
            else:
                t = lr
            if not deferred and gtd > -tolerance_change:
                break

            ls_func_evals = 0
            stop = False
            if line_search_fn is not None:
                x_init = x.clone() if it is None else it['x_init']
                gmax_of = {}

                def issue(tt, out=None):
                    torch.add(x_init, d, alpha=tt, out=x)
                    l = closure()
                    g_new = self._gather_flat_grad()
                    return g_new, self._scalars(g_new, d, l.detach().reshape(1).float(), out)

                def obj_func(tt):
                    t0 = self._tick('other', time.perf_counter()) if self.profile is None else self._tick('other', self._tp)
                    g_new, ev = issue(tt)
                    t0 = self._tick('closure_issue', t0)
                    gtd_new, gm, _, f_new = ev.tolist()
                    self._tp = self._tick('closure_wait', t0)
                    gmax_of[id(g_new)] = gm
                    return f_new, g_new, gtd_new
                self._tp = tp
                g_in = flat_grad
                first = None
                if deferred:
                    g_new = it['g_new']
                    # the NEXT iteration, for the common outcome "first trial accepted": issued before this iteration's scalars are read.
                    # (Not behind the last iteration of this call, and not when accepting the trial exhausts the evaluation budget.)
                    spec = None
                    if self.speculate and n_iter < max_iter and current_evals + 1 < max_eval:
                        spec = self._issue_iteration(closure, x, g_new, d, lr, lr, Hh['scal'][n_issued & 1])
                        n_issued += 1
                        self.spec_stats['issued'] += 1
                    tp = self._tick('closure_issue', tp)
                    vals = scal.tolist()
                    self._tp = tp = self._tick('closure_wait', tp)
                    gtd, d_norm, ys, yy = vals[0], vals[1], vals[4], vals[5]
                    if ys > 1e-10:
                        H_diag = ys / yy
                        first = (vals[11], g_new, vals[8])
                        gmax_of[id(g_new)] = vals[9]
                    else:
                        # curvature test failed: torch keeps the old pairs and scaling -- drop the pair, rebuild the direction, and let
                        # the line search start over from x_init (the trial evaluation along the discarded direction is not counted)
                        if spec is not None:
                            self._roll_back(spec, flat_grad, discard)
                            spec = None
                        x.copy_(x_init)
                        if discard is not None:
                            discard()
                        self._pop_pair(pushed)
                        d = self._direction(flat_grad, H_diag)
                        gtd, d_norm = self._scalars(d, flat_grad).tolist()[:2]
                    if gtd > -tolerance_change:
                        if spec is not None:
                            self._roll_back(spec, flat_grad, discard)
                            spec = None
                        if first is not None:
                            x.copy_(x_init)              # torch stops before the line search: the trial step is discarded
                            if discard is not None:
                                discard()
                        break
                    if spec is not None:
                        # does the first trial end the line search?  (_strong_wolfe's first tests, on the scalars just read)
                        f_new, gtd_new = first[0], first[2]
                        accepted = not (f_new > (loss + 1e-4 * t * gtd)) and abs(gtd_new) <= -0.9 * gtd
                        if not accepted:
                            self._roll_back(spec, flat_grad, discard)
                            spec = None
                loss, flat_grad, t, ls_func_evals = _strong_wolfe(obj_func, t, d_norm, loss, flat_grad, gtd, max_ls=max_eval - current_evals,
                                                                  first=first)
                tp = self._tp
                gmax = gmax_of[id(flat_grad)] if id(flat_grad) in gmax_of else (gmax if flat_grad is g_in else flat_grad.abs().max().item())
                if deferred and spec is not None:
                    # the speculative iteration stands unless a stopping test fires now (x already holds ITS trial point)
                    assert ls_func_evals == 1 and t == lr and flat_grad is it['g_new']
                    stop = (current_evals + 1 >= max_eval or gmax <= tolerance_grad or d_norm * abs(t) <= tolerance_change or
                            abs(loss - prev_loss) < tolerance_change)
                    if stop:
                        self._roll_back(spec, g_in, discard)
                        torch.add(x_init, d, alpha=t, out=x)
                    else:
                        pending = spec
                else:
                    torch.add(x_init, d, alpha=t, out=x)
            else:
                x.add_(d, alpha=t)
                if n_iter != max_iter:
                    l = closure()
                    flat_grad = self._gather_flat_grad()
                    _, gmax, _, loss = self._scalars(flat_grad, flat_grad, l.detach().reshape(1).float()).tolist()
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

        assert pending is None
        if not (math.isfinite(loss) and math.isfinite(gmax)):
            self.last_loss, self.last_gmax = loss, gmax          # (a later evaluation of this call went non-finite)
        state['d'], state['t'], state['H_diag'] = d, t, H_diag
        state['prev_loss'] = prev_loss
        return orig_loss

    def zero_grad(self, set_to_none=True):
        for p in self._params:
            p.grad = None
