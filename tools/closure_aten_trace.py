#!/usr/bin/env python
"""Every ATen op of one stage-3 objective evaluation (forward + backward) that launches a kernel, with the place that issues it: the
innermost humor_amd frame in the forward pass, the autograd node in the backward pass.  Runs on the CPU through the host SIMT-emulator
build of the kernels (tests/simt_emu) at a small size -- the op structure does not depend on the size.
usage: closure_aten_trace.py [B T]"""
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'simt_emu'))

VIEWS = ('view', 'reshape', 'expand', 'slice', 'select', 'permute', 'transpose', 'detach', 'alias', 'unsqueeze', 'squeeze', 'as_strided',
         '_unsafe_view', 'split', 'split_with_sizes', 't', 'unbind', 'narrow', 'empty', 'empty_like', 'empty_strided', 'new_empty',
         'lift_fresh', '_to_copy', 'is_same_size', 'sym_size', 'stride', 'size', 'numel', 'dim', 'unfold', 'chunk', 'view_as',
         'new_empty_strided', 'prim', '_local_scalar_dense', 'item', 'result_type', 'equal')


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = []
        self.backward = False

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEWS:
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            where = ''
            node = torch._C._current_autograd_node() if hasattr(torch._C, '_current_autograd_node') else None
            if node is not None:
                where = 'bwd node ' + node.name()
            for fr in reversed(traceback.extract_stack()[:-1]):
                if 'humor_amd/' in fr.filename and 'torch/' not in fr.filename:
                    where += f'  {os.path.basename(fr.filename)}:{fr.lineno} {fr.name}'
                    break
            self.rows.append((name, shapes, where))
        return func(*args, **(kwargs or {}))


def graph_report(opt, case, dev, FC):
    """The inputs of autograd nodes that receive more than one gradient (each extra one is an accumulation `add` launch)."""
    import collections
    captured = {}
    orig = torch.autograd.grad

    def spy(loss, *a, **k):
        captured['loss'] = loss
        return orig(loss, *a, **k)
    torch.autograd.grad = spy
    try:
        FC.eval_stage(opt, case, 2, dev)
    finally:
        torch.autograd.grad = orig
    root = captured['loss'].grad_fn
    indeg = collections.defaultdict(list)
    seen, stack = set(), [root]
    while stack:
        n = stack.pop()
        if n in seen:
            continue
        seen.add(n)
        for (m, idx) in n.next_functions:
            if m is None:
                continue
            indeg[(m, idx)].append(n)
            stack.append(m)
    print(f'{len(seen)} autograd nodes')
    for (m, idx), srcs in sorted(indeg.items(), key=lambda kv: -kv[0][0]._sequence_nr() if hasattr(kv[0][0], '_sequence_nr') else 0):
        if len(srcs) > 1:
            shape = ''
            if hasattr(m, 'variable'):
                shape = f' leaf {tuple(m.variable.shape)}'
            print(f'{m.name()}[input {idx}]{shape} <- ' + ', '.join(s_.name() for s_ in srcs))


def main():
    import build as emu_build
    import fitting_checks as FC
    from humor_amd import _lib, synth
    from oracle import closure_cases as CC
    B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2, 6)
    lib = _lib.load(emu_build.build(), emulator=True)
    dev = torch.device('cpu')
    npz = synth.write_smplh_npz('/tmp/model_at.npz', seed=0)
    case = CC.make_case('rgb', B, T, seed=3)
    opt = FC.build(lib, dev, 'rgb', B, T, npz)
    if '--graph' in sys.argv:
        graph_report(opt, case, dev, FC)
        return
    with Trace() as tr:
        FC.eval_stage(opt, case, 2, dev)
    print(f'{len(tr.rows)} kernel-launching ATen ops in one stage-3 evaluation (B={B}, T={T})')
    for i, (name, shapes, where) in enumerate(tr.rows):
        print(f'{i:3d} {name:22s} {str(shapes)[:70]:70s} {where}')


if __name__ == '__main__':
    main()
