"""Builds libhumor_amd.so (gfx950) in-tree with hipcc.  `python -m humor_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libhumor_amd.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps():
    inc = os.path.join(os.path.dirname(HERE), 'include', 'humor_amd.h')
    return sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc'))] + [inc]


def is_stale():
    return not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in _deps())


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    objs = []
    for src in sources():
        obj = src[:-4] + '.o'
        cmd = [HIPCC, f'--offload-arch={ARCH}', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, f'--offload-arch={ARCH}', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
