"""TEST INFRASTRUCTURE ONLY: builds the kernel sources of humor_amd/csrc for the host SIMT emulator
(tests/simt_emu/include shadows <hip/hip_runtime.h>) -> tests/simt_emu/_emu/libhumor_amd_emu.so."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'humor_amd', 'csrc')
OUT_DIR = os.path.join(HERE, '_emu')
OUT = os.path.join(OUT_DIR, 'libhumor_amd_emu.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def sources():
    # (rollout_persist.hip is compiled through rollout_persist_emu.cpp, which includes it and adds the ha_emu_* test hooks)
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip') and f != 'rollout_persist.hip')


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sources() + [os.path.join(HERE, 'simt_emu.cpp'), os.path.join(HERE, 'rollout_persist_emu.cpp')]
    deps = srcs + [os.path.join(CSRC, 'rollout_persist.hip')] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.inc'))] + \
        [os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'), os.path.join(ROOT, 'include', 'humor_amd.h')]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    cmd = [CLANG, '-x', 'c++', '-std=c++20', '-O2', '-g0', '-fPIC', '-shared', '-pthread', '-ffp-contract=off',
           '-Wno-unknown-attributes', '-Wno-ignored-attributes', '-Wno-pass-failed',
           '-I', os.path.join(HERE, 'include'), '-o', OUT] + srcs
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
