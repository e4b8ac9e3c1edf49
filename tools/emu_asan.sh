#!/bin/bash
# AddressSanitizer pass over the WHOLE persistent / pipelined roll-out kernels on the host emulator (CPU build only: GPU ASan is not available on this
# pool).  Every block's LDS is its own heap buffer and the hook's stash / state / exchange buffers are exactly sized vectors, so an out-of-bounds LDS or
# global access of the kernels is a heap-buffer-overflow report here.   usage: bash tools/emu_asan.sh [B S] [pipe B S]
set -e
cd "$(dirname "$0")/.."
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$($CL -print-file-name=libclang_rt.asan-x86_64.so)
OUT=tests/simt_emu/_emu/libhumor_amd_emu_asan.so
SRCS=$(ls humor_amd/csrc/*.hip | grep -v rollout_persist.hip)
$CL -x c++ -std=c++20 -O1 -g -fno-omit-frame-pointer -fsanitize=address -shared-libasan -fPIC -shared -pthread -ffp-contract=off \
  -Wno-unknown-attributes -Wno-ignored-attributes -Wno-pass-failed -I tests/simt_emu/include -o $OUT $SRCS tests/simt_emu/simt_emu.cpp tests/simt_emu/rollout_persist_emu.cpp
B=${1:-3}; S=${2:-2}; PB=${3:-0}; PS=${4:-1}
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 HUMOR_AMD_EMU_LIB=$PWD/$OUT python - <<PY
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import rollout_checks as RC
from humor_amd import _lib
lib = _lib.load(os.environ['HUMOR_AMD_EMU_LIB'], emulator=True)
print('persistent kernels, ${B} x ${S}:', RC.check_persistent_kernels_whole_team(lib, ${B}, ${S}, seed=${B} + ${S}), flush=True)
if ${PB} > 32:
    print('pipelined kernels, ${PB} x ${PS}:', RC.check_pipelined_kernels_whole_team(lib, ${PB}, ${PS}, seed=${PB} + ${PS}), flush=True)
print('AddressSanitizer: no report')
PY
