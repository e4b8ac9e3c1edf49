#!/bin/bash
# ThreadSanitizer pass over the WHOLE B <= 32 persistent kernels on the host emulator (one OS thread per work-item: a missing __syncthreads / wave barrier
# around an LDS hand-off is a data race here).  Racy BY DESIGN and therefore suppressed / expected: the tagged-granule exchange through global memory
# (sweep / publish: plain loads polling plain stores) and the adjoint's LDS flag hand-off between its waves (waves 2-3 store dL/dW + the dL/dpR columns,
# then the flag; lane 21 of waves 0-1 polls the flag -- LDS executes a wave's operations in order on the hardware; rollout_persist.hip "hand-off flag").
# usage: bash tools/emu_tsan.sh [B S]
set -e
cd "$(dirname "$0")/.."
CL=/opt/rocm/lib/llvm/bin/clang++
RT=$($CL -print-file-name=libclang_rt.tsan-x86_64.so)
OUT=tests/simt_emu/_emu/libhumor_amd_emu_tsan.so
SRCS=$(ls humor_amd/csrc/*.hip | grep -v rollout_persist.hip)
$CL -x c++ -std=c++20 -O1 -g -fno-omit-frame-pointer -fsanitize=thread -shared-libsan -fPIC -shared -pthread -ffp-contract=off \
  -Wno-unknown-attributes -Wno-ignored-attributes -Wno-pass-failed -I tests/simt_emu/include -o $OUT $SRCS tests/simt_emu/simt_emu.cpp tests/simt_emu/rollout_persist_emu.cpp
SUPP=$(mktemp)
printf 'race:__builtin_amdgcn_raw_buffer_load_b128\nrace:__builtin_amdgcn_raw_buffer_store_b128\nrace:__builtin_amdgcn_raw_buffer_store_b64\nrace:sweep\nrace:sweep_pairs\nrace:publish\nrace:libtorch_cpu\nrace:libgomp\n' > $SUPP
B=${1:-2}; S=${2:-1}
LOG=$(mktemp)
LD_PRELOAD=$RT TSAN_OPTIONS="suppressions=$SUPP history_size=2 halt_on_error=0 exitcode=0" HUMOR_AMD_EMU_LIB=$PWD/$OUT python - > $LOG 2>&1 <<PY
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import rollout_checks as RC
from humor_amd import _lib
lib = _lib.load(os.environ['HUMOR_AMD_EMU_LIB'], emulator=True)
print('persistent kernels, ${B} x ${S}:', RC.check_persistent_kernels_whole_team(lib, ${B}, ${S}, seed=${B} + ${S}), flush=True)
PY
grep "persistent kernels" $LOG
echo "ThreadSanitizer reports by first frame inside the kernels (source line: count):"
grep -A3 "WARNING: ThreadSanitizer" $LOG | grep "#0 void ha::" | sed 's/.*csrc\///; s/ (lib.*//' | sort | uniq -c | sort -rn
echo "other reports (first frame):"
grep -A3 "WARNING: ThreadSanitizer" $LOG | grep "#0 " | grep -v "void ha::" | awk '{print $2}' | sort | uniq -c | sort -rn | head -5
