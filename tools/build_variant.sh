#!/bin/bash
# Builds a variant of libhumor_amd.so next to the product library (tools/microbench/libhumor_amd_<name>.so) with extra -D flags,
# for same-box A/B measurements through HUMOR_AMD_LIB.   usage: tools/build_variant.sh <name> [-DFLAG ...]
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=tools/microbench/libhumor_amd_$name.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared "$@" -o $out humor_amd/csrc/*.hip 2>&1 | grep -v "warning\|^ *[0-9]* |\|\^\|generated" || true
ls -la $out
