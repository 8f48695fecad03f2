#!/bin/bash
# A/B build: libselfocc_hip_<name>.so = the standard objects with ONE source recompiled under extra flags.
#   scripts/build_variant.sh stats render_fwd.hip -DSO_STAGE_STATS        (select with SELFOCC_HIP_LIB=<path>)
set -euo pipefail
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../selfocc_amd/csrc"
bash build.sh > /dev/null
mkdir -p _obj/_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize -Wall -Wno-unused-function \
  "$@" -c "$src" -o _obj/_$name/${src%.hip}.o
objs=$(ls _obj/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj/_$name/${src%.hip}.o -o ../libselfocc_hip_$name.so
echo "built $(readlink -f ../libselfocc_hip_$name.so)"
