#!/bin/bash
# Build libselfocc_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: fused multiply-adds only where the source spells fmaf()
# (the arithmetic contract shared with oracle/).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libselfocc_hip.so
SRCS=$(ls *.hip)
mkdir -p _obj
pids=()
for s in $SRCS; do
  o=_obj/${s%.hip}.o
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ -n "$(find . -maxdepth 1 -name "*.h" -newer "$o")" ] || [ ../../include/selfocc_hip.h -nt "$o" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off \
      -fno-fast-math -Wall -Wno-unused-function -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
OBJS=""
for s in $SRCS; do OBJS="$OBJS _obj/${s%.hip}.o"; done      # only the objects of the sources that exist (a renamed file leaves a stale .o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $(readlink -f $OUT)"
