#!/bin/bash
# Build libselfocc_hip.so for gfx950 (cross-compiles without a GPU).
# -ffp-contract=off: fused multiply-adds only where the source spells fmaf()
# (the arithmetic contract shared with oracle/).
# -fno-slp-vectorize -fno-vectorize: no compiler-formed packed-FP32 VALU instructions (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32).  Measured
# on MI355X (round 5, profiles/r5_b_packed_fp32_mfma.txt): a wave executing them beside a wave that runs
# v_mfma_f32_16x16x32_bf16 on the same SIMD sporadically gets a wrong low half (seen as lx * W = 0 in the bilinear setup of
# the MSDA gathers: a few wrong (query, head) rows per launch) — whenever a bf16-MFMA kernel of this library shares the GPU
# with another of its kernels (a second stream or a second process; the removed msda_pro_fwd had both phases inside one block).
# Without the two vectorizers the library contains no v_pk_* instruction and the effect is gone.
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
      -fno-fast-math -fno-slp-vectorize -fno-vectorize -Wall -Wno-unused-function -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
OBJS=""
for s in $SRCS; do OBJS="$OBJS _obj/${s%.hip}.o"; done      # only the objects of the sources that exist (a renamed file leaves a stale .o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o "$OUT"
echo "built $(readlink -f $OUT)"
