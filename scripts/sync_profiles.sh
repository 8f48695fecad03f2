#!/bin/bash
# After `gpurun ... scripts/gpu_final_r6.sh <tag>`: copy what that session merged into gpurun_out/ to profiles/ under the names
# the docs and bench.py cite (the box's own profiles/ does not travel back).   usage: scripts/sync_profiles.sh <tag>
T=${1:?tag}; R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for f in gpurun_out/${T}_*; do
  b=$(basename $f)
  case $b in
    *.err|*.log) ;;
    ${T}_pmc_msda.json) cp $f profiles/pmc_msda.json ;;
    ${T}_pmc_bwd.json) cp $f profiles/pmc_bwd.json ;;
    ${T}_pmc_traffic.json) cp $f profiles/pmc_traffic.json ;;
    ${T}_c1_pmc.txt|${T}_c4_pmc.txt|${T}_c25_pmc.txt) cp $f profiles/${T}_render_${b#${T}_} ;;
    *) cp $f profiles/$b ;;
  esac
done
# the parity logs the GPU suite of the same call wrote
for k in parity_full_frame head_parity encoder_full_parity render_bwd_parity train_step_parity train_steps_parity; do
  [ -f gpurun_out/$k.jsonl ] && cp gpurun_out/$k.jsonl profiles/${T}_$k.jsonl
done
ls profiles | grep "^$T" | wc -l
