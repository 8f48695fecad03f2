#!/bin/bash
# Per-kernel register / LDS / occupancy table of one csrc/*.hip file (compiler's view, gfx950).
# usage: scripts/kernel_resources.sh selfocc_amd/csrc/render_fwd.hip [name-filter]
f=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize -c "$f" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk '
  /remark: Function Name:/ {name=$NF}
  /remark: +TotalSGPRs:/ {sg=$NF}
  /remark: +VGPRs:/ {v=$NF}
  /remark: +AGPRs:/ {ag=$NF}
  /remark: +ScratchSize/ {sc=$NF}
  /remark: +Occupancy/ {occ=$NF}
  /remark: +LDS Size/ {lds=$NF; printf "%s vgpr=%s agpr=%s sgpr=%s scratch=%s occ=%s lds=%s\n", name, v, ag, sg, sc, occ, lds}' | c++filt | grep -E "$filt"
