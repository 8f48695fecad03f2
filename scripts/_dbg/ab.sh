#!/bin/bash
# A/B of kernel builds: scripts/_dbg/ab.sh <grep-pattern> lib_a.so lib_b.so ...   (run on the GPU box)
PAT=$1; shift
for l in "$@"; do echo "== $l"; SELFOCC_HIP_LIB=$PWD/scripts/_dbg/$l python scripts/time_render.py cfg2 2>&1 | grep -E "$PAT" ; done
