#!/bin/bash
# One gpurun call: GPU test suite (or a -k selection), logs under gpurun_out/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_tests.sh [pytest args]'
mkdir -p gpurun_out
python -m pytest ${@:-tests} -m gpu -q -x 2>&1 | tail -40 | tee gpurun_out/gpu_tests.log
