#!/bin/bash
# Prefix of every gpurun call of round 6: one row of the per-box packed-FP32 survey, then the command.
#   gpurun --timeout N -- 'bash scripts/gpu_session.sh <command ...>'
mkdir -p gpurun_out
SURVEY_SECONDS=${SURVEY_SECONDS:-3} timeout 120 python scripts/pk_swizzle_survey.py > gpurun_out/pk_swizzle_last.json 2> gpurun_out/pk_swizzle_last.err || echo "survey failed" >&2
eval "$@"
