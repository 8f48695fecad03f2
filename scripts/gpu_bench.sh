#!/bin/bash
# One gpurun call: the driver's bench command, JSON line kept under gpurun_out/.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_bench.sh [tag] [extra bench.py args]'
mkdir -p gpurun_out
tag=${1:-bench}; shift || true
python bench.py --steps 20 --warmup 5 "$@" 2> gpurun_out/${tag}.err | tail -1 > gpurun_out/${tag}.json
python - <<PY
import json
d = json.load(open("gpurun_out/${tag}.json"))
keep = {k: d[k] for k in ("metric", "value", "ms_per_step", "roofline", "parity") if k in d}
keep["extras"] = d.get("extras")
keep["hot_path"] = d.get("hot_path")
print(json.dumps(keep, indent=1)[:6000])
PY
