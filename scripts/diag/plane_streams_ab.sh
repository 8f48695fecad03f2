#!/bin/bash
# dev A/B: the three planes' inference cross-attention branches on one stream vs three (SELFOCC_PLANE_STREAMS)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for i in 1 2; do for v in 0 1; do
  SELFOCC_PLANE_STREAMS=$v python scripts/bench_hotpath_all.py --only nuscenes_depth,nuscenes_occ --no-train --iters 10 2>/dev/null > /tmp/ab_$v.json
  python - $v <<'PY'
import json, sys
j = json.loads(open(f"/tmp/ab_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print("streams =", sys.argv[1], {k: (v["eval"]["encoder_fwd"], v["eval"]["total_ms"], v["eval"]["host_enqueue_ms"]) for k, v in j.items() if isinstance(v, dict)})
PY
done; done
SELFOCC_PLANE_STREAMS=1 timeout 900 python -m pytest tests/test_golden_encoder_full_gpu.py tests/test_encoder_glue_gpu.py tests/test_shipped_configs_gpu.py -q -x 2>&1 | tail -2
