cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/pytest_gpu5.txt 2>&1
grep -E "parity cfg2|switch|passed|failed|Error|error|FAILED" gpurun_out/pytest_gpu5.txt | head -40
timeout 900 python bench.py > gpurun_out/bench_r2b.json 2> gpurun_out/bench_r2b.err; python - <<'PY'
import json
l = json.loads(open("gpurun_out/bench_r2b.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "roofline", "parity", "gpu_torch_baseline", "cpu_baseline", "extras", "hot_path"):
    print(k, json.dumps(l.get(k))[:900])
PY
tail -3 gpurun_out/bench_r2b.err
