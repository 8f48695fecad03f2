#!/bin/bash
# Turnkey scaling run on ONE node with >= 8 GPUs (the driver's SCALE tier; never run on the 1-GPU gpurun boxes):
#   bash scripts/scale.sh [steps] [warmup]   -> gpurun_out/scale_<mode>_n<N>.json, one bench.py line each
# frames = one frame per rank (weak scaling, the default bench line); rays = ONE frame split into row blocks (strong).
# bench.py asserts backend == nccl (RCCL) and distinct_devices == N on every N > 1 line.
STEPS=${1:-200}; WARM=${2:-10}; R=$(cd "$(dirname "$0")/.." && pwd); mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in frames rays; do
  for n in 1 2 4 8; do
    out=$R/gpurun_out/scale_${mode}_n${n}.json
    if [ $n -eq 1 ]; then
      python $R/bench.py --gpus 1 --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras > $out
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) \
        $R/bench.py --gpus $n --steps $STEPS --warmup $WARM --shard $mode --no-cpu-baseline --no-extras > $out
    fi
    python - "$out" "$mode" "$n" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
j = json.loads(l[-1])
print(sys.argv[2], "n =", sys.argv[3], "rays/s", j["value"], "ms/step", j["ms_per_step"], "backend", j["ranks_seen"]["backend"],
      "devices", j["ranks_seen"]["distinct_devices"])
PY
  done
  # ONE SCALE-shaped JSON line per mode (what the driver's SCALE_rNN.json holds: per-N values; efficiency is the reader's to compute)
  python - "$R" "$mode" <<'PY'
import json, sys
R, mode = sys.argv[1], sys.argv[2]
per_n = {}
for n in (1, 2, 4, 8):
    try:
        l = [x for x in open(f"{R}/gpurun_out/scale_{mode}_n{n}.json") if x.startswith("{")]
        j = json.loads(l[-1])
        per_n[str(n)] = {"value": j["value"], "unit": j["unit"], "ms_per_step": j["ms_per_step"], "scaling": j["scaling"],
                         "backend": j["ranks_seen"]["backend"], "distinct_devices": j["ranks_seen"]["distinct_devices"]}
    except Exception as e:
        per_n[str(n)] = {"error": repr(e)[:120]}
line = {"metric": "rendered rays/sec (6-cam 450x800, 128 samples/ray)", "mode": mode, "per_n": per_n}
open(f"{R}/gpurun_out/SCALE_{mode}.json", "w").write(json.dumps(line) + "\n")
print(json.dumps(line))
PY
done
