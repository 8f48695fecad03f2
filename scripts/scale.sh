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
done
