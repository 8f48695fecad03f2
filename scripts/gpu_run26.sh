cd $GRAFT_REPO_ROOT
for i in 1 2; do python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
python - <<'PY'
import os, sys, subprocess
PY
sed -i 's/^FUSED_WGRAD = True/FUSED_WGRAD = False/' selfocc_amd/model/bricks.py
for i in 1 2; do python scripts/bench_hotpath_train.py 2>&1 | tail -1; done
