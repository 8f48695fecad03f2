cd $GRAFT_REPO_ROOT
python scripts/bench_msda.py 2>&1 | grep '^{"kernel' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('swz  ', r['kernel'][:34].ljust(34), r['shape'][:22].ljust(22), r.get('ms'), r.get('GBps'))"
SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_nosw.so python scripts/bench_msda.py 2>&1 | grep '^{"kernel' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('plain', r['kernel'][:34].ljust(34), r['shape'][:22].ljust(22), r.get('ms'), r.get('GBps'))"
python scripts/bench_hotpath_eval.py 2>&1 | tail -1
SELFOCC_HIP_LIB=$GRAFT_REPO_ROOT/selfocc_amd/libselfocc_hip_nosw.so python scripts/bench_hotpath_eval.py 2>&1 | tail -1
