# two-process GPU sharing diagnostics: runtime-level twin activity (no kernels of this library) beside the msda victim
S=scripts/diag/op_replay_race.py
for mode in streams h2d malloc sync procs; do
echo "== twin: $mode"
(python scripts/diag/twin_load.py $mode 32 > /dev/null 2>&1 &)
sleep 7
DIAG_TAG=v DIAG_ONLY=msda DIAG_REPEAT=300 python $S 2>&1 | grep "^v bricks\|^v attention" | cut -c1-200
sleep 14
done
