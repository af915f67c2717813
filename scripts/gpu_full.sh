# usage (GPU box, repo root): bash scripts/gpu_full.sh <tag>
# The whole GPU suite, then (only if it passes) one bench line without the CPU legs; prints the 8-shard rehearsal record.
TAG=${1:-full}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
RC=$?
tail -5 $O/pytest.log
[ $RC -eq 0 ] || exit $RC
timeout -k 10 200 python bench.py --no-cpu-baseline --no-bf16-leg --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step %.2f  sigma %.2f  " % (d["ms_per_step"], d["step_ms"]["sigma"]), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "mae", d["mae"])
PY
[ -f gpurun_out/shard8_timings.json ] && cat gpurun_out/shard8_timings.json
exit 0
