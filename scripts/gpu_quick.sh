# usage (on the GPU box, from the repo root): bash scripts/gpu_quick.sh <tag> [pytest -k expression]
# The inner loop of kernel work: the parity tests that pin the kernels bit for bit (ml-100k shapes + the ml-25m shape, every
# row against the oracle) and then, only if they pass, one bench line without the CPU legs.
TAG=${1:-q}
K=${2:-"ml100k or full_size or hybrid or random_small or micro"}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" > $O/pytest.log 2>&1
RC=$?
tail -4 $O/pytest.log
[ $RC -eq 0 ] || exit $RC
timeout -k 10 200 python bench.py --no-cpu-baseline --no-bf16-leg --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("ms/step %.2f  sigma %.2f  " % (d["ms_per_step"], d["step_ms"]["sigma"]), {k: round(v, 2) for k, v in d["stage_ms_per_step"].items()}, "H", d["hybrid"]["head_items"], "shortlist", round(d["shortlist_mean"], 1), "mae", d["mae"])
PY
