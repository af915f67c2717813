# usage (GPU box, repo root): bash scripts/round_final.sh <tag>
# The round's closing measurement set in ONE call: phase profiles of k_rerank / k_tail_select (profiling builds), then the default
# build's bench + rocprofv3 kernel summary + PMC passes (profile_round.sh), the 8-shard rehearsal and a fuzz parity sweep.
TAG=${1:-rXX}
O=gpurun_out/$TAG
mkdir -p $O
KNNCF_EXTRA_HIPCC_FLAGS=-DKNNCF_RERANK_PROFILE python -c "
import importlib
importlib.import_module('movie-recommender-system_amd.build').build(force=True)" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-bf16-leg --steps 1 --warmup 1 > $O/rerank_phases.json 2> $O/rerank_phases.err
grep "rerank profile" $O/rerank_phases.err | tail -3
bash scripts/select_phase_profile.sh $TAG > $O/select_phases.txt 2>&1
tail -14 $O/select_phases.txt
python -c "
import importlib
importlib.import_module('movie-recommender-system_amd.build').build(force=True)" || exit 1
bash scripts/profile_round.sh $TAG || exit 1
python scripts/shard_rehearsal.py --out $O/shard8_timings.json > $O/shard8.log 2>&1; tail -10 $O/shard8.log
timeout -k 10 400 python scripts/fuzz_parity.py 1000 250 1 8 > $O/fuzz_a.log 2>&1; tail -3 $O/fuzz_a.log
echo round_final done
