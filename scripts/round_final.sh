# usage (GPU box, repo root): bash scripts/round_final.sh <tag>
# The round's closing measurement set in ONE call: phase profiles of k_tail_select / k_rerank / k_predict_knn_items (profiling
# builds), then the default build's bench + rocprofv3 kernel summary + PMC passes (profile_round.sh), the 8-shard rehearsal and
# a fuzz parity sweep.
TAG=${1:-rXX}
O=gpurun_out/$TAG
mkdir -p $O
for K in SELECT RERANK PREDICT; do
  bash scripts/phase_profile.sh ${TAG}_phase_$K KNNCF_${K}_PROFILE > $O/phases_$K.txt 2>&1
  grep -A11 -i "profile\]" gpurun_out/${TAG}_phase_$K/phases.err | tail -14 >> $O/phases_$K.txt
  tail -12 $O/phases_$K.txt
done
python -c "
import importlib
importlib.import_module('movie-recommender-system_amd.build').build(force=True)" || exit 1
bash scripts/profile_round.sh $TAG || exit 1
python scripts/shard_rehearsal.py --out $O/shard8_timings.json > $O/shard8.log 2>&1; tail -10 $O/shard8.log
timeout -k 10 400 python scripts/fuzz_parity.py 1000 250 1 8 > $O/fuzz_a.log 2>&1; tail -3 $O/fuzz_a.log
echo round_final done
