# usage (GPU box, repo root): bash scripts/select_phase_profile.sh <tag>
# Rebuilds the library with -DKNNCF_SELECT_PROFILE (in-kernel cycle counters of k_tail_select's phases, wave 0 of every
# workgroup) and runs one bench step: the phase split of each of the step's three launches lands in <tag>/phases.err.
TAG=${1:-prof}
O=gpurun_out/$TAG
mkdir -p $O
KNNCF_EXTRA_HIPCC_FLAGS=-DKNNCF_SELECT_PROFILE python -c "
import importlib
importlib.import_module('movie-recommender-system_amd.build').build(force=True)" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-bf16-leg --steps 1 --warmup 1 > $O/phases.json 2> $O/phases.err
grep -A11 "select profile" $O/phases.err | tail -40
