# usage (GPU box, repo root): bash scripts/phase_profile.sh <tag> <KNNCF_SELECT_PROFILE|KNNCF_RERANK_PROFILE|KNNCF_PREDICT_PROFILE> [extra hipcc flags]
# Rebuilds the library with the kernel's in-kernel cycle counters (thread 0 of every workgroup adds the cycles of each phase to a
# global table, dumped after the launch) and runs two bench steps; the profiled build is slower than the product.
TAG=${1:-phase}
O=gpurun_out/$TAG
mkdir -p $O
KNNCF_EXTRA_HIPCC_FLAGS="-D$2 $3" python -c "
import importlib
importlib.import_module('movie-recommender-system_amd.build').build(force=True)" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-bf16-leg --steps 2 --warmup 1 > $O/phases.json 2> $O/phases.err
grep -i "profile\]" $O/phases.err | tail -4
