# usage (GPU box, repo root): bash scripts/gemm_variants.sh "<k list>" "<flags of variant 1>" "<flags of variant 2>" ...
# A/B of compile-time variants of the similarity GEMM alone (scripts/microbench/gemm_bench.py): rebuilds the library per flag set.
KS=$1; shift
for FLAGS in "$@"; do
  KNNCF_EXTRA_HIPCC_FLAGS="$FLAGS" python -c "
import importlib
importlib.import_module('movie-recommender-system_amd.build').build(force=True)" || exit 1
  echo "== [$FLAGS]"
  python scripts/microbench/gemm_bench.py --k $KS 2>&1 | grep -v amdgpu.ids
done
