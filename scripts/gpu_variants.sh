# usage (GPU box, repo root): bash scripts/gpu_variants.sh <tag> "<flags of variant 1>" "<flags of variant 2>" ...
# A/B of compile-time variants in one call: for each flag set rebuilds the library and runs one bench line without the CPU
# legs (the first variant should be "" = the default build); prints one summary line per variant.
TAG=$1; shift
i=0
for FLAGS in "$@"; do
  bash scripts/gpu_variant.sh ${TAG}_v$i "$FLAGS" || exit 1
  i=$((i+1))
done
