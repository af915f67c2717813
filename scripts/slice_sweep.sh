# usage (GPU box, repo root): bash scripts/slice_sweep.sh [values...]
# how many of a shard's heaviest rows to re-rank as slices (KNNCF_DEBUG_SLICE_ROWS): shard_rehearsal.py, ranks 0 and 7 of 8
mkdir -p gpurun_out/slice_sweep
for n in ${@:-0 40 80 160 320 640}; do
  KNNCF_DEBUG_SLICE_ROWS=$n timeout -k 10 200 python scripts/shard_rehearsal.py --world 8 --ranks 0,7 --passes 3 --out gpurun_out/slice_sweep/shard_$n.json > gpurun_out/slice_sweep/shard_$n.log 2>&1 || exit 1
  echo "slice rows $n"; grep "^rank" gpurun_out/slice_sweep/shard_$n.log
done
