# usage (GPU box, repo root): bash scripts/kstats.sh <tag> [grep pattern]
# rocprofv3 kernel summary of a 3-step bench run; prints the rows matching the pattern (default: the prep kernels)
TAG=${1:-ks}
PAT=${2:-"k_user_hash_order|k_item|k_pack|k_make_keys|k_unpack|k_gather|k_invert|k_ordered_fold|onesweep|k_dev|k_pre|k_mark|k_dense|k_table|k_resolve|k_raw|k_segment|k_col_keys|k_id_range|k_check|k_exact"}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o ks -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 3 --warmup 1 > $O/bench.json 2> $O/err
cd $R
python - <<PY
import csv, re
rows = list(csv.reader(open("$O/prof/ks_kernel_stats.csv")))
tot = 0
for r in rows[1:]:
    if re.search(r"$PAT", r[0]):
        per_step = float(r[2]) / 1e6 / 4
        tot += per_step
        print("%-90s calls %5s  avg %8.3f ms  per step %7.3f ms" % (r[0][:90], r[1], float(r[3]) / 1e6, per_step))
print("sum per step %.3f ms" % tot)
PY
rm -f $O/prof/*_kernel_trace.csv
