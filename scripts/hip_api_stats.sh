# usage (GPU box, repo root): bash scripts/hip_api_stats.sh <tag>
# rocprofv3 HIP-API summary of a 5-step bench run (no counters): which host-side calls the step spends its time in
TAG=${1:-api}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $O/prof -o api -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 5 --warmup 2 > $O/bench.json 2> $O/err
cd $R
ls $O/prof | head
python - <<PY
import csv, glob
for f in glob.glob("$O/prof/*hip_api_stats.csv") + glob.glob("$O/prof/*hip_stats.csv"):
    rows = list(csv.reader(open(f)))
    print(f)
    for r in rows[:16]:
        print(r[:6])
PY
rm -f $O/prof/*_trace.csv
