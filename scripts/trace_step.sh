# usage (GPU box): bash scripts/trace_step.sh <tag>   — kernel trace of one bench step, printed as a timeline of the last step
TAG=${1:-trace}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 1 --warmup 1 > $O/bench.json 2> $O/err.log
cd $R
python - <<PY
import csv, glob
f = glob.glob("$O/kt/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = after the last k_id_range launch
idx = max(i for i, r in enumerate(rows) if "k_id_range" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
out = []
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][-70:]
    out.append("%9.3f %8.3f gap %7.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, (s - prev_end) / 1e6, name))
    prev_end = max(prev_end, e)
open("$O/timeline.txt", "w").write("\n".join(out))
print("\n".join(out[:140]))
PY
