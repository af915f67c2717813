# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh <tag>
# Writes gpurun_out/<tag>/: the default bench line, the rocprofv3 kernel summary of a 5-step run and the three PMC passes
# (one counter set per run, --pmc with --kernel-trace only) that scripts/pmc_traffic.py turns into the traffic JSON.
set -e
TAG=${1:-rXX}
R=$(pwd)
O=$R/gpurun_out/$TAG
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $TAG -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 5 --warmup 2 > $O/bench_prof.json 2> $O/prof.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 1 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 1 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq -o s -- python3 $R/bench.py --no-cpu-baseline --no-bf16-leg --steps 1 --warmup 1 > $O/pmc_sq.json 2> $O/pmc_sq.err
cd $R
python scripts/pmc_traffic.py $O/pmc_fetch/f_counter_collection.csv $O/pmc_write/w_counter_collection.csv $O/pmc_sq/s_counter_collection.csv $O/pmc_traffic.json $O/pmc_fetch.json
rm -f $O/prof/*_kernel_trace.csv $O/pmc_*/?_kernel_trace.csv
echo done $TAG
