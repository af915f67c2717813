mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -k "not full_size" 2>&1 | tail -4 || exit 1
for H in 0 256 512 1024 1856; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --head-items $H 2>gpurun_out/sw_$H.err > gpurun_out/sw_$H.json
  python -c "
import json; d=json.load(open('gpurun_out/sw_$H.json')); s=d['stage_ms_per_step']; print($H, round(d['value']), round(d['ms_per_step'],1), d['mae'], {k:round(v,1) for k,v in s.items()}, d['hybrid'], round(d['shortlist_mean'],1))"
done
