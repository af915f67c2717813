# usage: bash scripts/sweep_flags.sh "<flags list>" "<head list>"  — bench sweeps over engine flags / head widths
mkdir -p gpurun_out
for F in $1; do for H in $2; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --head-items $H --engine-flags $F 2>gpurun_out/sf_${F}_$H.err > gpurun_out/sf_${F}_$H.json
  python -c "
import json; d=json.load(open('gpurun_out/sf_${F}_$H.json')); s=d['stage_ms_per_step']; ra=d['roofline_all']
print('flags',$F,'H',$H, round(d['value']), round(d['ms_per_step'],1), d['mae'], {k:round(v,1) for k,v in s.items()}, d['hybrid']['head_items'], round(d['shortlist_mean'],1), 'launches', ra['k_gemm_nt_bf16']['launches_per_step'], 'gemmTF', round(ra['k_gemm_nt_bf16']['executed_tflops']), 'dominant', d['roofline']['kernel'].split()[0], round(d['roofline']['frac'],3))"
done; done
