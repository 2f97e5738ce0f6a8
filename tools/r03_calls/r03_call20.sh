cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
PRECISION=f16x3 bash tools/round_profile.sh r03_p_x3 > gpurun_out/r03_p_x3_round.log 2>&1
PRECISION=f16x3 bash tools/pmc_bench_mfma.sh > gpurun_out/r03_p_x3_mfma_util.md 2>gpurun_out/r03_p_x3_mfma.err
cat gpurun_out/r03_p_x3_mfma_util.md | head -12
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_p_x3_bench.json').read().strip().splitlines()[-1])
for k in ('value','dtype','ms_per_step','within_tolerance','max_abs_dev_yaw_pitch_clip0','verified'): print(k, d.get(k))
r=d['roofline']; print({k:r[k] for k in ('achieved','frac','traffic','algorithmic_bytes_per_launch','traffic_over_algorithmic','kernel','launches_per_step')}); print(r.get('hbm_step'))
print({k:(v['launches'],v['ms'],v['tflops']) for k,v in r['all_contraction_launches'].items()})
print('thr', {k:d['throughput_engine'].get(k) for k in ('value','ms_per_step','within_tolerance')})
print('backbone', {k:v.get('value') if isinstance(v,dict) else None for k,v in d.get('backbone',{}).items()})
print('lat', {k:v.get('ms_per_clip') if isinstance(v,dict) else v for k,v in d.get('latency_single_clip',{}).items()})
PY
bash tools/decoder_prof.sh f16x3 2>&1 | grep "decoder only"
