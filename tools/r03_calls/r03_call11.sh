set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_k_gputest.log 2>&1
tail -3 gpurun_out/r03_k_gputest.log
timeout 300 python tools/layer_profile.py 64 f16x3 > gpurun_out/r03_k_layers_x3.log 2>&1
head -16 gpurun_out/r03_k_layers_x3.log | tail -14; tail -1 gpurun_out/r03_k_layers_x3.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_k_bench.json 2> gpurun_out/r03_k_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_k_bench.json').read().strip().splitlines()[-1])
for k in ('value','dtype','ms_per_step','within_tolerance','max_abs_dev_yaw_pitch_clip0','verified'): print(k, d.get(k))
print('thr', {k:d['throughput_engine'].get(k) for k in ('value','within_tolerance','max_abs_dev_yaw_pitch_clip0','verified')})
print('backbone', {k:v.get('value') if isinstance(v,dict) else None for k,v in d.get('backbone',{}).items()})
print('mae', json.dumps(d.get('mae_proxy',{}).get('engines')))
print('lat', {k:v.get('ms_per_clip') if isinstance(v,dict) else v for k,v in d.get('latency_single_clip',{}).items()})
PY
