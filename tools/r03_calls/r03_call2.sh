set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03_b_gputest.log 2>&1
tail -5 gpurun_out/r03_b_gputest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err
tail -c 600 gpurun_out/r03_b_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_b_bench.json').read().strip().splitlines()[-1])
for k in ('value','dtype','ms_per_step','within_tolerance','max_abs_dev_yaw_pitch_clip0','verified'): print(k, d.get(k))
print('thr', {k:d['throughput_engine'].get(k) for k in ('value','within_tolerance','max_abs_dev_yaw_pitch_clip0','verified')})
print('backbone', d.get('backbone'))
print('mae', json.dumps(d.get('mae_proxy',{}).get('engines')), d.get('mae_proxy',{}).get('oracle_synthetic_gt_deg'), d.get('mae_proxy',{}).get('oracle_cpu_seconds'))
print('roof', {k:d['roofline'][k] for k in ('achieved','frac','traffic','algorithmic_bytes_per_launch','kernel')})
PY
MCG_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --latency 0 --mae-videos 0 --backbone-clips 0 --second-engine none 2>gpurun_out/r03_b_dist.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dist', d.get('world_size'), d.get('rccl_ranks_verified'), d['value'])"
