cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 2> gpurun_out/r03_zz_bench.err | tail -1 > gpurun_out/r03_zz_bench.json
python -c "
import json; d=json.loads(open('gpurun_out/r03_zz_bench.json').read())
print(d['value'], d['ms_per_step'], d['verified'], d['within_tolerance'], d['roofline']['frac'], d['host_input']['f32']['value'], d['latency_single_clip']['f16x3']['ms_per_clip'], d['cpu_baseline']['value'])"
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r03_w_gputests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r03_w_gputests.log | tail -3
