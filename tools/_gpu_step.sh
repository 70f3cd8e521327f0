mkdir -p gpurun_out
tools/issue_cost > gpurun_out/r03_issue_costs.json 2> gpurun_out/issue_cost.err; python -c "
import json; d=json.load(open('gpurun_out/r03_issue_costs.json'))['waves_per_simd_4']; print({k:v for k,v in d.items() if k.startswith('mix') or k in ('v_cndmask_b32','v_cmp_lt_f32','v_fma_f32','v_exp_f32','v_mad_u64_u32','v_bitop3_b32')})"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo pytest rc=$rc; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -20
if [ $rc -ne 0 ]; then tail -40 gpurun_out/pytest_gpu.log; exit $rc; fi
python bench.py --cpu-seconds 0 --steps 6 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_full.json'))
print('value %.4g' % d['value'], d['roofline']['kernel_ms'])
for k, v in d['other_single_gpu_readings'].items(): print('%.4g' % v['value'], k[:70])
PY
