mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo pytest rc=$rc; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
P=$PWD/rwm-pt-pytorch_amd/lib_prev/libptrwm_hip.so
for w in cfg4 cfg5 cfg3 cfg2; do 
  python bench.py --workload $w --cpu-seconds 0 --no-extras --steps 8 --warmup 2 > gpurun_out/ab_${w}_cur.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_${w}_cur.json')); print('$w cur', '%.4g'%d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
done
python bench.py --cpu-seconds 0 --steps 5 --warmup 2 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_full.json'))
i1 = d['roofline']['hbm_stream_inner1']; print('inner1', i1['kernel_ms_mean'], i1['kernel_ms_median'], i1['frac'])
for k, v in d['other_single_gpu_readings'].items(): print('%.4g' % v['value'], k[:70])
PY
