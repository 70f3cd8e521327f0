mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo pytest rc=$rc; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
python bench.py --cpu-seconds 0 > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_full.json'))
print('value %.4g' % d['value'], 'kernel_ms', d['roofline']['kernel_ms'])
i1 = d['roofline']['hbm_stream_inner1']; print('inner1', i1['kernel_ms_mean'], i1['kernel_ms_median'], i1['frac'])
for k, v in d['other_single_gpu_readings'].items(): print('%.4g' % v['value'], k[:70])
PY
(cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > /root/repo/gpurun_out/rocprof_counters.txt 2>&1); grep -c . gpurun_out/rocprof_counters.txt; grep -o "SQ_INSTS_VALU[A-Z0-9_]*" gpurun_out/rocprof_counters.txt | sort -u | tr '\n' ' '
python tools/form_sweep.py dense > gpurun_out/r03_form_dense.txt 2>&1; tail -3 gpurun_out/r03_form_dense.txt
