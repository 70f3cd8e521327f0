mkdir -p gpurun_out
for w in cfg5 cfg4 cfg2; do python bench.py --workload $w --cpu-seconds 0 --no-extras --steps 10 --warmup 3 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; python -c "
import json; d=json.load(open('gpurun_out/bench_$w.json')); print('$w', '%.4g'%d['value'], d['roofline'].get('kernel_ms'), d['config']['workload'][:60])"; done
python bench.py --cpu-seconds 0 --no-extras --steps 10 --warmup 3 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err; python -c "
import json; d=json.load(open('gpurun_out/bench_cfg3.json')); print('cfg3', '%.4g'%d['value'], d['roofline'].get('kernel_ms'))"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo pytest rc=$rc; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
timeout -k 10 900 python tools/check_all_variants.py > gpurun_out/check_all_variants.txt 2>&1; echo cav rc=$?; tail -5 gpurun_out/check_all_variants.txt
