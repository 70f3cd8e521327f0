mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo pytest rc=$rc; grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -20
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then exit $rc; fi
python tools/cliff_sweep.py > gpurun_out/r03_dim_cliff.txt 2>&1; cat gpurun_out/r03_dim_cliff.txt
