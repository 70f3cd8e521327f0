#!/bin/bash
# long randomised runs for the record (development aid): kernel vs oracle (proven flips), lane-split vs thread form,
# split steps vs fused kernel.  Stops at the first failing tool.
mkdir -p gpurun_out
O=${BIGFUZZ_OFFSET:-0}   # added to every seed: a different offset = a different set of cases
for seed in $((101+O)) $((102+O)) $((103+O)) $((104+O)); do
  timeout -k 10 600 python3 tools/fuzz_vs_oracle.py 500 $seed > gpurun_out/bigfuzz_oracle_$seed.log 2>&1 || { echo "fuzz_vs_oracle seed $seed FAILED"; tail -5 gpurun_out/bigfuzz_oracle_$seed.log; exit 1; }
  tail -1 gpurun_out/bigfuzz_oracle_$seed.log
done
for seed in $((201+O)) $((202+O)); do
  timeout -k 10 600 python3 tools/check_quad.py 1000 $seed > gpurun_out/bigfuzz_quad_$seed.log 2>&1 || { echo "check_quad seed $seed FAILED"; tail -5 gpurun_out/bigfuzz_quad_$seed.log; exit 1; }
  tail -1 gpurun_out/bigfuzz_quad_$seed.log
done
timeout -k 10 600 python3 tools/fuzz_split.py 600 $((301+O)) > gpurun_out/bigfuzz_split.log 2>&1 || { echo "fuzz_split FAILED"; tail -5 gpurun_out/bigfuzz_split.log; exit 1; }
tail -1 gpurun_out/bigfuzz_split.log
