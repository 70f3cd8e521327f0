#!/bin/bash
# A/B on ONE box: the shipped library against alternative builds (PTRWM_LIB), alternating, bench.py main reading only.
#   tools/ab_bench.sh <name>=<lib path> ...    (development aid; output under gpurun_out/ab_*.json)
mkdir -p gpurun_out
for rep in 1 2; do
  for spec in "cur=" "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    PTRWM_LIB=$lib python3 bench.py --cpu-seconds 0 --no-extras --steps 10 --warmup 2 ${AB_ARGS} > gpurun_out/ab_${name}_$rep.json 2> gpurun_out/ab_${name}_$rep.err || exit 1
    python3 -c "import json,sys; d=json.load(open('gpurun_out/ab_${name}_$rep.json')); print('$name', $rep, '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
  done
done
