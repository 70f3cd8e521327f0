#!/bin/bash
# wide-ladder lane-split shapes (development aid)
mkdir -p gpurun_out
run() { local name=$1 form=$2; shift 2
  PTRWM_KERNEL_FORM=$form python3 bench.py --cpu-seconds 0 --no-extras --steps 5 --warmup 2 "$@" > gpurun_out/qb_${name}_$form.json 2> gpurun_out/qb_${name}_$form.err || { tail -3 gpurun_out/qb_${name}_$form.err; return; }
  python3 -c "import json; d=json.load(open('gpurun_out/qb_${name}_$form.json')); print('$name', '$form', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
}
run d100T32 quad --dim 100 --inner 200
run d100T64 quad --dim 100 --temps 64 --chains 32768 --inner 200
run d30T32 quad --inner 500
run d30T32c1024 quad --chains 1024 --inner 5000
run d30T32c1024 thread --chains 1024 --inner 5000
run d30T64c512 quad --temps 64 --chains 512 --inner 5000
run d30T64c512 thread --temps 64 --chains 512 --inner 5000
run d30T128 quad --temps 128 --chains 8192 --inner 500
