#!/bin/bash
mkdir -p gpurun_out
run() { local name=$1 form=$2; shift 2
  PTRWM_KERNEL_FORM=$form python3 bench.py --cpu-seconds 0 --no-extras --steps 4 --warmup 2 "$@" > gpurun_out/qb_${name}_$form.json 2> gpurun_out/qb_${name}_$form.err || { tail -3 gpurun_out/qb_${name}_$form.err; return; }
  python3 -c "import json; d=json.load(open('gpurun_out/qb_${name}_$form.json')); print('$name', '$form', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
}
run d100T17 quad --dim 100 --temps 17 --inner 200
run d100T20 quad --dim 100 --temps 20 --inner 200
run d100T24 quad --dim 100 --temps 24 --inner 200
run d100T32 quad --dim 100 --temps 32 --inner 200
run d100T40 quad --dim 100 --temps 40 --chains 32768 --inner 200
run d30T20 quad --temps 20 --inner 500
run d30T20 thread --temps 20 --inner 500
