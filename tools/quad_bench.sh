#!/bin/bash
# thread vs quad form on ONE box for several workloads (development aid)
mkdir -p gpurun_out
run() { # name form args...
  local name=$1 form=$2; shift 2
  PTRWM_KERNEL_FORM=$form python3 bench.py --cpu-seconds 0 --no-extras --steps 6 --warmup 2 "$@" > gpurun_out/qb_${name}_$form.json 2> gpurun_out/qb_${name}_$form.err || { tail -3 gpurun_out/qb_${name}_$form.err; return; }
  python3 -c "import json; d=json.load(open('gpurun_out/qb_${name}_$form.json')); print('$name', '$form', '%.4g' % d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
}
for form in thread quad auto; do
  run cfg2 $form --workload cfg2 --inner 2000
  run cfg3 $form --inner 500
  run cfg4 $form --workload cfg4 --inner 500
  run cfg5 $form --workload cfg5 --inner 200
  run d100 $form --dim 100 --inner 200
  run d50 $form --dim 50 --inner 300
  run c4096 $form --chains 4096 --inner 2000
  run onelad $form --chains 1 --temps 8 --inner 20000
done
