mkdir -p gpurun_out
P=$PWD/rwm-pt-pytorch_amd/lib_prev/libptrwm_hip.so
for rep in 1 2; do for w in cfg4 cfg5 cfg3 cfg2; do for lib in cur prev; do
  if [ $lib = prev ]; then export PTRWM_LIB=$P; else unset PTRWM_LIB; fi
  python bench.py --workload $w --cpu-seconds 0 --no-extras --steps 8 --warmup 2 > gpurun_out/ab_${w}_${lib}_$rep.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_${w}_${lib}_$rep.json')); print('$w $lib $rep', '%.4g'%d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
done; done; done
unset PTRWM_LIB
tools/issue_cost > gpurun_out/r03_issue_costs.json 2> gpurun_out/issue_cost.err; head -c 1500 gpurun_out/r03_issue_costs.json
