mkdir -p gpurun_out
X=$PWD/rwm-pt-pytorch_amd/lib_exp/libptrwm_hip.so
for rep in 1 2; do for lib in cur exp; do
  if [ $lib = exp ]; then export PTRWM_LIB=$X; else unset PTRWM_LIB; fi
  python bench.py --workload cfg2 --cpu-seconds 0 --no-extras --steps 8 --warmup 2 > gpurun_out/ab_cfg2_${lib}_$rep.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; }
  python -c "
import json; d=json.load(open('gpurun_out/ab_cfg2_${lib}_$rep.json')); print('cfg2 $lib $rep', '%.4g'%d['value'], '%.3f ms' % d['roofline']['kernel_ms'])"
done; done
unset PTRWM_LIB
python tools/form_sweep.py heldout > gpurun_out/r03_form_heldout.txt 2>&1; tail -2 gpurun_out/r03_form_heldout.txt
