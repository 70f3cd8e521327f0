#!/bin/bash
# SQ counters of the step kernel for the shipped library and for alternative builds (development aid)
#   tools/pmc_ab.sh <name>=<lib> ...
cd /tmp; export TMPDIR=/tmp
for spec in "cur=" "$@"; do
  name=${spec%%=*}; lib=${spec#*=}
  export PTRWM_LIB=$lib
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d /root/repo/gpurun_out/pmcab_$name -o p -- python3 /root/repo/bench.py --cpu-seconds 0 --no-extras --steps 3 --warmup 1 --inner 500 ${AB_ARGS} > /root/repo/gpurun_out/pmcab_$name.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob('/root/repo/gpurun_out/pmcab_$name/**/*counter_collection.csv',recursive=True)[-1]
acc={}
for r in csv.DictReader(open(f)):
    if 'step_kernel' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'],{}).setdefault(r['Dispatch_Id'],0.0)
        acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
print('$name', {k: '%.5g' % (sum(v.values())/len(v)) for k,v in acc.items()})
PY
done
