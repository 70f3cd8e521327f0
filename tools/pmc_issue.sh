#!/bin/bash
# Where the waves of the headline kernel spend their cycles (development aid; one rocprofv3 --pmc pass, counters only):
#   tools/pmc_issue.sh [bench.py args]        -> gpurun_out/pmc_issue/, summary on stdout
# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md) and WAIT_ANY +
# WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES per wave.  The ACTIVE windows of the waves of one SIMD overlap (an
# instruction is "active" from issue to completion), so ACTIVE_INST_VALU is NOT an exclusive pipe-busy time and no roofline
# fraction is derived from it; what the split shows is where a wave's lifetime goes: parked on memory / LDS / barriers, or
# ready and waiting for the issue slot.
cd /tmp; export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc_issue
rm -rf $OUT
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT -o p -- python3 /root/repo/bench.py --cpu-seconds 0 --no-extras --steps 3 --warmup 1 --inner 500 "$@" > $OUT.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/root/repo/gpurun_out/pmc_issue/**/*counter_collection.csv', recursive=True)[-1]
acc = {}
for r in csv.DictReader(open(f)):
    if 'step_kernel' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'], {}).setdefault(r['Dispatch_Id'], 0.0)
        acc[r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
m = {k: sum(v.values()) / len(v) for k, v in acc.items()}
cyc = m['GRBM_GUI_ACTIVE'] / 8.0  # sum over the 8 XCDs -> kernel cycles
simds = 1024
print({k: '%.5g' % v for k, v in m.items()})
print('kernel cycles %.4g; waves resident per SIMD (WAVE_CYCLES*4 / SIMDs / cycles) %.2f' % (cyc, m['SQ_WAVE_CYCLES'] * 4 / simds / cyc))
print('of a wave\'s lifetime: parked (s_waitcnt / barrier) %.3f, ready but waiting to issue %.3f, instruction in flight %.3f; VALU share of the in-flight time %.3f'
      % (m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES'], m['SQ_ACTIVE_INST_ANY'] / m['SQ_WAVE_CYCLES'],
         m['SQ_ACTIVE_INST_VALU'] / m['SQ_ACTIVE_INST_ANY']))
PY
