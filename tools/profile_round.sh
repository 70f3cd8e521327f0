#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r01'): rocprofv3 kernel-trace stats and the PMC passes
# of the default bench command, raw output under gpurun_out/prof_<tag>/; tools/profile_summary.py then writes the
# summaries under profiles/.  Counters are collected in their own runs (one TCC counter per pass), never together
# with API tracing.
set -e
TAG=${1:-r01}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
BENCH="python3 /root/repo/bench.py --cpu-seconds 0 --no-extras"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- $BENCH --steps 25 --warmup 3 > "$OUT/kt.log" 2>&1
echo "kernel trace done"
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -o p -- $BENCH --steps 5 --warmup 1 > "$OUT/pmc_$c.log" 2>&1
  echo "pmc $c done"
done
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_SQ" -o p -- $BENCH --steps 5 --warmup 1 > "$OUT/pmc_SQ.log" 2>&1
echo "pmc SQ done"
cd /root/repo
timeout -k 10 400 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench done"
python3 tools/profile_summary.py "$TAG" "$OUT"
