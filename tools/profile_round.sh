#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r02'): rocprofv3 kernel-trace stats and the PMC passes of
# the default bench command (BASELINE configs[2], the headline), of the same workload with ONE Metropolis step per launch
# (the HBM-streaming formulation), of BASELINE configs[1] (RWM, one temperature) and of the per-GPU shards of configs[3] /
# configs[4]; raw output under
# gpurun_out/prof_<tag>/; tools/profile_summary.py then writes the summaries under profiles/.  Counters are collected
# in their own runs (one TCC counter per pass), never together with API tracing; the program after `--` is python3 itself.
set -e
TAG=${1:-r02}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
B="python3 /root/repo/bench.py --cpu-seconds 0 --no-extras"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"
prof() {  # prof <name> <kernel-trace steps> <pmc steps> <bench args...>
  local name=$1 kts=$2 ps=$3; shift 3
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name/kt" -o kt -- $B "$@" --steps $kts --warmup 3 > "$OUT/$name.kt.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/$name/pmc_$c" -o p -- $B "$@" --steps $ps --warmup 1 > "$OUT/$name.pmc_$c.log" 2>&1
  done
  timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d "$OUT/$name/pmc_SQ" -o p -- $B "$@" --steps $ps --warmup 1 > "$OUT/$name.pmc_SQ.log" 2>&1
  echo "profiled $name"
}
prof cfg3 20 4
prof cfg3_inner1 200 40 --inner 1
prof cfg2 20 4 --workload cfg2
prof cfg4 10 3 --workload cfg4 --inner 500
prof cfg5 10 3 --workload cfg5 --inner 200
cd /root/repo
timeout -k 10 500 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench done"
python3 tools/profile_summary.py "$TAG" "$OUT"
# second bench run so that the committed line carries the counters of THIS build (traffic.json was just rewritten)
timeout -k 10 500 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cp "$OUT/bench_n1.json" profiles/${TAG}_bench_n1.json
