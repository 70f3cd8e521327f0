#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash tools/profile_round.sh r02'): rocprofv3 kernel-trace stats and the PMC passes of
# the default bench command (BASELINE configs[2], the headline), of the same workload with ONE Metropolis step per launch
# (the HBM-streaming formulation), of BASELINE configs[1] (RWM, one temperature) and of the per-GPU shards of configs[3] /
# configs[4]; raw output under
# gpurun_out/prof_<tag>/; tools/profile_summary.py then writes the summaries under profiles/.  Counters are collected
# in their own runs (one TCC counter per pass), never together with API tracing; the program after `--` is python3 itself.
set -e
TAG=${1:-r03}
OUT=/root/repo/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# per-opcode issue costs on this box (tools/issue_cost.hip), the price list of tools/issue_model.py
# (raw copy under $OUT: gpurun brings gpurun_out/ home, not profiles/ - tools/profile_summary.py and tools/issue_model.py are
# then re-run at home on the raw output and write the same files under profiles/)
if [ -x /root/repo/tools/issue_cost ]; then /root/repo/tools/issue_cost > "$OUT/issue_costs.json" 2> "$OUT/issue_cost.err" && cp "$OUT/issue_costs.json" /root/repo/profiles/${TAG}_issue_costs.json; fi
# the issue time of the headline kernel's own instruction mix, replayed without dependencies (tools/mix_probe.py)
if [ -x /root/repo/tools/mix_probe ]; then timeout -k 5 120 /root/repo/tools/mix_probe > "$OUT/mix_probe.json" 2> "$OUT/mix_probe.err" || true; fi
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
# the same launch without swap events (swap_every beyond the horizon): the reference count of tools/issue_model.py, whose
# static opcode histogram of one Metropolis step must reproduce these SQ_INSTS_VALU / SQ_INSTS_SALU per wave-step
prof cfg3_noswap 10 3 --swap-every 1073741824
# instruction mix of the headline launch by class, as the hardware counts it (two more SQ passes)
for grp in "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d "$OUT/cfg3/pmc_MIX_$n" -o p -- $B --steps 4 --warmup 1 > "$OUT/cfg3.pmc_MIX_$n.log" 2>&1 || echo "(class counters $n not collected)"
done
prof cfg3_inner1 200 40 --inner 1
prof cfg2 20 4 --workload cfg2
prof cfg4 10 3 --workload cfg4 --inner 500
prof cfg5 10 3 --workload cfg5 --inner 200
cd /root/repo
timeout -k 10 500 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
echo "bench done"
python3 tools/profile_summary.py "$TAG" "$OUT"
# cost-weighted issue model of the headline kernel (recompiles one translation unit with -save-temps: ~1 min of CPU)
python3 tools/issue_model.py --from-traffic pt_d30_T32_C65536_noswap --costs profiles/${TAG}_issue_costs.json \
    --json profiles/${TAG}_issue_model.json > profiles/${TAG}_issue_model.txt 2>&1 || tail -5 profiles/${TAG}_issue_model.txt
# second bench run so that the committed line carries the counters of THIS build (traffic.json was just rewritten)
timeout -k 10 500 python3 bench.py > "$OUT/bench_n1.json" 2> "$OUT/bench_n1.err"
cp "$OUT/bench_n1.json" profiles/${TAG}_bench_n1.json
