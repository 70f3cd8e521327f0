"""Condenses the raw rocprofv3 output of tools/profile_round.sh into the summaries kept under profiles/.

    python tools/profile_summary.py r01 gpurun_out/prof_r01

Writes profiles/<tag>_kernel_stats_bench_pt_d30_T32.csv (the --stats table), <tag>_kernel_trace_ptrwm_step_kernel.csv
(per-dispatch durations of the step kernel), <tag>_pmc_ptrwm_step_kernel_cfg3.csv (per-dispatch counter means) and
updates profiles/traffic.json (HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are in KiB, FETCH doubled per the
gfx950 note in MI355X_MICROARCH.md) and profiles/<tag>_bench_n1.json."""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")
KERNEL = "ptrwm_step_kernel"


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]


# --- kernel trace ----------------------------------------------------------------------------------------------
shutil.copy(find(os.path.join(out, "kt"), "kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats_bench_pt_d30_T32.csv"))
rows = [r for r in csv.DictReader(open(find(os.path.join(out, "kt"), "kernel_trace.csv"))) if KERNEL in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
with open(os.path.join(prof, f"{tag}_kernel_trace_{KERNEL}.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --no-extras --steps 25 --warmup 3; "
            "one line per dispatch of the step kernel\n")
    f.write("dispatch,duration_ns,grid,workgroup,lds_bytes,vgprs,sgprs,scratch\n")
    for i, r in enumerate(rows):
        f.write(f"{i},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])},{r['Grid_Size_X']},{r['Workgroup_Size_X']},"
                f"{r['LDS_Block_Size']},{r['VGPR_Count']},{r['SGPR_Count']},{r['Scratch_Size']}\n")
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
timed = dur[3:] if len(dur) > 3 else dur
mean_ms = sum(timed) / len(timed) / 1e6


# --- counters --------------------------------------------------------------------------------------------------
def counters(d):
    """{counter: [value per dispatch of the step kernel]} (values summed over the XCD instances rocprofv3 lists)."""
    acc = {}
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if KERNEL not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in acc.items()}


allc = {}
for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_GRBM_GUI_ACTIVE", "pmc_SQ"):
    allc.update(counters(os.path.join(out, sub)))
with open(os.path.join(prof, f"{tag}_pmc_{KERNEL}_cfg3.csv"), "w") as f:
    f.write("# rocprofv3 --pmc <counter> --output-format csv -- python3 bench.py --cpu-seconds 0 --no-extras --steps 5 --warmup 1  "
            "(one pass per TCC counter; SQ counters in one pass)\n")
    f.write("# kernel: ptrwm_step_kernel<RoughCarpet2<30>, NormalProposal<30>, 30, exact, production>; per dispatch "
            "(65536 ladders x 32 temps x 100 steps)\n")
    f.write("counter,dispatches,mean,min,max\n")
    for k, v in allc.items():
        f.write(f"{k},{len(v)},{sum(v) / len(v):.6g},{min(v):.6g},{max(v):.6g}\n")
mean = {k: sum(v) / len(v) for k, v in allc.items()}
fetch, write = mean["FETCH_SIZE"] * 1024 * 2, mean["WRITE_SIZE"] * 1024
tj_path = os.path.join(prof, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
# GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of one XCD = sum / 8
clock = mean["GRBM_GUI_ACTIVE"] / 8 / (mean_ms * 1e-3) / 1e9
tj["pt_d30_T32_C65536_inner100"] = {
    "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
    "source": f"profiles/{tag}_pmc_{KERNEL}_cfg3.csv (FETCH_SIZE and WRITE_SIZE in KiB, separate --pmc passes; FETCH "
              "doubled per the gfx950 note in MI355X_MICROARCH.md)",
    "valu_insts_per_launch": mean["SQ_INSTS_VALU"], "salu_insts_per_launch": mean["SQ_INSTS_SALU"],
    "grbm_gui_active_per_launch_sum_over_8_xcd": mean["GRBM_GUI_ACTIVE"], "profiled_kernel_ms": mean_ms,
    "shader_clock_ghz": clock,
}
json.dump(tj, open(tj_path, "w"), indent=1)
shutil.copy(os.path.join(out, "bench_n1.json"), os.path.join(prof, f"{tag}_bench_n1.json"))
print(f"step kernel: {len(rows)} dispatches, mean of the timed ones {mean_ms:.4f} ms; HBM {(fetch + write) / 1e6:.1f} MB/launch; "
      f"VALU {mean['SQ_INSTS_VALU']:.4g} wave-insts/launch; clock {clock:.3f} GHz")
