"""Condenses the raw rocprofv3 output of tools/profile_round.sh into the summaries kept under profiles/.

    python tools/profile_summary.py r02 gpurun_out/prof_r02

For each profiled configuration <cfg> in {cfg3 (BASELINE configs[2], default steps per launch), cfg3_inner1 (one
Metropolis step per launch), cfg2 (BASELINE configs[1])} writes profiles/<tag>_kernel_stats_<cfg>.csv (the --stats
table), <tag>_kernel_trace_<cfg>.csv (per-dispatch durations of the step kernel), <tag>_pmc_<cfg>.csv (per-dispatch
counter means) and updates profiles/traffic.json (HBM bytes per launch: FETCH_SIZE and WRITE_SIZE are in KiB, FETCH
doubled per the gfx950 note in MI355X_MICROARCH.md; VALU instructions per launch; shader clock from GRBM_GUI_ACTIVE / 8
XCDs / kernel time; the sha256 of the library the counters belong to)."""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
prof = os.path.join(ROOT, "profiles")
KERNEL = "step_kernel"  # ptrwm_step_kernel (one thread per replica) or ptrwm_quad_step_kernel (lane-split)
LIB = os.path.join(ROOT, "rwm-pt-pytorch_amd", "lib", "libptrwm_hip.so")
lib_sha = hashlib.sha256(open(LIB, "rb").read()).hexdigest()
# cfg -> (traffic.json key, Metropolis steps per launch, description, warm-up launches in the kernel-trace run)
CFGS = {
    "cfg3": ("pt_d30_T32_C65536", None, "BASELINE configs[2]: 65536 ladders x 32 temps", 3),
    "cfg3_noswap": ("pt_d30_T32_C65536_noswap", None, "BASELINE configs[2] with swap_every beyond the horizon (no swap events)", 3),
    "cfg3_inner1": ("pt_d30_T32_C65536_inner1", 1, "BASELINE configs[2], ONE Metropolis step per launch", 3),
    "cfg2": ("rwm_d30_T1_C65536", None, "BASELINE configs[1]: 65536 chains x 1 temperature", 3),
    "cfg4": ("cfg4", 500, "BASELINE configs[3] per-GPU shard: EvenRosenbrock d30, Laplace, 32 temps, 65536 ladders, 500 steps per launch", 3),
    "cfg5": ("cfg5", 200, "BASELINE configs[4] per-GPU shard: ThreeMixture d50, UniformRadius, 64 temps, 131072 ladders, 200 steps per launch", 3),
}


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", f"*{suffix}"), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[-1]


def counters(d):
    """{counter: [value per dispatch of the step kernel]} (values summed over the XCD instances rocprofv3 lists)."""
    acc = {}
    for r in csv.DictReader(open(find(d, "counter_collection.csv"))):
        if KERNEL not in r["Kernel_Name"]:
            continue
        acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
        acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: list(v.values()) for k, v in acc.items()}


tj_path = os.path.join(prof, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
for cfg, (key, inner, desc, warm) in CFGS.items():
    base = os.path.join(out, cfg)
    if not os.path.isdir(base):
        print(f"(no raw output for {cfg})")
        continue
    shutil.copy(find(os.path.join(base, "kt"), "kernel_stats.csv"), os.path.join(prof, f"{tag}_kernel_stats_{cfg}.csv"))
    rows = [r for r in csv.DictReader(open(find(os.path.join(base, "kt"), "kernel_trace.csv"))) if KERNEL in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    kname = rows[0]["Kernel_Name"] if rows else "?"
    with open(os.path.join(prof, f"{tag}_kernel_trace_{cfg}.csv"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --no-extras ... ({desc}); one line per "
                f"dispatch of the step kernel, the first {warm} are warm-up\n# kernel: {kname}\n")
        f.write("dispatch,duration_ns,grid,workgroup,lds_bytes,vgprs,sgprs,scratch\n")
        for i, r in enumerate(rows):
            f.write(f"{i},{int(r['End_Timestamp']) - int(r['Start_Timestamp'])},{r['Grid_Size_X']},{r['Workgroup_Size_X']},"
                    f"{r['LDS_Block_Size']},{r['VGPR_Count']},{r['SGPR_Count']},{r['Scratch_Size']}\n")
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    timed = dur[warm:] if len(dur) > warm else dur
    mean_ms = sum(timed) / len(timed) / 1e6
    allc = {}
    for sub in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE", "pmc_GRBM_GUI_ACTIVE", "pmc_SQ"):
        allc.update(counters(os.path.join(base, sub)))
    for sub in sorted(glob.glob(os.path.join(base, "pmc_MIX_*"))):  # instruction-class counters (headline launch only)
        try:
            allc.update(counters(sub))
        except SystemExit:
            pass
    with open(os.path.join(prof, f"{tag}_pmc_{cfg}.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc <counter> --output-format csv -- python3 bench.py --cpu-seconds 0 --no-extras ... ({desc}); "
                "one pass per TCC counter, SQ counters in one pass; per dispatch of the step kernel\n")
        f.write(f"# kernel: {kname}\n# library sha256: {lib_sha}\n")
        f.write("counter,dispatches,mean,min,max\n")
        for k, v in allc.items():
            f.write(f"{k},{len(v)},{sum(v) / len(v):.6g},{min(v):.6g},{max(v):.6g}\n")
    mean = {k: sum(v) / len(v) for k, v in allc.items()}
    fetch, write = mean["FETCH_SIZE"] * 1024 * 2, mean["WRITE_SIZE"] * 1024
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles of one XCD = sum / 8 (reads high on sub-0.3 ms dispatches)
    clock = mean["GRBM_GUI_ACTIVE"] / 8 / (mean_ms * 1e-3) / 1e9
    # Metropolis steps per launch of the profiled command: stated for inner1, else read from the bench line
    steps_per_launch = inner
    if steps_per_launch is None:
        try:
            steps_per_launch = json.load(open(os.path.join(out, "bench_n1.json")))["config"]["mh_steps_per_launch"]
        except Exception:
            steps_per_launch = None
    tj[key] = {
        "hbm_bytes_per_launch": fetch + write, "fetch_bytes_corrected_x2": fetch, "write_bytes": write,
        "source": f"profiles/{tag}_pmc_{cfg}.csv (FETCH_SIZE and WRITE_SIZE in KiB, separate --pmc passes; FETCH doubled per "
                  "the gfx950 note in MI355X_MICROARCH.md)",
        "valu_insts_per_launch": mean["SQ_INSTS_VALU"], "salu_insts_per_launch": mean["SQ_INSTS_SALU"],
        "mh_steps_per_launch": steps_per_launch,
        "grbm_gui_active_per_launch_sum_over_8_xcd": mean["GRBM_GUI_ACTIVE"], "profiled_kernel_ms": mean_ms,
        "shader_clock_ghz": clock, "lib_sha256": lib_sha, "kernel": kname,
    }
    mix = {k: v for k, v in mean.items() if k.startswith("SQ_INSTS_VALU_") or k == "SQ_INSTS_LDS"}
    if mix:
        tj[key]["instruction_classes_per_launch"] = mix
    if "SQ_WAVES" in mean:
        tj[key]["waves_per_launch"] = mean["SQ_WAVES"]
    print(f"{cfg}: {len(rows)} dispatches, mean of the timed ones {mean_ms:.4f} ms; HBM {(fetch + write) / 1e6:.1f} MB/launch "
          f"({(fetch + write) / (mean_ms * 1e-3) / 1e12:.2f} TB/s); VALU {mean['SQ_INSTS_VALU']:.4g} wave-insts/launch; "
          f"clock {clock:.3f} GHz")
json.dump(tj, open(tj_path, "w"), indent=1)
