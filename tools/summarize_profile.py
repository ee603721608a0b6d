#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (rocprofv3 kernel trace + separate PMC passes, made by tools/profile_gpu.sh) into
profiles/<tag>_*.  usage: tools/summarize_profile.py <tag> [kernel substring]"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
kname = sys.argv[2] if len(sys.argv) > 2 else "k_rollout"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(trace)) if kname in r["Kernel_Name"]]
durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows)
# the launches of the profiled length: with few long launches they are the longest ones; with hundreds of short ones
# (--steps 20) they are the bulk, and the longest are the cold first launch / the flushes of deferred work
med = durs[len(durs) // 2]
big = [d for d in durs if 0.6 * med <= d <= 1.6 * med] if len(durs) >= 50 else [d for d in durs if d > 0.5 * durs[-1]]
lo, hi = big[0], big[-1]
summary = {
    "tag": tag, "kernel": rows[0]["Kernel_Name"], "launches_total": len(durs), "launches_full": len(big),
    "avg_full_launch_ms": sum(big) / len(big) / 1e6, "min_full_launch_ms": big[0] / 1e6, "max_full_launch_ms": big[-1] / 1e6,
    "vgpr": int(rows[0]["VGPR_Count"]), "agpr": int(rows[0]["Accum_VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]),
    "lds_bytes": int(rows[0]["LDS_Block_Size"]), "scratch_bytes": int(rows[0]["Scratch_Size"]),
    "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"]),
}
# workload of the profiled command (tools/profile_gpu.sh passes it through PK_PROFILE_WORKLOAD)
wl = os.path.join(src, "workload.json")
if os.path.exists(wl):
    summary["workload"] = json.load(open(wl))
counters = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = sorted(v)
        if k in ("FETCH_SIZE", "WRITE_SIZE"):           # one read + one write of the table state whatever the launch
            full = [v[len(v) // 2]]                     # length: the median (a pass now and then reports a 4x outlier)
        elif len(v) >= 50:                                # many short launches: the bulk around the median
            m = v[len(v) // 2]
            full = [x for x in v if 0.5 * m <= x <= 2.0 * m] or v
        else:
            full = [x for x in v if x > 0.5 * v[-1]]
        counters[k] = sum(full) / len(full)             # per launch of the profiled length
summary["pmc_per_full_launch"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    # guides/MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes
    # of a coalesced streaming read (calibrated there for 16 B/lane; our loads are 8 and 4 B/lane -> treat as an upper
    # estimate); WRITE_SIZE reads exactly.
    summary["hbm_traffic_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
    summary["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE doubled per the guide's gfx950 correction"
if "SQ_INSTS_VALU" in counters and "SQ_WAVES" in counters:
    summary["valu_insts_per_wave"] = counters["SQ_INSTS_VALU"] / counters["SQ_WAVES"]
    summary["salu_insts_per_wave"] = counters.get("SQ_INSTS_SALU", 0) / counters["SQ_WAVES"]
    k = summary.get("workload", {}).get("steps_per_launch")
    if k:
        summary["valu_insts_per_wave_step"] = summary["valu_insts_per_wave"] / k
        summary["salu_insts_per_wave_step"] = summary["salu_insts_per_wave"] / k
        # chip VALU issue: wave-instructions per second vs 256 CU x 4 SIMD x 2.4 GHz / 2 cycles (guides/MI355X_MICROARCH.md)
        rate = counters["SQ_INSTS_VALU"] / (summary["avg_full_launch_ms"] * 1e-3)
        summary["valu_issue_rate_wave_insts_per_s"] = rate
        summary["valu_issue_frac_of_peak"] = rate / (256 * 4 * 2.4e9 / 2)
if "SQ_THREAD_CYCLES_VALU" in counters and counters.get("SQ_ACTIVE_INST_VALU"):
    summary["lanes_active"] = counters["SQ_THREAD_CYCLES_VALU"] / (64.0 * counters["SQ_ACTIVE_INST_VALU"])
if "SQ_WAVES" in counters:
    summary["waves_per_simd"] = counters["SQ_WAVES"] / 1024.0
if "SQ_ACTIVE_INST_VALU" in counters and "SQ_WAVE_CYCLES" in counters:
    summary["valu_active_frac_of_wave_cycles"] = counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_WAVE_CYCLES"]
    summary["wait_any_frac_of_wave_cycles"] = counters.get("SQ_WAIT_ANY", 0) / counters["SQ_WAVE_CYCLES"]
if "TCC_HIT_sum" in counters:
    summary["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
