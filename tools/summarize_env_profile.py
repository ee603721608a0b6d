#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (rocprofv3 kernel trace + separate PMC passes of ONE `bench.py --mode env` command, made by
tools/profile_env.sh) into profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.json: per MEAN LAUNCH of the PokerGameEnv
kernel (k_env_step / k_env_step_async) -- VALU / SALU / LDS wave instructions, lanes active, waits, HBM traffic by the guide's
recipe (separate --pmc passes; FETCH_SIZE doubled on gfx950) -- which bench.py's env legs multiply by their own launch counts
and divide by their own HIP-event time.     usage: tools/summarize_env_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")


def bench_line(log):
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


line = bench_line(os.path.join(src, "trace.log"))
kname = line["roofline"]["kernel"] + "<"
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(trace)) if kname in r["Kernel_Name"]]
durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows)
waves = int(rows[0]["Grid_Size_X"]) // int(rows[0]["Workgroup_Size_X"])
vgpr = int(rows[0]["VGPR_Count"])
cfg = line["config"]
import re
m = re.match(r"(\d+) batch\(es\) x (\d+) tables x (\d+) seats", cfg["workload"])
am = re.search(r"bounded launches of (\d+) betting passes", line["metric"])
summary = {
    "tag": tag, "kernel": rows[0]["Kernel_Name"], "launches_total": len(durs),
    "avg_launch_ms": sum(durs) / len(durs) / 1e6, "min_launch_ms": durs[0] / 1e6, "max_launch_ms": durs[-1] / 1e6,
    "median_launch_ms": durs[len(durs) // 2] / 1e6,
    "vgpr": vgpr, "agpr": int(rows[0]["Accum_VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]),
    "lds_bytes": int(rows[0]["LDS_Block_Size"]), "scratch_bytes": int(rows[0]["Scratch_Size"]),
    "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"]), "waves_per_launch": waves,
    "workload": {"tables": int(m.group(2)), "players": int(m.group(3)), "env_batches": int(m.group(1)),
                 "env_inner_batches": cfg.get("env_inner_batches", 1), "env_async": int(am.group(1)) if am else 0,
                 "command": json.load(open(os.path.join(src, "workload.json")))["command"]},
    "bench_line_of_the_traced_run": {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup", "game_steps_per_env_step",
                                                          "ready_fraction_per_launch", "device_ms", "launches")},
}
# resident waves per SIMD: what one launch brings, capped by the kernel's registers (rocprofv3's VGPR_Count is HALF the allocation: 88 for 174)
summary["waves_per_simd_launched"] = waves / 1024.0
summary["waves_per_simd_resident"] = min(waves / 1024.0, float(max(1, min(8, 512 // max(1, 2 * vgpr)))))
counters = {}
# The counters of one dispatch accumulate over rocprofv3's bracket around it -- the dispatch plus the packets that start and stop the
# counters -- which is a few microseconds LONGER than the kernel's own start/end timestamps (round 4 divided GRBM_GUI_ACTIVE by the sum of
# those timestamps and got an "effective clock" of 2.46 / 2.87 / 2.74 GHz: above the 2.4 GHz the chip runs at, by the share of the bracket
# in a 30 us launch).  So: (1) show that the PMC passes SERIALISE the dispatches (no launch of the pass starts before the previous one has
# ended: a dispatch's counters hold that dispatch's events only, whatever the un-profiled run overlaps -- instruction and byte counts per
# launch are exact and need no window at all); (2) take the clock over the pass's own accumulation window, first start to last end of
# its kernels: busy cycles / elapsed time, <= the true clock by construction.
serial = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    ptr = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))
    krows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(ptr[0])) if kname in r["Kernel_Name"])) if ptr else []
    pass_ns = sum(e - b for b, e in krows)
    span_ns = (krows[-1][1] - krows[0][0]) if krows else 0
    overlaps = sum(1 for i in range(1, len(krows)) if krows[i][0] < krows[i - 1][1])
    serial[os.path.basename(d)] = {"launches": len(krows), "overlapping_launches": overlaps, "kernel_ns_sum": pass_ns, "span_ns": span_ns}
    agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(files[0])):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    for k, v in agg.items():
        counters[k] = v / cnt[k]                        # per mean launch of THAT pass
        if k == "GRBM_GUI_ACTIVE" and span_ns:          # summed over the 8 XCDs
            summary["effective_clock_GHz"] = v / 8.0 / span_ns
            summary["effective_clock_note"] = ("GRBM_GUI_ACTIVE / 8 XCDs / (first start .. last end of the pass's kernels): busy cycles over the pass's own "
                                               "accumulation window; per sum of kernel timestamps it would read %.2f GHz (the counter bracket of a dispatch is "
                                               "longer than the kernel)" % (v / 8.0 / max(1, pass_ns)))
            summary["counter_bracket_us_per_launch"] = (v / 8.0 / 2.4 - pass_ns) / max(1, len(krows)) / 1e3
summary["pmc_passes"] = serial
summary["pmc_dispatches_serialised"] = all(v["overlapping_launches"] == 0 for v in serial.values())
summary["pmc_per_launch"] = counters
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    summary["hbm_traffic_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
    summary["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB per mean launch: FETCH_SIZE doubled per the guide's gfx950 correction"
    summary["hbm_frac_of_peak_in_the_traced_run"] = summary["hbm_traffic_bytes_per_launch"] * line["launches"] / (line["device_ms"] * 1e-3) / 8e12
if "SQ_INSTS_VALU" in counters:
    summary["valu_wave_insts_per_launch"] = counters["SQ_INSTS_VALU"]
    summary["valu_insts_per_wave"] = counters["SQ_INSTS_VALU"] / waves
    summary["salu_insts_per_wave"] = counters.get("SQ_INSTS_SALU", 0) / waves
    summary["lds_insts_per_wave"] = counters.get("SQ_INSTS_LDS", 0) / waves
    summary["valu_issue_frac_of_peak_in_the_traced_run"] = counters["SQ_INSTS_VALU"] * line["launches"] / (line["device_ms"] * 1e-3) / (256 * 4 * 2.4e9 / 2)
if "SQ_THREAD_CYCLES_VALU" in counters and counters.get("SQ_ACTIVE_INST_VALU"):
    summary["lanes_active"] = counters["SQ_THREAD_CYCLES_VALU"] / (64.0 * counters["SQ_ACTIVE_INST_VALU"])
if "SQ_ACTIVE_INST_VALU" in counters and "SQ_WAVE_CYCLES" in counters:
    summary["valu_active_frac_of_wave_cycles"] = counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_WAVE_CYCLES"]
    summary["wait_any_frac_of_wave_cycles"] = counters.get("SQ_WAIT_ANY", 0) / counters["SQ_WAVE_CYCLES"]
try:    # which kernel sources this was measured on (pk_build_info of the library in the tree: bench.py marks figures from another build `profile_stale`)
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    from pokerl_amd import _lib as _pk_lib
    summary["source_hash"] = _pk_lib.source_hash()
    for _d in (locals().get("src"), locals().get("d")):      # the hash the profiling script recorded ON THE BOX, if it did (lib.txt), wins
        _p = _os.path.join(_d, "lib.txt") if isinstance(_d, str) else None
        if _p and _os.path.exists(_p) and open(_p).read().strip():
            summary["source_hash"] = open(_p).read().strip()
            break
except Exception as _e:   # noqa: BLE001
    summary["source_hash"] = None

json.dump(summary, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
