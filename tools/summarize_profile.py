#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (rocprofv3 kernel trace + separate PMC passes of ONE bench.py command, made by
tools/profile_gpu.sh) into profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.json.
usage: tools/summarize_profile.py <tag> [kernel substring]

Asynchronous rollout calls are merged on the host into launches of varying length (pk_set_coalesce), so nothing here
assumes "the launch": every pass's own bench line says how many launches and steps it made (config.launch_stats, plus the
launches before the timed region), and every figure is either a SUM over all launches of the kernel divided by those
steps (per wave-step) or by those launches (per mean launch)."""
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


def bench_line(log):
    """(launches, steps) of ALL rollout launches of the run that wrote `log` (timed region + what came before it)."""
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{") and '"launch_stats"' in line:
            r = json.loads(line)
            a, b = r["config"]["launch_stats"], r["config"]["launch_stats_before_timed_region"]
            return a["launches"] + b["launches"], a["steps"] + b["steps"], r
    return None, None, None


stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(trace)) if kname in r["Kernel_Name"]]
durs = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows)
n_l, n_s, line = bench_line(os.path.join(src, "trace.log"))
waves = int(rows[0]["Grid_Size_X"]) // int(rows[0]["Workgroup_Size_X"])
_dom = max(set(r["Kernel_Name"] for r in rows), key=lambda k: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == k))
rows = [r for r in rows if r["Kernel_Name"] == _dom] + [r for r in rows if r["Kernel_Name"] != _dom]     # registers / LDS below: of the kernel the time goes to
summary = {
    "tag": tag, "kernel": max(set(r["Kernel_Name"] for r in rows), key=lambda k: sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"] == k)),
    "kernels_matched": sorted(set(r["Kernel_Name"] for r in rows)), "launches_total": len(durs),
    "avg_launch_ms": sum(durs) / len(durs) / 1e6, "min_launch_ms": durs[0] / 1e6, "max_launch_ms": durs[-1] / 1e6,
    "median_launch_ms": durs[len(durs) // 2] / 1e6,
    "vgpr": int(rows[0]["VGPR_Count"]), "agpr": int(rows[0]["Accum_VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]),
    "lds_bytes": int(rows[0]["LDS_Block_Size"]), "scratch_bytes": int(rows[0]["Scratch_Size"]),
    "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"]), "waves_per_launch": waves,
}
wl = os.path.join(src, "workload.json")
if os.path.exists(wl):
    summary["workload"] = json.load(open(wl))
if n_l:
    assert n_l == len(durs), ("the bench line's launch count differs from the trace", n_l, len(durs))
    summary["steps_total"] = n_s
    summary["workload"]["steps_per_launch"] = n_s / float(n_l)          # mean over ALL launches of the traced run
    summary["us_per_step_of_kernel_time"] = sum(durs) / 1e3 / n_s
    summary["bench_line_of_the_traced_run"] = {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup")}
    summary["bench_line_of_the_traced_run"]["kernel"] = line["config"]["kernel"]
counters, per_step = {}, {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    pl, ps, _ = bench_line(d + ".log")
    ptr = glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))
    pass_ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ptr[0])) if kname in r["Kernel_Name"]) if ptr else 0
    agg = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(files[0])):
        if kname in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[r["Counter_Name"]] += 1
    for k, v in agg.items():
        counters[k] = v / cnt[k]                        # per mean launch of THAT pass
        if ps:
            per_step[k] = v / ps                        # per step of every table (sum over the 1 024 waves)
        if k == "GRBM_GUI_ACTIVE" and pass_ns:          # summed over the 8 XCDs (guide, DVFS): clock of that pass
            summary["effective_clock_GHz"] = v / 8.0 / pass_ns
summary["pmc_per_full_launch"] = counters
summary["pmc_per_step"] = per_step
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    # guides/MI355X_MICROARCH.md, HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes
    # of a coalesced streaming read (calibrated there for 16 B/lane; our loads are 8 and 4 B/lane -> treat as an upper
    # estimate); WRITE_SIZE reads exactly.
    summary["hbm_traffic_bytes_per_launch"] = (2.0 * counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0
    summary["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB per mean launch: FETCH_SIZE doubled per the guide's gfx950 correction"
if "SQ_INSTS_VALU" in per_step:
    summary["valu_insts_per_wave_step"] = per_step["SQ_INSTS_VALU"] / waves
    summary["salu_insts_per_wave_step"] = per_step.get("SQ_INSTS_SALU", 0) / waves
    summary["lds_insts_per_wave_step"] = per_step.get("SQ_INSTS_LDS", 0) / waves
    # chip VALU issue: wave-instructions per second vs 256 CU x 4 SIMD x 2.4 GHz / 2 cycles (guides/MI355X_MICROARCH.md)
    rate = per_step["SQ_INSTS_VALU"] * n_s / (sum(durs) * 1e-9)      # instructions per step x the traced run's steps per second of kernel time
    summary["valu_issue_rate_wave_insts_per_s"] = rate
    summary["valu_issue_frac_of_peak"] = rate / (256 * 4 * 2.4e9 / 2)
if "SQ_WAVE_CYCLES" in per_step:
    summary["wave_cycles_per_wave_step_x4"] = 4.0 * per_step["SQ_WAVE_CYCLES"] / waves      # the counter ticks in quad-cycles
if "SQ_THREAD_CYCLES_VALU" in per_step and per_step.get("SQ_ACTIVE_INST_VALU"):   # two passes: compare per STEP, not per launch
    summary["lanes_active"] = per_step["SQ_THREAD_CYCLES_VALU"] / (64.0 * per_step["SQ_ACTIVE_INST_VALU"])
if "SQ_WAVES" in counters:
    summary["waves_per_simd"] = counters["SQ_WAVES"] / 1024.0
if "SQ_ACTIVE_INST_VALU" in counters and "SQ_WAVE_CYCLES" in counters:
    summary["valu_active_frac_of_wave_cycles"] = counters["SQ_ACTIVE_INST_VALU"] / counters["SQ_WAVE_CYCLES"]
    summary["wait_any_frac_of_wave_cycles"] = counters.get("SQ_WAIT_ANY", 0) / counters["SQ_WAVE_CYCLES"]
if "TCC_HIT_sum" in counters:
    summary["l2_hit_rate"] = counters["TCC_HIT_sum"] / (counters["TCC_HIT_sum"] + counters["TCC_MISS_sum"])
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
