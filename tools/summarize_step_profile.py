#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (rocprofv3 kernel trace + separate PMC passes of ONE `bench.py --mode step` command, made by
tools/profile_step.sh) into profiles/<tag>_kernel_stats.csv + profiles/<tag>_summary.json: per kernel of the Game.step loop
(k_pick, k_step, k_reset) the average launch duration and the per-launch counters, and per LOOP ITERATION (one launch of each) the HBM
traffic by the guide's recipe (separate --pmc passes; FETCH_SIZE doubled on gfx950) -- bench.py's Game.step legs quote
`hbm_traffic_bytes_per_step` as `roofline.traffic`.     usage: tools/summarize_step_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
KERNELS = ("k_pick<", "k_step<", "k_step_async<", "k_reset<", "k_obs_packed(", "k_obs(")


def bench_line(log):
    for line in reversed(open(log).read().splitlines()):
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


def short(name):
    for k in KERNELS:
        if k in name:
            return k[:-1]
    return None


line = bench_line(os.path.join(src, "trace.log"))
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
trace = glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]
per = collections.defaultdict(list)
first = {}
for r in csv.DictReader(open(trace)):
    k = short(r["Kernel_Name"])
    if k:
        per[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        first.setdefault(k, r)
m = re.search(r"(\d+) x (\d+)", line["name"])
if "ready_fraction_per_launch" in line:
    pass
obs = "packed" if "packed observation rows" in line["name"] else "dense" if "dense observation rows" in line["name"] else None
summary = {"tag": tag, "workload": {"tables": int(m.group(1)), "players": int(m.group(2)), "replay": "replayed" in line["name"], "bounded": "bounded launches" in line["name"],
                                    "obs": obs, "obs_fused": bool(obs) and "from the step kernel" in line["name"],
                                    "command": json.load(open(os.path.join(src, "workload.json")))["command"]},
           "bench_line_of_the_traced_run": {k: line[k] for k in ("value", "kernel_ms", "launches", "device_ms", "ms_per_step", "ready_fraction_per_launch") if k in line},
           "kernels": {}}
for k, d in per.items():
    d.sort()
    r = first[k]
    summary["kernels"][k] = {"launches": len(d), "avg_ms": sum(d) / len(d) / 1e6, "median_ms": d[len(d) // 2] / 1e6, "min_ms": d[0] / 1e6, "max_ms": d[-1] / 1e6,
                             "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                             "lds_bytes": int(r["LDS_Block_Size"]), "scratch_bytes": int(r["Scratch_Size"]),
                             "grid": int(r["Grid_Size_X"]), "workgroup": int(r["Workgroup_Size_X"])}
summary["k_step_avg_ms"] = sum(summary["kernels"][k]["avg_ms"] for k in summary["kernels"] if k.startswith("k_step"))
summary["k_step_min_median_max_ms"] = [[summary["kernels"][k][x] for x in ("min_ms", "median_ms", "max_ms")] for k in summary["kernels"] if k.startswith("k_step")][0]
summary["kernel_ms_per_step_sum"] = sum(v["avg_ms"] for v in summary["kernels"].values() if v["launches"] * 2 >= max(x["launches"] for x in summary["kernels"].values()))
# PMC passes: per kernel, the mean per launch; a loop iteration = one launch of each kernel of the loop
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        if k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        summary["kernels"].setdefault(k, {}).setdefault("pmc_per_launch", {})[c] = v / cnt[(k, c)]
tot = 0.0
most = max(v.get("launches", 0) for v in summary["kernels"].values())
summary["loop_kernels"] = [k for k, v in summary["kernels"].items() if v.get("launches", 0) * 2 >= most]   # (not the one k_reset of the set-up)
for k, v in summary["kernels"].items():
    p = v.get("pmc_per_launch", {})
    if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
        v["hbm_traffic_bytes_per_launch"] = (2.0 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024.0
        if k in summary["loop_kernels"]:
            tot += v["hbm_traffic_bytes_per_launch"]
    if p.get("SQ_ACTIVE_INST_VALU") and "SQ_THREAD_CYCLES_VALU" in p:
        v["lanes_active"] = p["SQ_THREAD_CYCLES_VALU"] / (64.0 * p["SQ_ACTIVE_INST_VALU"])
    if p.get("SQ_WAVES"):
        v["valu_insts_per_wave"] = p.get("SQ_INSTS_VALU", 0.0) / p["SQ_WAVES"]
summary["hbm_traffic_bytes_per_step"] = tot
summary["hbm_traffic_note"] = "(2*FETCH_SIZE + WRITE_SIZE) KiB per mean launch, summed over the kernels of one loop iteration: FETCH_SIZE doubled per the guide's gfx950 correction (an upper estimate)"
T, N = summary["workload"]["tables"], summary["workload"]["players"]
row_bytes = 0 if obs is None else (16 + 8 * (3 * N + 1)) if obs == "packed" else 8 * (17 + 3 * N)
summary["algorithmic_bytes_per_step"] = (2 * (35 * N + 21) + 16 + row_bytes) * T
summary["traffic_over_algorithmic"] = tot / summary["algorithmic_bytes_per_step"] if tot else None
if summary["k_step_avg_ms"]:
    summary["k_step_hbm_frac_algorithmic"] = summary["algorithmic_bytes_per_step"] / (summary["k_step_avg_ms"] * 1e-3) / 8e12
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
