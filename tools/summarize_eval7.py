#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>/ (made by tools/profile_eval7.sh) into profiles/<tag>_kernel_stats.csv + _summary.json.
usage: tools/summarize_eval7.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
kname = "k_eval7_"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
shutil.copy(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0], os.path.join(dst, tag + "_kernel_stats.csv"))
rows = [r for r in csv.DictReader(open(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0]))
        if kname in r["Kernel_Name"] and "stream" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
run = json.load(open(os.path.join(src, "unprofiled.json")))
all_durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
skip = int(run.get("warm_dispatches", 0)) + 1          # tools/eval7_bench.py's warm-up call and the untimed first dispatch of the timed call: the clock ramps
rows = rows[skip:] if len(rows) > skip + 2 else rows
durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows]
m = run["hands"]
s = {"tag": tag, "kernel": rows[0]["Kernel_Name"], "hands_per_launch": m, "launches": len(durs),
     "avg_launch_ms": sum(durs) / len(durs) / 1e6, "min_launch_ms": min(durs) / 1e6,
     "launch_ms_of_every_dispatch_traced": [d / 1e6 for d in all_durs], "warm_dispatches_skipped": skip if len(durs) != len(all_durs) else 0,
     "vgpr": int(rows[0]["VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]), "lds_bytes": int(rows[0]["LDS_Block_Size"]),
     "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"]), "unprofiled_run": run}
c = {}
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"] and "stream" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        v = v[skip:] if len(v) > skip + 2 else v          # (the CSV is in dispatch order: the same warm-up dispatches as in the trace are left out)
        c[k] = sum(v) / len(v)
s["pmc_per_launch"] = c
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    # guide (HBM): KiB units; FETCH_SIZE reports half of a 16 B/lane streaming read on gfx950 (our loads ARE 16 B/lane), WRITE_SIZE exact
    s["hbm_traffic_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
    s["hbm_traffic_over_algorithmic"] = s["hbm_traffic_bytes_per_launch"] / (12.0 * m)
    s["hbm_GBps_from_traffic"] = s["hbm_traffic_bytes_per_launch"] / (s["avg_launch_ms"] * 1e-3) / 1e9
s["algorithmic_GBps"] = 12.0 * m / (s["avg_launch_ms"] * 1e-3) / 1e9
s["hbm_frac_of_8TBps"] = s["algorithmic_GBps"] / 8000.0
if "SQ_INSTS_VALU" in c:
    s["valu_wave_insts_per_eval_x64"] = c["SQ_INSTS_VALU"] / (m / 64.0)      # wave-instructions per 64 evaluations = per-lane instructions per evaluation
    s["salu_wave_insts_per_eval_x64"] = c.get("SQ_INSTS_SALU", 0) / (m / 64.0)
    s["lds_wave_insts_per_eval_x64"] = c.get("SQ_INSTS_LDS", 0) / (m / 64.0)
    rate = c["SQ_INSTS_VALU"] / (s["avg_launch_ms"] * 1e-3)
    s["valu_issue_rate_wave_insts_per_s"] = rate
    s["valu_issue_frac_of_peak"] = rate / (256 * 4 * 2.4e9 / 2)
if c.get("SQ_WAVE_CYCLES"):
    s["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0) / c["SQ_WAVE_CYCLES"]
    s["wait_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
    s["wait_inst_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
if c.get("SQ_WAVES"):
    s["waves_per_launch"] = c["SQ_WAVES"]
if c.get("GRBM_GUI_ACTIVE"):
    s["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (s["avg_launch_ms"] * 1e-3) / 1e9
try:    # which kernel sources this was measured on (pk_build_info of the library in the tree: bench.py marks figures from another build `profile_stale`)
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    from pokerl_amd import _lib as _pk_lib
    s["source_hash"] = _pk_lib.source_hash()
    for _d in (locals().get("src"), locals().get("d")):      # the hash the profiling script recorded ON THE BOX, if it did (lib.txt), wins
        _p = _os.path.join(_d, "lib.txt") if isinstance(_d, str) else None
        if _p and _os.path.exists(_p) and open(_p).read().strip():
            s["source_hash"] = open(_p).read().strip()
            break
except Exception as _e:   # noqa: BLE001
    s["source_hash"] = None

json.dump(s, open(os.path.join(dst, tag + "_summary.json"), "w"), indent=1)
print(json.dumps(s, indent=1))
