#!/usr/bin/env python3
"""Distils gpurun_out/prof_<tag>_case<k>/ (tools/profile_eval_hands.sh) into profiles/<tag>_summary.json: per workload case of
pk_eval_hands_d (7 distinct cards; the 2/5/6/7 mix) the kernel's launch time, VALU / LDS instructions per hand, waits, HBM traffic.
usage: tools/summarize_eval_hands.py <tag> [log2 hands = 24]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
m = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 24)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"tag": tag, "hands_per_launch": m, "cases": {}}
for case, label in ((0, "7 distinct cards (ncards = NULL: k_eval_hands_tab<false>)"), (1, "2/5/6/7 distinct cards mixed (k_eval_hands_tab<true>)")):
    src = os.path.join(root, "gpurun_out", "prof_%s_case%d" % (tag, case))
    if not os.path.isdir(src):
        continue
    rows = [r for r in csv.DictReader(open(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0])) if "k_eval_hands_tab" in r["Kernel_Name"]]
    durs = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    s = {"workload": label, "kernel": rows[0]["Kernel_Name"], "launches": len(durs), "avg_launch_ms": sum(durs) / len(durs) / 1e6, "min_launch_ms": durs[0] / 1e6,
         "vgpr": int(rows[0]["VGPR_Count"]), "sgpr": int(rows[0]["SGPR_Count"]), "lds_bytes": int(rows[0]["LDS_Block_Size"]),
         "workgroup": int(rows[0]["Workgroup_Size_X"]), "grid": int(rows[0]["Grid_Size_X"]), "unprofiled": open(os.path.join(src, "unprofiled.txt")).read().strip()}
    c = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_eval_hands_tab" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            c[k] = sum(v) / len(v)
    s["pmc_per_launch"] = c
    s["evals_per_s_by_trace"] = m / (s["avg_launch_ms"] * 1e-3)
    if "SQ_INSTS_VALU" in c:
        s["valu_insts_per_hand"] = c["SQ_INSTS_VALU"] * 64.0 / m
        s["lds_insts_per_hand"] = c.get("SQ_INSTS_LDS", 0.0) * 64.0 / m
        s["vmem_insts_per_hand"] = (c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0)) * 64.0 / m
        s["valu_issue_frac_of_peak"] = c["SQ_INSTS_VALU"] / (s["avg_launch_ms"] * 1e-3) / (256 * 4 * 2.4e9 / 2)
    if "SQ_WAVE_CYCLES" in c:
        s["wait_inst_any_frac_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        s["valu_active_frac_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
    if "SQ_THREAD_CYCLES_VALU" in c and c.get("SQ_ACTIVE_INST_VALU"):
        s["lanes_active"] = c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        s["hbm_traffic_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0   # FETCH_SIZE doubled: the guide's gfx950 correction (an upper estimate at 8 B per lane)
        alg = (7 + (1 if case else 0) + 6) * m
        s["algorithmic_bytes_per_launch"] = alg
        s["hbm_traffic_over_algorithmic"] = s["hbm_traffic_bytes_per_launch"] / alg
        s["hbm_frac_algorithmic"] = alg / (s["avg_launch_ms"] * 1e-3) / 8e12
    out["cases"][str(case)] = s
try:    # which kernel sources this was measured on (pk_build_info of the library in the tree: bench.py marks figures from another build `profile_stale`)
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
    from pokerl_amd import _lib as _pk_lib
    out["source_hash"] = _pk_lib.source_hash()
    for _d in (locals().get("src"), locals().get("d")):      # the hash the profiling script recorded ON THE BOX, if it did (lib.txt), wins
        _p = _os.path.join(_d, "lib.txt") if isinstance(_d, str) else None
        if _p and _os.path.exists(_p) and open(_p).read().strip():
            out["source_hash"] = open(_p).read().strip()
            break
except Exception as _e:   # noqa: BLE001
    out["source_hash"] = None

json.dump(out, open(os.path.join(root, "profiles", tag + "_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
