#!/bin/bash
# Runs ON THE GPU BOX: one SQ counter pass of the asynchronous env bench; prints per-launch figures of k_env_step_async.
# usage: tools/quick_pmc_env.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/qpmc_env_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--mode env --steps 1000 --warmup 100 --env-async 8 $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
fs = glob.glob(out + "/pmc/*/*_counter_collection.csv")
if not fs:
    sys.exit("pmc failed: see %s/pmc.log" % out)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "k_env_step_async" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
line = json.loads([l for l in open(out + "/pmc.log") if l.startswith("{")][-1])
w = c["SQ_WAVES"]
print("launches %d; per wave and launch: VALU %.0f  SALU %.0f  LDS %.0f  wave-cycles x4 %.0f" % (
    len(agg["SQ_WAVES"]), c["SQ_INSTS_VALU"] / w, c["SQ_INSTS_SALU"] / w, c["SQ_INSTS_LDS"] / w, 4 * c["SQ_WAVE_CYCLES"] / w))
print("lanes_active %.3f  valu_active_frac %.3f  wait_any_frac %.3f" % (
    c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]), c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
print("bench: %.3f G env.step/s, %.2f Game.steps per env.step, ready fraction %.3f" % (line["value"] / 1e9, line["game_steps_per_env_step"], line["ready_fraction_per_launch"]))
PY
