#!/bin/bash
# Runs ON THE GPU BOX: one SQ counter pass (8 counters) + one kernel trace of a short bench run; prints per-wave-step figures.
# usage: tools/quick_pmc.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/qpmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2048 --warmup 512 --chunk 2048 --samples 2 --min-steps 2048 --no-cpu-baseline --no-evaluator $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc -- python3 $ROOT/bench.py $ARGS > $OUT/pmc.log 2>&1 || echo "pmc failed"
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
fs = glob.glob(out + "/pmc/*/*_counter_collection.csv")
if not fs:
    sys.exit("pmc failed: no counter_collection CSV under %s/pmc (rejected counter set? see %s/pmc.log)" % (out, out))
f = fs[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_rollout" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {}
for k, v in agg.items():
    v = sorted(v); full = [x for x in v if x > 0.5 * v[-1]]
    c[k] = sum(full) / len(full)
K = 2048
w = c["SQ_WAVES"]
print("per wave-step: VALU %.0f  SALU %.0f  LDS %.1f  wave-cycles %.0f (x4 = %.0f clk)" % (
    c["SQ_INSTS_VALU"] / w / K, c["SQ_INSTS_SALU"] / w / K, c["SQ_INSTS_LDS"] / w / K, c["SQ_WAVE_CYCLES"] / w / K, 4 * c["SQ_WAVE_CYCLES"] / w / K))
print("lanes_active %.3f  valu_active_frac %.3f  wait_any_frac %.3f" % (
    c["SQ_THREAD_CYCLES_VALU"] / (64 * c["SQ_ACTIVE_INST_VALU"]), c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
