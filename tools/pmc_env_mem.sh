#!/bin/bash
# Runs ON THE GPU BOX: memory-side counter passes (FETCH_SIZE, WRITE_SIZE, TA / TCP busy -- each in its own pass) of the
# asynchronous env bench; prints per-launch figures of k_env_step_async.  usage: tools/pmc_env_mem.sh <tag> [bench args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_env_mem_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--mode env --steps 600 --warmup 100 --env-async 8 $*"
for pass in "FETCH_SIZE" "WRITE_SIZE" "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  d=$OUT/$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $d -- python3 $ROOT/bench.py $ARGS > $d.log 2>&1 || echo "pass failed: $pass"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
dur = []
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_env_step_async" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(out + "/FETCH_SIZE/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_env_step_async" in r["Kernel_Name"]:
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k in sorted(agg):
    v = agg[k]
    print("%-32s mean per launch %.4g (n=%d)" % (k, sum(v) / len(v), len(v)))
if dur:
    print("kernel duration under the profiler: mean %.1f us (n=%d)" % (sum(dur) / len(dur) / 1e3, len(dur)))
PY
