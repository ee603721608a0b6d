#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + separate PMC passes of ONE bench command.  Output: gpurun_out/prof_<tag>/
# usage: tools/profile_gpu.sh <tag> <steps_per_launch> [bench args...]      (bench args must set --steps / --chunk to match)
# env: PK_TABLES / PK_PLAYERS / PK_POLICY describe the workload for the summary (defaults 65536 / 6 / random)
set -u
TAG=$1; K=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd $ROOT && python3 -c "from pokerl_amd import _lib; print(_lib.source_hash())") > $OUT/lib.txt 2>/dev/null   # the kernel sources this is measured on (pk_build_info)
cd /tmp && export TMPDIR=/tmp
ARGS="--full-line --no-cpu-baseline --no-evaluator --no-extra --samples 3 --min-steps 65536 $*"
echo "{\"tables\": ${PK_TABLES:-65536}, \"players\": ${PK_PLAYERS:-6}, \"policy\": \"${PK_POLICY:-random}\", \"steps_per_launch\": $K, \"fused\": true, \"command\": \"python3 bench.py $ARGS\"}" > $OUT/workload.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
echo "profiled $TAG"
