#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel trace + separate PMC passes of ONE `bench.py --mode step` command (Game.step with the
# caller's actions on device buffers: pk_pick_actions_d + pk_step_d + pk_reset_d per step).
# Output: gpurun_out/prof_<tag>/ ; afterwards, locally: tools/summarize_step_profile.py <tag>
# usage: tools/profile_step.sh <tag> [bench args after --mode step ...]     e.g. r05_step_65536x6 --steps 1000 --warmup 100
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd $ROOT && python3 -c "from pokerl_amd import _lib; print(_lib.source_hash())") > $OUT/lib.txt 2>/dev/null   # the kernel sources this is measured on (pk_build_info)
cd /tmp && export TMPDIR=/tmp
ARGS="--mode step $*"
echo "{\"command\": \"python3 bench.py $ARGS\"}" > $OUT/workload.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
echo "profiled $TAG"
