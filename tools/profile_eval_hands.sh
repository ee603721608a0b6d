#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + separate PMC passes of pk_eval_hands_d (tools/eval_hands_bench.py), one workload case at a time.
# Output: gpurun_out/prof_<tag>_case<k>/ ; then locally: tools/summarize_eval_hands.py <tag>
set -u
TAG=$1; LOG2=${2:-24}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for CASE in 0 1; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_case$CASE
  rm -rf $OUT; mkdir -p $OUT
(cd $ROOT && python3 -c "from pokerl_amd import _lib; print(_lib.source_hash())") > $OUT/lib.txt 2>/dev/null   # the kernel sources this is measured on (pk_build_info)
  export PK_EHB_CASE=$CASE
  python3 $ROOT/tools/eval_hands_bench.py $LOG2 > $OUT/unprofiled.txt 2>/dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/eval_hands_bench.py $LOG2 > $OUT/trace.log 2>&1 || echo "trace failed"
  for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT"; do
    N=$(echo $C | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/tools/eval_hands_bench.py $LOG2 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
  done
done
echo "profiled $TAG"
