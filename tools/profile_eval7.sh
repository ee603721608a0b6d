#!/bin/bash
# Runs ON THE GPU BOX: kernel trace + separate PMC passes of the stand-alone evaluator leg (tools/eval7_bench.py).
# Output: gpurun_out/prof_<tag>/ ; then locally: tools/summarize_eval7.py <tag>
set -u
TAG=$1; LOG2=${2:-28}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
(cd $ROOT && python3 -c "from pokerl_amd import _lib; print(_lib.source_hash())") > $OUT/lib.txt 2>/dev/null   # the kernel sources this is measured on (pk_build_info)
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/eval7_bench.py $LOG2 20 > $OUT/unprofiled.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/tools/eval7_bench.py $LOG2 20 > $OUT/trace.log 2>&1 || echo "trace failed"
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -- python3 $ROOT/tools/eval7_bench.py $LOG2 20 > $OUT/pmc_$N.log 2>&1 || echo "pmc $C failed"
done
echo "profiled $TAG"
