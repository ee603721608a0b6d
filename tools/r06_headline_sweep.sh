#!/bin/bash
# Headline kernel A/B on ONE box (round 6): dev-6 libraries built by pokerl_amd.build.build_dev_lib -- carried high bet (-DPK_CARRY_HB), betting
# passes per look at the parked lanes (PK_BET_PASSES) with their cursor_tail masks (PK_TAIL_MASK) -- each over the parking thresholds (PK_PARK).
# usage (GPU box): tools/r06_headline_sweep.sh [out]
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/r06/headline_sweep.txt}
mkdir -p "$(dirname "$out")"; : > "$out"
for rep in 1 2; do
for v in base hb p3 p5 p5b p6 p4t p4c; do
  for park in 24 28 32 36 40; do
    if [ $rep = 2 ] && [ $park != 28 ] && [ $park != 32 ]; then continue; fi
    echo "rep $rep park $park: $(POKERL_HIP_LIB=$PWD/pokerl_amd/libpokerl_hip_dev6_$v.so PK_PARK=$park python3 tools/variant_bench.py 6 0 2>&1 | tail -1)" >> "$out"
  done
done
done
cat "$out"
