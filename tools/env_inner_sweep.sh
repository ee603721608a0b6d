#!/bin/bash
# GPU box: `bench.py --mode env --env-async 8` with ONE handle whose tables are split into in-handle sub-batches
# (pk_set_env_batches), over table counts x sub-batch counts.  usage: tools/env_inner_sweep.sh [out.txt]
out=${1:-gpurun_out/env_inner_sweep.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
for t in 196608 262144 393216 524288 589824 786432 1048576; do
  for ib in 1 2 3 4 5; do
    timeout -k 10 300 python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --tables $t --env-inner-batches $ib > gpurun_out/_sweep.json 2> gpurun_out/_sweep.err || exit 1
    python - "$t" "$ib" >> "$out" <<PY
import json, sys
d = json.loads(open("gpurun_out/_sweep.json").read().strip().splitlines()[-1])
print("tables %8s  sub-batches %s  %.3f G env.step/s  %.1f us per call  ready fraction %.3f" % (sys.argv[1], sys.argv[2], d["value"] / 1e9, d["ms_per_step"] * 1e3, d["ready_fraction_per_launch"]))
PY
  done
done
cat "$out"
