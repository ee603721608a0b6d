#!/bin/bash
# Game.step + observation row: two launches (step, then the getter kernel) against the row written by the step kernel (pk_set_step_obs).
# usage (GPU box): tools/r06_step_obs_ab.sh [out-file]
cd "$(dirname "$0")/.."
out=${1:-gpurun_out/r06/step_obs_ab.txt}
mkdir -p "$(dirname "$out")"
: > "$out"
run() {
    echo "== $*" >> "$out"
    python3 bench.py --mode step "$@" 2>>"$out" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']
print('%-60s %8.4f G steps/s  %7.2f us/iter  hbm frac %.3f  (%d B/step alg)' % (r['kernel'], d['value']/1e9, d['kernel_ms']*1e3, r['frac'], r['algorithmic_bytes_per_step']/ (d['value']*d['kernel_ms']*1e-3) if d['value'] else 0))
" >> "$out" || return 1
}
for T in 65536 1048576; do
  S=2000; W=200; [ $T -gt 100000 ] && S=300 && W=50
  run --tables $T --players 6 --steps $S --warmup $W --step-replay &&
  run --tables $T --players 6 --steps $S --warmup $W --step-replay --step-obs packed --step-obs-separate &&
  run --tables $T --players 6 --steps $S --warmup $W --step-replay --step-obs packed &&
  run --tables $T --players 6 --steps $S --warmup $W --step-replay --step-obs dense --step-obs-separate &&
  run --tables $T --players 6 --steps $S --warmup $W --step-replay --step-obs dense || exit 1
done
run --tables 65536 --players 6 --steps 2000 --warmup 200 --step-async 1 &&
run --tables 65536 --players 6 --steps 2000 --warmup 200 --step-async 1 --step-obs packed --step-obs-separate &&
run --tables 65536 --players 6 --steps 2000 --warmup 200 --step-async 1 --step-obs packed || exit 1
cat "$out"
