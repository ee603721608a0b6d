mkdir -p gpurun_out/r05
for lib in "" pokerl_amd/libpokerl_hip_dev6.so; do
  if [ -n "$lib" ]; then export POKERL_HIP_LIB=$PWD/$lib; tag="PK_NO_LONE (before)"; else unset POKERL_HIP_LIB; tag="lone-table paths (after)"; fi
  echo "== $tag"
  for rep in 1 2; do
  python bench.py --mode step --steps 2000 --warmup 200 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  Game.step loop          %.3f G steps/s  %.2f us/step' % (r['value']/1e9, r['kernel_ms']*1e3))"
  python bench.py --mode step --steps 2000 --warmup 200 --step-replay 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  Game.step replay        %.3f G steps/s  %.2f us/step' % (r['value']/1e9, r['kernel_ms']*1e3))"
  python bench.py --mode env --steps 300 --warmup 30 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  env.step sync           %.4f G env.step/s  %.1f us/launch' % (r['value']/1e9, r['roofline']['kernel_ms']*1e3))"
  python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  env.step async8         %.4f G env.step/s  %.1f us/launch' % (r['value']/1e9, r['roofline']['kernel_ms']*1e3))"
  done
  python bench.py --full-line --unfused --steps 512 --warmup 64 --no-cpu-baseline --no-evaluator --samples 3 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  rollout unfused (k_rollout_single) %.3f G' % (r['value']/1e9))"
done
