for tpb in 64 32 16 8; do
  echo "== PK_ENV_TPB=$tpb"
  PK_ENV_TPB=$tpb python bench.py --mode env --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  sync fused      %.3f G env.step/s'%(r['value']/1e9))"
  PK_ENV_TPB=$tpb python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  async8 1 batch  %.3f G env.step/s ready %.2f'%(r['value']/1e9, r['ready_fraction_per_launch']))"
  PK_ENV_TPB=$tpb python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --env-batches 4 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('  async8 4 batch  %.3f G env.step/s'%(r['value']/1e9))"
done
