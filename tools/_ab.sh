for lib in libpokerl_hip_old.so libpokerl_hip.so; do
POKERL_HIP_LIB=$PWD/pokerl_amd/$lib timeout -k 10 120 python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib async x1', '%.4f G env.step/s'%(d['value']/1e9), d['ready_fraction_per_launch'])"
POKERL_HIP_LIB=$PWD/pokerl_amd/$lib timeout -k 10 120 python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --env-batches 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib async x4', '%.4f G env.step/s'%(d['value']/1e9))"
POKERL_HIP_LIB=$PWD/pokerl_amd/$lib timeout -k 10 120 python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --tables 524288 --env-inner-batches 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib async inner3', '%.4f G env.step/s'%(d['value']/1e9))"
done
