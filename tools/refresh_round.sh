#!/bin/bash
# Runs ON THE GPU BOX: every profile and bench line that DESIGN.md section 6 quotes, for round tag $1 (e.g. r04).
# Afterwards, locally:  tools/refresh_round_local.sh $1     (delete gpurun_out/prof_$1_* locally BEFORE the gpurun call)
# $2 = "profiles_a" (rollouts) | "profiles_b" (PokerGameEnv, evaluator, Game.step) | "profiles" (both) | "benches" | "all" (default): each part fits one gpurun call
set -u
R=$1
WHAT=${2:-all}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
if [ "$WHAT" = "all" ] || [ "$WHAT" = "profiles" ] || [ "$WHAT" = "profiles_a" ]; then
tools/profile_gpu.sh ${R}_65536x6_k20 20 --steps 20 --warmup 5
tools/profile_gpu.sh ${R}_65536x6_k20_nocoalesce 20 --steps 20 --warmup 5 --coalesce 0
tools/profile_gpu.sh ${R}_65536x6_k4096 4096 --steps 4096 --warmup 512
PK_PLAYERS=9 PK_POLICY=allin tools/profile_gpu.sh ${R}_65536x9_allin_k4096 4096 --steps 4096 --warmup 512 --players 9 --policy allin
PK_TABLES=4096 PK_PLAYERS=2 tools/profile_gpu.sh ${R}_4096x2_k4096 4096 --steps 4096 --warmup 512 --tables 4096 --players 2
PK_TABLES=1048576 tools/profile_gpu.sh ${R}_1048576x6_k1024 1024 --steps 1024 --warmup 128 --tables 1048576 --samples 2 --min-steps 4096
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "profiles" ] || [ "$WHAT" = "profiles_b" ]; then
tools/profile_env.sh ${R}_env_sync_65536x6 --steps 300 --warmup 30
tools/profile_env.sh ${R}_env_async8_65536x6 --env-async 8 --steps 2000 --warmup 200
tools/profile_env.sh ${R}_env_async8_inner3_524288x6 --env-async 8 --tables 524288 --env-inner-batches 3 --steps 1000 --warmup 200
tools/profile_eval7.sh ${R}_eval7
tools/profile_step.sh ${R}_step_65536x6 --steps 1000 --warmup 100
tools/profile_step.sh ${R}_step_1048576x6 --tables 1048576 --steps 200 --warmup 50
tools/profile_step.sh ${R}_step_async_65536x6 --steps 1000 --warmup 100 --step-async 1
tools/profile_step.sh ${R}_step_obs2_65536x6 --steps 1000 --warmup 100 --step-replay --step-obs packed --step-obs-separate
tools/profile_step.sh ${R}_step_obsfused_65536x6 --steps 1000 --warmup 100 --step-replay --step-obs packed
fi
if [ "$WHAT" = "all" ] || [ "$WHAT" = "benches" ]; then
cd $ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver_line.json 2> gpurun_out/${R}_bench_driver.err   # the driver's command: the compact line ...
python -c "import json; print(json.dumps(json.load(open('bench_detail.json'))))" > gpurun_out/${R}_bench_driver.json                                                                      # ... and everything behind it
python bench.py --full-line --no-extra > gpurun_out/${R}_bench_65536x6.json 2>/dev/null
python bench.py --full-line --gpus 1 --steps 20 --warmup 5 --coalesce 0 --no-cpu-baseline --no-evaluator --no-extra > gpurun_out/${R}_bench_driver_nocoalesce.json 2>/dev/null
python bench.py --full-line --players 9 --policy allin --no-cpu-baseline --no-evaluator --no-extra > gpurun_out/${R}_bench_65536x9_allin.json 2>/dev/null
python bench.py --full-line --tables 4096 --players 2 --no-cpu-baseline --no-evaluator --no-extra > gpurun_out/${R}_bench_4096x2.json 2>/dev/null
python bench.py --full-line --tables 1048576 --no-cpu-baseline --no-evaluator --no-extra --samples 3 --min-steps 8192 > gpurun_out/${R}_bench_1048576x6.json 2>/dev/null
python bench.py --full-line --unfused --steps 512 --warmup 64 --no-cpu-baseline --no-evaluator --samples 3 > gpurun_out/${R}_bench_65536x6_unfused.json 2>/dev/null
for n in 10 12 13 15 16; do python bench.py --full-line --players $n --no-cpu-baseline --no-evaluator --no-extra --samples 3 --min-steps 131072 > gpurun_out/${R}_bench_65536x$n.json 2>/dev/null; done
python bench.py --mode step --steps 2000 --warmup 200 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_65536x6.json
python bench.py --mode step --steps 2000 --warmup 200 --step-replay 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_65536x6_replay.json
python bench.py --mode step --steps 2000 --warmup 200 --step-unfused-reset 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_65536x6_unfused_reset.json
python bench.py --mode step --tables 1048576 --steps 300 --warmup 50 --step-replay 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_1048576x6_replay.json
python bench.py --mode step --steps 2000 --warmup 200 --step-async 1 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_async_65536x6.json
python bench.py --mode step --steps 2000 --warmup 200 --step-replay --step-obs packed --step-obs-separate 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_obs2_65536x6.json
python bench.py --mode step --steps 2000 --warmup 200 --step-replay --step-obs packed 2>/dev/null | tail -1 > gpurun_out/${R}_bench_step_obsfused_65536x6.json
tools/r06_step_obs_ab.sh gpurun_out/${R}_step_obs_ab.txt > /dev/null 2>&1
python bench.py --mode env --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env.json
python bench.py --mode env --steps 200 --warmup 20 --env-batches 4 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_sync_batches4.json
python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_async8_batches1.json
python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --env-batches 4 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_async8_batches4.json
python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --tables 524288 --env-inner-batches 3 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_async8_inner3_524288.json
python tools/launch_overhead.py > gpurun_out/${R}_launch_overhead.txt 2>&1
python tools/measure_api.py > gpurun_out/${R}_measure_api.txt 2>&1
python tools/step_sweep.py 6 > gpurun_out/${R}_step_sweep.txt 2>&1
python tools/eval_hands_bench.py > gpurun_out/${R}_eval_hands_bench.txt 2>&1
for c in 128 256 512 1024 2048; do echo "coalesce $c: $(python bench.py --full-line --gpus 1 --steps 20 --warmup 5 --coalesce $c --no-cpu-baseline --no-evaluator --no-extra --samples 3 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('%.2f G  %s' % (r['value']/1e9, r['config']['launch_stats']))")"; done > gpurun_out/${R}_coalesce_sweep.txt 2>&1
fi
echo refreshed $R $WHAT
