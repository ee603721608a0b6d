#!/bin/bash
# Runs ON THE GPU BOX: the bench lines DESIGN.md section 6 quotes (no profiling; tools/refresh_round.sh does both).
set -u
R=$1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_driver.json 2> gpurun_out/${R}_bench_driver.err
python bench.py > gpurun_out/${R}_bench_65536x6.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 --coalesce 0 --no-cpu-baseline --no-evaluator > gpurun_out/${R}_bench_driver_nocoalesce.json 2>/dev/null
python bench.py --players 9 --policy allin --no-cpu-baseline --no-evaluator > gpurun_out/${R}_bench_65536x9_allin.json 2>/dev/null
python bench.py --tables 4096 --players 2 --no-cpu-baseline --no-evaluator > gpurun_out/${R}_bench_4096x2.json 2>/dev/null
python bench.py --tables 1048576 --no-cpu-baseline --no-evaluator --samples 3 --min-steps 8192 > gpurun_out/${R}_bench_1048576x6.json 2>/dev/null
python bench.py --unfused --steps 512 --warmup 64 --no-cpu-baseline --no-evaluator --samples 3 > gpurun_out/${R}_bench_65536x6_unfused.json 2>/dev/null
python bench.py --mode env --steps 200 --warmup 20 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env.json
python bench.py --mode env --steps 200 --warmup 20 --env-batches 4 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_sync_batches4.json
python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_async8_batches1.json
python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --env-batches 4 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_async8_batches4.json
python bench.py --mode env --steps 2000 --warmup 200 --env-async 8 --tables 524288 --env-inner-batches 3 2>/dev/null | tail -1 > gpurun_out/${R}_bench_env_async8_inner3_524288.json
python tools/launch_overhead.py > gpurun_out/${R}_launch_overhead.txt 2>&1
python tools/measure_api.py > gpurun_out/${R}_measure_api.txt 2>&1
echo refreshed bench lines $R
