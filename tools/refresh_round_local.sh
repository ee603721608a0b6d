#!/bin/bash
# Build container: distil what tools/refresh_round.sh left under gpurun_out/ into profiles/ and print the figures.
R=$1
cd "$(dirname "$0")/.."
for t in ${R}_65536x6_k4096 ${R}_65536x6_k20 ${R}_65536x6_k20_nocoalesce ${R}_65536x9_allin_k4096 ${R}_4096x2_k4096 ${R}_1048576x6_k1024; do
  [ -d gpurun_out/prof_$t ] || continue
  python tools/summarize_profile.py $t > /tmp/sum_$t.txt 2>&1 || tail -3 /tmp/sum_$t.txt
  python - <<PY
import json
d = json.load(open('profiles/${t}_summary.json'))
keys = ('launches_total', 'avg_launch_ms', 'vgpr', 'lds_bytes', 'hbm_traffic_bytes_per_launch', 'valu_insts_per_wave_step',
        'salu_insts_per_wave_step', 'valu_issue_frac_of_peak', 'lanes_active', 'wait_any_frac_of_wave_cycles')
print('$t', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in keys})
PY
done
for t in ${R}_env_sync_65536x6 ${R}_env_async8_65536x6 ${R}_env_async8_inner3_524288x6; do
  [ -d gpurun_out/prof_$t ] || continue
  python tools/summarize_env_profile.py $t > /tmp/sum_$t.txt 2>&1 || tail -3 /tmp/sum_$t.txt
  python - <<PY
import json
d = json.load(open('profiles/${t}_summary.json'))
keys = ('launches_total', 'avg_launch_ms', 'vgpr', 'valu_insts_per_wave', 'lanes_active', 'hbm_traffic_bytes_per_launch',
        'hbm_frac_of_peak_in_the_traced_run', 'valu_issue_frac_of_peak_in_the_traced_run', 'wait_any_frac_of_wave_cycles')
print('$t', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k in keys})
PY
done
[ -d gpurun_out/prof_${R}_eval7 ] && python tools/summarize_eval7.py ${R}_eval7 | tail -3
for t in ${R}_step_65536x6 ${R}_step_1048576x6 ${R}_step_async_65536x6 ${R}_step_obs2_65536x6 ${R}_step_obsfused_65536x6; do
  [ -d gpurun_out/prof_$t ] || continue
  python tools/summarize_step_profile.py $t > /tmp/sum_$t.txt 2>&1 || tail -3 /tmp/sum_$t.txt
  python - <<PY
import json
d = json.load(open('profiles/${t}_summary.json'))
print('$t', {k: (round(v['avg_ms'] * 1e3, 2), round(v.get('hbm_traffic_bytes_per_launch', 0) / 1e6, 1)) for k, v in d['kernels'].items()}, 'k_step hbm frac', round(d.get('k_step_hbm_frac_algorithmic', 0), 3), 'traffic/alg', d.get('traffic_over_algorithmic'))
PY
done
for f in launch_overhead coalesce_sweep measure_api step_sweep eval_hands_bench step_obs_ab; do [ -f gpurun_out/${R}_$f.txt ] && cp gpurun_out/${R}_$f.txt profiles/${R}_$f.txt; done
[ -s gpurun_out/${R}_bench_driver_line.json ] && tail -1 gpurun_out/${R}_bench_driver_line.json > profiles/${R}_bench_driver_line.json && wc -c profiles/${R}_bench_driver_line.json
for f in driver driver_nocoalesce 65536x6 65536x9_allin 4096x2 1048576x6 65536x6_unfused 65536x10 65536x12 65536x13 65536x15 65536x16 step_65536x6 step_65536x6_replay step_65536x6_unfused_reset step_1048576x6_replay step_async_65536x6 step_obs2_65536x6 step_obsfused_65536x6 env env_sync_batches4 env_async8_batches1 env_async8_batches4 env_async8_inner3_524288; do
  [ -s gpurun_out/${R}_bench_$f.json ] || continue
  tail -1 gpurun_out/${R}_bench_$f.json > profiles/${R}_bench_$f.json
  python - <<PY
import json
d = json.load(open('profiles/${R}_bench_$f.json'))
r = d.get('roofline') or {}
print('$f', 'value %.4g' % d['value'], 'ms/step %.5f' % d['ms_per_step'], 'kern_ms', r.get('kernel_ms'), 'frac', r.get('frac'),
      'bound', r.get('bound'), 'steps/launch', r.get('steps_per_launch'), 'valu/wave-step', r.get('valu_insts_per_wave_step'), 'lanes', r.get('lanes_active'), 'ceil', (r.get('ceiling_mix') or {}).get('frac_of_ceiling'),
      'evals/s', d.get('hand_evals_per_s'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), 'game_steps/s', d.get('game_steps_per_s'), 'hbm', (r.get('hbm') or {}).get('frac'))
for x in d.get('extra_workloads', []):
    xr = x['roofline']
    print('   extra:', x['name'][:90], '| %.4g %s' % (x['value'], x['unit']), '| frac', xr.get('frac'), '| hbm', (xr.get('hbm') or {}).get('frac'), '| src', xr.get('source'))
PY
done
