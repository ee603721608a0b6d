#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path: env-steps/s (+ showdown hand-evals/s) of random-agent 6-max NLHE,
65 536 tables per MI355X (BASELINE.json configs[2]; N GPUs = configs[3] weak scaling, 65 536 tables per GPU).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one lockstep pass of the hot path over the whole batch: one Game.step() on every table (T env-steps).
Table state is resident in HBM before the timed region starts; agents run in-kernel (Philox), finished games
auto-reset.  Rank 0 prints ONE JSON line.  torch is used only for torch.distributed (barrier / max / sum across the
one-process-per-GPU ranks); the product path itself is plain HIP behind a ctypes C ABI.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s)


def b_step(n):
    """Algorithmic bytes per env-step: read + write the table state once, plus action in and
    mask/flags out: 2*(35N+21)+16 -> 478 B at N=6."""
    return 2 * (35 * n + 21) + 16


class DistContext:
    """One process per GPU.  Barrier, MAX of the timed seconds over ranks, SUM of the units all ranks processed."""

    def __init__(self, backend=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend is None:  # PK_BENCH_BACKEND=gloo: rehearsals where several ranks share one GPU (RCCL refuses that)
                backend = os.environ.get("PK_BENCH_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            # RCCL prints a banner (HIP / ROCm version, library path) on stdout when it initialises: keep stdout for the
            # ONE JSON line by pointing fd 1 at stderr until the first collective has run.
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                if backend == "nccl":
                    torch.cuda.set_device(self.local_rank)
                    self.device = torch.device("cuda", self.local_rank)
                    dist.init_process_group("nccl", device_id=self.device)   # "nccl" is RCCL on ROCm
                    warm = torch.zeros(1, device=self.device)
                    dist.all_reduce(warm)
                    torch.cuda.synchronize()
                else:
                    dist.init_process_group("gloo")
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.dist = dist
            self.torch = torch

    def barrier(self):
        if self.dist is not None:
            if self.device is not None:
                t = self.torch.zeros(1, device=self.device)
                self.dist.all_reduce(t)
                self.torch.cuda.synchronize()
            else:
                self.dist.barrier()

    def aggregate(self, local_units, local_seconds):
        """(sum of units over ranks, max of seconds over ranks)."""
        if self.dist is None:
            return local_units, local_seconds
        torch, dist = self.torch, self.dist
        dev = self.device if self.device is not None else "cpu"
        u = torch.tensor([float(local_units)], dtype=torch.float64, device=dev)
        s = torch.tensor([float(local_seconds)], dtype=torch.float64, device=dev)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        dist.all_reduce(s, op=dist.ReduceOp.MAX)
        return int(round(u.item())), s.item()

    def sum_list(self, values):
        if self.dist is None:
            return list(values)
        torch, dist = self.torch, self.dist
        dev = self.device if self.device is not None else "cpu"
        v = torch.tensor([float(x) for x in values], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
        return [int(round(x)) for x in v.tolist()]

    def describe(self, device, n_local):
        """`dist` block of the bench line for N > 1 (None for one process): backend, world size, every rank's device index and
        table count -- "did the collective library see N ranks" is checkable from the line.  Collective: call on every rank."""
        if self.dist is None:
            return None
        devs = self.sum_list([device if r == self.rank else 0 for r in range(self.world)])
        tabs = self.sum_list([n_local if r == self.rank else 0 for r in range(self.world)])
        return {"backend": self.dist.get_backend(), "world": self.dist.get_world_size(), "devices": devs, "tables_per_rank": tabs,
                "collectives_on_step_path": 0}

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


def shard(total_tables, ctx):
    from pokerl_amd import shard_tables
    return shard_tables(total_tables, ctx.rank, ctx.world)


def baseline_config_index(tables, players, policy, world):
    """Which entry of BASELINE.json `configs` this run is (None: a parity-test shape, not a listed config)."""
    if world > 1:
        return 3 if (tables, players, policy) == (65536, 6, "random") else None
    return {(4096, 2, "random"): 1, (65536, 6, "random"): 2, (65536, 9, "allin"): 4}.get((tables, players, policy))


VALU_PEAK_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 2.0   # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles at 2.4 GHz
#   (guides/MI355X_MICROARCH.md: "v_fma_f32 (wave64) 2 cyc (SIMD-32); one wave alone: 4")


# Cycles per wave-instruction measured by tools/microbench/valu_rates.hip on MI355X (profiles/r05_valu_rates.txt, re-run in round 5:
# the same figures as r03_valu_rates.txt to the second digit), by waves per SIMD: (the plain VOP2 ALU kinds -- v_add / v_sub / v_and / v_or / v_xor, independent
# stream; everything else the step machine is made of -- every f64 op, shifts, v_bfe, v_bcnt, v_ffbh, multiplies, v_perm,
# v_max, the three-operand forms v_or3 / v_lshl_or / v_max3 / v_lshl_add_u64, v_cndmask on an SGPR mask -- which all issue
# at half rate).  A lone wave issues ~one instruction per 5 cycles whatever its kind -- REAL cycles: the probe at the end of the
# microbenchmark reads the shader-clock counter against the 100 MHz one (2 397 MHz, 4.75 ticks per instruction of an independent v_add_u32
# stream); the guide's "one wave alone: 4" is the pipeline time of a wave64 instruction, the rest is the issue gap a lone wave cannot
# fill (SQ_ACTIVE_INST_ANY = 79 % of the wave cycles in k_rollout).  Three waves per SIMD are interpolated
# (towards the four-wave figures: with the two-wave ones the 1 M-table run exceeded its own ceiling by 1 %).
ISSUE_CYCLES = {1: (5.0, 5.0), 2: (2.7, 4.5), 3: (2.5, 4.3), 4: (2.45, 4.3)}
HALF_RATE_SHARE = 0.60   # fallback only (round 3's hand estimate): half_rate_share() below reads the figure GENERATED from
#                          k_rollout<6>'s ISA (tools/isa_report.py -> profiles/rNN_isa_report.json: v_mov and the 32-bit v_cmp
#                          count as plain; v_cndmask, every f64 op, shifts / bfe / bcnt / ffbl / mul / perm, the three-operand
#                          and 64-bit integer forms as half rate).  The ceiling stays a model, not a measurement.


def mix_ceiling(waves_per_simd):
    """Issue rate this kernel's instruction MIX can reach at its occupancy: (wave-instr/s, cycles per instruction)."""
    full, half = ISSUE_CYCLES[max(1, min(4, int(round(waves_per_simd))))]
    share = half_rate_share()[0]
    cyc = share * half + (1.0 - share) * full
    return 256 * 4 * 2.4e9 / cyc, cyc


def profile_summary(tables, players, policy, kern_steps=None):
    """The committed rocprofv3 summary (profiles/*_summary.json: kernel trace + separate PMC passes of THIS bench command,
    made by tools/profile_gpu.sh + tools/summarize_profile.py) for this workload -- the one profiled at the launch length
    closest to this run's (instructions per wave-step depend a little on it: ramp and tail of every launch), the latest
    among equals; None if none."""
    import glob
    import math
    import re
    cands = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("tables"), w.get("players"), w.get("policy")) == (tables, players, policy) and w.get("fused", True):
            m = re.match(r"r(\d+)_", os.path.basename(f))
            k = w.get("steps_per_launch") or 0
            dist = abs(math.log(max(k, 1e-9) / kern_steps)) if kern_steps else 0.0
            cands.append((int(m.group(1)) if m else -1, -dist, d, os.path.basename(f)))
    if not cands:
        return None
    newest = max(c[0] for c in cands)                      # only the latest round's profiles describe the current kernels
    best = max((c for c in cands if c[0] == newest), key=lambda c: c[1])
    return best[2], best[3]


def profile_stale(summary):
    """True when a committed counter summary was measured on OTHER kernel sources than the library now running (its `source_hash` -- stamped by
    the tools/summarize_*.py scripts from pk_build_info -- differs, or it predates the stamp): every figure derived from it is then a figure
    of an older kernel."""
    from pokerl_amd import _lib as L
    return summary.get("source_hash") != L.source_hash()


def evaluator_leg(device, log2_m=None, reps=20):
    """Second half of the metric as a stand-alone kernel: pk_eval7_d streams 2^28 device-resident 7-card hands
    (2 GiB in, 1 GiB out: far beyond L2 / Infinity Cache) -- 12 algorithmic bytes per evaluation, HBM by definition.
    Counter evidence (what bounds it in practice) comes from the committed rocprofv3 PMC summary of the same kernel."""
    log2_m = int(os.environ.get("PK_BENCH_EVAL_LOG2", "28")) if log2_m is None else log2_m
    from pokerl_amd import judger
    from pokerl_amd.hipmem import DeviceBuffer
    m = 1 << log2_m
    hands, out = DeviceBuffer(m * 8, device), DeviceBuffer(m * 4, device)
    judger.make_hands(hands.ptr, m, device=device)
    ms = judger.time_eval7_stream(hands.ptr, m, out.ptr, True, reps, device)
    hands.free(); out.free()
    gbs = 12.0 * m / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "bytes_per_eval": 12, "traffic": None}
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_eval7_summary.json")))
    src = cands[-1] if cands else ""
    if src:
        d = json.load(open(src))
        per_eval = d.get("hbm_traffic_bytes_per_launch", 0) / float(d.get("hands_per_launch", 1))
        roof.update({"profile_stale": profile_stale(d), "traffic": per_eval * m, "traffic_unit": "HBM bytes per launch of this size (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, KiB), = %.3f x algorithmic" % (per_eval / 12.0),
                     "source": os.path.basename(src), "valu_insts_per_eval": d.get("valu_wave_insts_per_eval_x64"),
                     "lds_lookups_per_eval": d.get("lds_wave_insts_per_eval_x64"),
                     "valu_issue_frac_of_peak": d.get("valu_issue_frac_of_peak"),
                     "wait_inst_any_frac": d.get("wait_inst_any_frac_of_wave_cycles"),
                     "note": "streams exactly its algorithmic bytes; what keeps it below the HBM roofline is VALU issue: %.0f wave "
                             "instructions per 64 evaluations, mostly half-rate kinds (shifts, v_bcnt, v_cndmask), issue-stalled "
                             "%.0f %% of the wave cycles (SQ_WAIT_INST_ANY)" % (d.get("valu_wave_insts_per_eval_x64", 0),
                                                                               100 * d.get("wait_inst_any_frac_of_wave_cycles", 0))})
    return {"kernel": "k_eval7_tab_stream (pk_eval7_d, 7 distinct cards, 32 KB rank-mask table in LDS)", "hands": m,
            "hand_evals_per_s": m / (ms * 1e-3), "kernel_ms": ms, "roofline": roof}


def cpu_baseline(n_players, policy, budget_s=None):
    """The scalar C oracle (bit-exact restatement of the reference) timed on ONE host core on a bounded sample of the
    same workload.  Reported beside the GPU number; it is not the target (the roofline fraction is)."""
    budget_s = float(os.environ.get("PK_BENCH_CPU_BUDGET", "3")) if budget_s is None else budget_s
    import numpy as np
    from oracle import loader as O
    tables, chunk = 2048, 50
    g = O.OracleGame(tables, n_players)
    g.reset()
    g.rollout(chunk, policy, True)  # warm
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        c, _ = g.rollout(chunk, policy, True)
        steps += int(c[0])
    dt = time.perf_counter() - t0
    out = dict(value=steps / dt, unit="env-steps/s", cores=1, kind="port",
               sample="%d tables x %d lockstep steps, random agents, N=%d, oracle/pokerl_oracle.c single thread"
                      % (tables, steps // tables, n_players))
    # the same restatement on every host core (one independent batch per thread; ctypes releases the GIL), SURVEY 8d
    from concurrent.futures import ThreadPoolExecutor
    ncores = max(1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    ncores = min(ncores, 16)  # a one-GPU box's CPU share is 16 cores, whatever the host exposes
    games = [O.OracleGame(tables, n_players, table_id_base=tables * (i + 1)) for i in range(ncores)]
    for gg in games:
        gg.reset()

    def work(gg):
        n, t_end = 0, time.perf_counter() + budget_s * 2 / 3
        while time.perf_counter() < t_end:
            n += int(gg.rollout(chunk, policy, True)[0][0])
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(ncores) as ex:
        total = sum(ex.map(work, games))
    out["all_cores"] = dict(value=total / (time.perf_counter() - t0), unit="env-steps/s", cores=ncores)
    return out


ENV_KERNEL = {"sync": "k_env_step", "async": "k_env_step_async"}


def env_profile_summary(tables, players, batches, inner, async_passes):
    """The committed rocprofv3 summary of THIS PokerGameEnv workload (profiles/rNN_env_*_summary.json, made by
    tools/profile_env.sh + tools/summarize_env_profile.py), latest round; None if none."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_env_*_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("tables"), w.get("players"), w.get("env_batches", 1), w.get("env_inner_batches", 1), w.get("env_async", 0)) == \
                (tables, players, batches, inner, async_passes):
            m = re.match(r"r(\d+)_", os.path.basename(f))
            key = int(m.group(1)) if m else -1
            if best is None or key >= best[0]:
                best = (key, d, os.path.basename(f))
    return (best[1], best[2]) if best else None


def env_workload(ctx, device, tables, players, batches=1, inner=1, async_passes=0, steps=200, warmup=20, unfused=False):
    """The RL-facing path as a device-resident loop: seat 0 picks with the in-kernel random agent,
    PokerGameEnv.step auto-plays the opponents, finished episodes are reset, the observation row is written -- ONE launch
    per step (pk_env_step_fused_d), or bounded launches whose unfinished steps stay in flight (pk_env_step_async_d,
    async_passes > 0; `inner` > 1: sub-batches inside each handle).  Timed with wall clock AND HIP events on each handle's
    stream (pk_record_event waits for the launches on the handle's internal streams too).  Returns the measurements."""
    import ctypes as C
    import numpy as np
    import pokerl_amd
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceBuffer, DeviceEvent
    T, N, B = tables, players, max(1, batches)
    D = 17 + 3 * N
    lib = L.lib()

    # ONE environment object: a pool of B batches of T tables, each batch on its own handle and stream (global table ids:
    # the pool's tables play what a single handle of B*T tables would)
    pool = pokerl_amd.VecPokerGameEnvPool(0, num_tables=B * T, num_batches=B, num_players=N, device=device,
                                          table_id_base=ctx.rank * B * T)

    class Batch:   # the launches of one batch of the pool and its output buffers
        def __init__(self, b):
            self.env = pool.envs[b]
            self.g = self.env.game
            self.act, self.rew, self.done, self.hand, self.terr, self.obs = (
                DeviceBuffer(T * 4, device), DeviceBuffer(T * 8, device), DeviceBuffer(T, device),
                DeviceBuffer(T, device), DeviceBuffer(T, device), DeviceBuffer(T * D * 8, device))
            L.check(lib.pk_env_reset_d(self.g._h, None, 0), self.g._h)
            self.inner = self.env.set_env_batches(inner) if (async_passes > 0 and inner > 1) else 1
            # async: one ready[T] slice per timed launch, summed after the timed region
            self.ready = DeviceBuffer(T * (steps + 1), device) if async_passes > 0 else None
            self.launch = 0
            self.delivered = []     # per timed launch: (slot, begin, end) of the delivered range (inner batches)
            self.ev0, self.ev1 = DeviceEvent(), DeviceEvent()

        def step(self, timed=False):
            g, h = self.g, self.g._h
            if async_passes > 0:   # bounded launches: tables whose env.step has not returned stay in flight
                slot = 1 + self.launch if timed else 0
                self.launch += 1 if timed else 0
                if self.inner > 1 and timed:    # the range this call LAUNCHES = the range the previous call delivered; its ready
                    self.delivered.append((slot, self.env.last_range()[:2]))   # flags go to this call's slot
                L.check(lib.pk_env_step_async_d(h, None, 0, 0, 1, async_passes, self.rew.ptr, self.done.ptr, self.hand.ptr,
                                                self.terr.ptr, self.obs.ptr, C.c_void_p(self.ready.ptr.value + slot * T)), h)
            elif unfused:   # five launches per env step
                L.check(lib.pk_pick_actions_d(h, 0, self.act.ptr), h)
                L.check(lib.pk_env_step_d(h, self.act.ptr, 0, self.rew.ptr, self.done.ptr, self.hand.ptr, self.terr.ptr), h)
                L.check(lib.pk_env_reset_d(h, self.done.ptr, 0), h)   # finished episodes ...
                L.check(lib.pk_env_reset_d(h, self.terr.ptr, 0), h)   # ... and tables the reference would never return from
                L.check(lib.pk_get_obs_d(h, -1, self.obs.ptr), h)     #     (PK_TERR_HAND_CAP: ~1 per 25 M game steps)
            else:                  # the same work in ONE launch: seat 0 in-kernel, auto-reset, observation from registers
                L.check(lib.pk_env_step_fused_d(h, None, 0, 0, 1, self.rew.ptr, self.done.ptr, self.hand.ptr, self.terr.ptr,
                                                self.obs.ptr), h)

        def free(self):
            for b in (self.act, self.rew, self.done, self.hand, self.terr, self.obs, self.ready):
                if b is not None:
                    b.free()

    bs = [Batch(b) for b in range(B)]

    def loop(k, timed=False):   # round-robin over the batches: each handle launches on its own stream, nothing waits in between
        for _ in range(k):
            for b in bs:
                b.step(timed)

    def sync():
        for b in bs:
            b.g.sync()

    loop(warmup)
    sync()
    s0 = sum(int(b.g.step_serial.sum()) for b in bs)
    ctx.barrier()
    for b in bs:
        b.g.record_event(b.ev0.handle)
    t0 = time.perf_counter()
    loop(steps, True)
    for b in bs:
        b.g.record_event(b.ev1.handle)      # after everything launched so far, the handle's internal streams included
    sync(); ctx.barrier()
    dt = time.perf_counter() - t0
    dev_ms = max(DeviceEvent.elapsed_ms(b.ev0, b.ev1) for b in bs)
    env_steps = B * T * steps
    launched_tables = env_steps
    if async_passes > 0 and bs[0].inner > 1:     # a call steps ONE range of its handle
        launched_tables = sum(r1 - r0 for b in bs for _, (r0, r1) in b.delivered)
    if async_passes > 0:   # delivered env.steps = ready flags of the timed launches; then drain so that the tables can be read
        def delivered_steps(b):
            flags = b.ready.download(np.uint8, T * (steps + 1))
            if b.inner <= 1:
                return int(flags[T:].sum(dtype=np.int64))
            # inner batches: every timed call wrote the flags of the ONE range it launched into its own slot
            return sum(int(flags[slot * T + r0:slot * T + r1].sum(dtype=np.int64)) for slot, (r0, r1) in b.delivered)
        env_steps = sum(delivered_steps(b) for b in bs)
        for b in bs:
            L.check(lib.pk_env_step_async_d(b.g._h, None, 0, 0, 1, 0, b.rew.ptr, b.done.ptr, b.hand.ptr, b.terr.ptr, b.obs.ptr,
                                            b.ready.ptr), b.g._h)
        sync()
    game_steps = sum(int(b.g.step_serial.sum()) for b in bs) - s0
    capped = sum(int((b.terr.download(np.uint8, T) != 0).sum()) for b in bs)
    launches = B * steps * (5 if (unfused and async_passes == 0) else 1)
    res = dict(tables=T, players=N, batches=B, inner=bs[0].inner, async_passes=async_passes, unfused=unfused, steps=steps, warmup=warmup,
               seconds=dt, device_ms=dev_ms, env_steps=env_steps, launched_tables=launched_tables, game_steps=game_steps,
               capped=capped, launches=launches)
    for b in bs:
        b.free()
    pool.close()
    return res


def env_roofline(res):
    """Roofline block of a PokerGameEnv workload: VALU issue (binding) from the committed PMC summary of THAT workload x
    this run's HIP-event time, with the measured HBM traffic of the same passes beside it."""
    kern = ENV_KERNEL["async" if res["async_passes"] > 0 else "sync"]
    ms_launch = res["device_ms"] / max(1, res["launches"])
    roof = {"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_WAVE_INSTS_PER_S, "unit": "wave-instr/s", "frac": None,
            "traffic": None, "kernel": kern, "kernel_ms": ms_launch, "launches_timed": res["launches"],
            "kernel_ms_source": "HIP events (pk_record_event) around the timed region on each handle's stream (the launches on a handle's "
                                "internal sub-batch streams are waited for first); the region's device time / the launches inside it -- "
                                "launches of different handles / sub-batches OVERLAP, so this is the effective time per launch"}
    prof = env_profile_summary(res["tables"], res["players"], res["batches"], res["inner"], res["async_passes"]) if not res["unfused"] else None
    if prof:
        d, src = prof
        per_launch = d.get("valu_wave_insts_per_launch")
        if per_launch:
            rate = per_launch * res["launches"] / (res["device_ms"] * 1e-3)
            resident = d.get("waves_per_simd_resident") or 1.0
            ceil_rate, ceil_cyc = mix_ceiling(resident)
            traffic = d.get("hbm_traffic_bytes_per_launch")
            roof.update({"achieved": rate, "frac": rate / VALU_PEAK_WAVE_INSTS_PER_S, "source": src, "profile_stale": profile_stale(d),
                         "valu_wave_insts_per_launch": per_launch, "lanes_active": d.get("lanes_active"),
                         "waves_per_simd": resident, "wait_any_frac": d.get("wait_any_frac_of_wave_cycles"),
                         "ceiling_mix": {"peak": ceil_rate, "frac_of_ceiling": rate / ceil_rate, "cycles_per_instruction": ceil_cyc,
                                         "half_rate_share": half_rate_share()[0], "source": "profiles/r05_valu_rates.txt + " + half_rate_share()[1]},
                         "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, KiB)",
                         "hbm": {"bound": "hbm (measured traffic)", "achieved": (traffic or 0.0) * res["launches"] / (res["device_ms"] * 1e-3) / 1e9,
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": (traffic or 0.0) * res["launches"] / (res["device_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "note": "real traffic of the committed PMC passes (table state in and out once per launch + the "
                                         "observation rows and outputs of the delivered tables) x this run's launches / its device time"},
                         "note": "achieved = wave-level VALU instructions per launch (SQ_INSTS_VALU of the COMMITTED rocprofv3 PMC pass of "
                                 "this workload) x this run's launches / its HIP-event device time; peak = 256 CU x 4 SIMD x 2.4 GHz / 2"})
    if roof["achieved"] is None:
        roof["note"] = "no rocprofv3 PMC summary of this PokerGameEnv workload under profiles/ (tools/profile_env.sh makes one)"
    return roof


def env_line(res, ctx):
    B, T, N = res["batches"], res["tables"], res["players"]
    dt = res["seconds"]
    return {
        "metric": "PokerGameEnv.step seat-0 steps/s (device-resident loop: pick + env_step + env_reset(done) + obs; %s)"
                  % ("bounded launches of %d betting passes, steps that have not returned stay in flight "
                     "(pk_env_step_async_d); value counts DELIVERED env.steps" % res["async_passes"] if res["async_passes"] > 0 else
                     "five launches per step" if res["unfused"] else "fused into one launch per step"),
        "value": res["env_steps"] / dt, "unit": "env.step/s", "n_gpus": ctx.world, "steps": res["steps"],
        "warmup": res["warmup"], "ms_per_step": dt / res["steps"] * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%d batch(es) x %d tables x %d seats per GPU, seat 0 + opponents random in-kernel, "
                               "episodes auto-reset%s; reference pokerl/envs/game_env.py:20-53"
                               % (B, T, N, ", each handle in %d sub-batches (pk_set_env_batches): one call launches one of them"
                                  % res["inner"] if res["inner"] > 1 else ""),
                   "env_batches": B, "env_inner_batches": res["inner"],
                   "note": "one env.step of a batch lasts as long as its slowest table (a busted seat 0 waits for the "
                           "rest of the game, game_env.py:44-47); independent batches on their own streams fill that tail"},
        "game_steps_per_s": res["game_steps"] / dt, "game_steps_per_env_step": res["game_steps"] / max(1, res["env_steps"]),
        "ready_fraction_per_launch": res["env_steps"] / float(res["launched_tables"]), "inner_batches": res["inner"],
        "tables_with_error_bits_in_last_step": res["capped"], "device_ms": res["device_ms"], "launches": res["launches"],
        "roofline": env_roofline(res)}


def env_mode(args, ctx, device):
    """--mode env: the RL-facing path as its own bench line (not the headline metric)."""
    res = env_workload(ctx, device, args.tables, args.players, args.env_batches, args.env_inner_batches, args.env_async,
                       args.steps, args.warmup, args.env_unfused)
    if ctx.rank == 0:
        print(json.dumps(env_line(res, ctx)))


def rollout_workload(ctx, device, tables, players, policy_name, K, warmup, chunk=4096, reps=0, min_steps=524288, samples=7,
                     fused=True, coalesce=-1):
    """The fused rollout of one configuration: `samples` timed samples of reps x K steps per table, each bracketed by HIP
    events on the handle's stream and completed by a sync; the device counters prove the steps were executed."""
    import pokerl_amd
    from pokerl_amd.hipmem import DeviceEvent
    policy = 0 if policy_name == "random" else 1
    total_tables = tables * ctx.world
    n_local, base = shard(total_tables, ctx)
    game = pokerl_amd.VecGame(n_local, num_players=players, device=device, table_id_base=base)
    game.reset()
    if coalesce >= 0:
        game.set_coalesce(coalesce)
    K = max(1, K)
    if reps <= 0:   # auto: as many blocks as make a sample >= min_steps steps (unfused: one block -- K launches -- per sample)
        reps = max(1, -(-min_steps // K)) if fused else 1

    def block():  # K steps on every table; launches are asynchronous and may defer their stragglers to the next launch
        done = 0
        while done < K:
            k = min(chunk, K - done)
            game.rollout(k, policy, True, fused, counters=False)
            done += k

    done = 0
    while done < warmup:
        k = min(chunk, warmup - done)
        game.rollout(k, policy, True, fused, counters=False)
        done += k
    game.rollout(0, policy, True, fused, counters=True)  # complete + zero the device counters
    ev0, ev1 = DeviceEvent(), DeviceEvent()
    stats_warm = game.launch_stats(reset=True)           # launches so far (reset + warm-up): a profiler sees those too
    sample_s, sample_dev_ms, sample_stats = [], [], []
    for _ in range(max(1, samples)):
        ctx.barrier(); game.sync()
        game.record_event(ev0.handle)                    # HIP events on the stream the kernel is launched on
        t0 = time.perf_counter()
        for _ in range(reps):
            block()
        game.record_event(ev1.handle)                    # completes every deferred / host-held step first, then records
        game.sync(); ctx.barrier()   # exactly reps*K steps per table are inside
        sample_s.append(ctx.aggregate(0, time.perf_counter() - t0)[1])   # MAX over ranks
        sample_dev_ms.append(DeviceEvent.elapsed_ms(ev0, ev1))
        sample_stats.append(game.launch_stats(reset=True))
    stats = dict(launches=sum(x["launches"] for x in sample_stats), steps=sum(x["steps"] for x in sample_stats),
                 min=min(x["min"] for x in sample_stats), max=max(x["max"] for x in sample_stats))
    c = game.rollout(0, policy, True, fused, counters=True)
    assert c["steps"] == n_local * K * reps * len(sample_s), (c, n_local, K, reps)
    assert stats["steps"] == K * reps * len(sample_s), (stats, K, reps)
    med = sorted(range(len(sample_s)), key=lambda i: sample_s[i])[len(sample_s) // 2]   # the MEDIAN sample: value, ms_per_step and kernel_ms are all its
    seconds = sample_s[med]
    total_steps = ctx.aggregate(n_local * K * reps, 0.0)[0]
    scale = 1.0 / (sum(sample_s))   # counters cover all samples
    hands, evals, games = ctx.sum_list([c["hands"], c["evals"], c["games"]])
    # roofline leg: the timed region itself, bracketed by HIP events on the handle's stream; the dominant kernel is the only
    # kernel in it, so its average launch duration (launch gaps included) = event time / launches
    launches = max(1, sample_stats[med]["launches"])
    ms_launch = sample_dev_ms[med] / launches            # of the median sample, like `value` and `ms_per_step`
    kern_steps = sample_stats[med]["steps"] / float(launches)        # mean Game.step()s per table per launch
    game.close()
    return dict(tables=tables, players=players, policy=policy_name, K=K, warmup=warmup, chunk=chunk, reps=reps, fused=fused,
                n_local=n_local, seconds=seconds, total_steps=total_steps, sample_s=sample_s, sample_dev_ms=sample_dev_ms,
                stats=stats, stats_warm=stats_warm, launches=launches, ms_launch=ms_launch, kern_steps=kern_steps,
                hand_evals_per_s=evals * scale, hands_per_s=hands * scale, games_per_s=games * scale)


_HALF = None


def half_rate_share():
    """(share, source): half-rate share of k_rollout<6>'s static VALU instructions, GENERATED from its ISA by tools/isa_report.py
    (profiles/rNN_isa_report.json, made by tools/resource_usage.sh); the hand estimate of round 3 if no report is committed."""
    global _HALF
    if _HALF is None:
        import glob
        _HALF = (HALF_RATE_SHARE, "bench.py HALF_RATE_SHARE (hand estimate)")
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_isa_report.json"))):
            try:
                six = json.load(open(f))["6"]
                kern = "k_rollout_tab<6>" if "k_rollout_tab<6>" in six else "k_rollout<6>"       # (what the headline workload runs since round 6)
                _HALF = (float(six[kern]["half_rate_share_static"]), "profiles/" + os.path.basename(f) + " (opcode histogram of %s's ISA)" % kern)
            except Exception:
                pass
    return _HALF


def rollout_roofline(w, world):
    """The `roofline` object of a rollout workload (dict from rollout_workload): VALU issue leads (binding), SURVEY 8d's
    algorithmic-HBM figure and the measured traffic are nested."""
    n_local, kern_steps, ms_launch, launches = w["n_local"], w["kern_steps"], w["ms_launch"], w["launches"]
    alg_bytes = b_step(w["players"]) * n_local * kern_steps
    achieved = alg_bytes / (ms_launch * 1e-3) / 1e9
    prof = profile_summary(w["tables"], w["players"], w["policy"], int(round(kern_steps))) if w["fused"] else None
    hbm = {"bound": "hbm (algorithmic bytes, SURVEY 8d)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg_bytes,
           "note": "ALGORITHMIC bytes ((2*(35N+21)+16) B/env-step x tables x steps per launch) / launch time: what a "
                   "one-HBM-round-trip-per-step design would move.  The fused kernel keeps the table state in VGPRs "
                   "for all steps of a launch and really moves `traffic` (one read + one write of the state per "
                   "launch), so this fraction can exceed 1 and bounds nothing; kept because SURVEY 8d defines it."}
    roof = {"bound": "valu-issue", "achieved": None, "peak": VALU_PEAK_WAVE_INSTS_PER_S, "unit": "wave-instr/s",
            "frac": None, "traffic": None,
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, KiB)",
            "kernel_ms": ms_launch, "launches_timed": launches, "steps_per_launch": kern_steps,
            "kernel_ms_source": "HIP events (pk_record_event) around the timed samples on the handle's stream; the median "
                                "sample's device time / the launches inside it (pk_get_launch_stats)",
            "hbm_algorithmic": hbm}
    if prof:
        d, src = prof
        pmc = d.get("pmc_per_full_launch", {})
        roof["traffic"] = hbm["traffic"] = d.get("hbm_traffic_bytes_per_launch")
        roof["traffic_source"] = src
        roof["profile_stale"] = profile_stale(d)
        per_wave_step = d.get("valu_insts_per_wave_step")
        waves = pmc.get("SQ_WAVES")
        if per_wave_step and waves:
            waves_here = waves * (n_local / float(w["tables"]))
            valu_rate = per_wave_step * waves_here * kern_steps / (ms_launch * 1e-3)
            lanes = d.get("lanes_active")
            # resident waves per SIMD: the launch's waves, capped by what the kernel's registers allow (rocprofv3's VGPR_Count
            # is half the allocation; 512 registers per lane and SIMD: guides/MI355X_MICROARCH.md, register files)
            resident = min(waves / 1024.0, float(max(1, min(8, 512 // (2 * d["vgpr"]))))) if d.get("vgpr") else waves / 1024.0
            ceil_rate, ceil_cyc = mix_ceiling(resident)
            roof.update({"achieved": valu_rate, "frac": valu_rate / VALU_PEAK_WAVE_INSTS_PER_S,
                         "valu_insts_per_wave_step": per_wave_step, "salu_insts_per_wave_step": d.get("salu_insts_per_wave_step"),
                         "lanes_active": lanes, "waves_per_simd": resident, "waves_per_simd_launched": waves / 1024.0,
                         "wait_any_frac": d.get("wait_any_frac_of_wave_cycles"), "source": src,
                         "ceiling_mix": {"peak": ceil_rate, "frac_of_ceiling": valu_rate / ceil_rate,
                                         "cycles_per_instruction": ceil_cyc, "source": "profiles/r05_valu_rates.txt",
                                         "half_rate_share": half_rate_share()[0], "half_rate_share_source": half_rate_share()[1],
                                         "note": "what this instruction mix can issue at this occupancy: measured cycles per "
                                                 "wave-instruction of the plain and of the half-rate kinds, weighted by their "
                                                 "share of the kernel's VALU instructions (ISSUE_CYCLES / mix_ceiling in bench.py)"},
                         "note": "HYBRID figure: achieved = wave-level VALU instructions per wave-step from the COMMITTED rocprofv3 PMC "
                                 "pass of this workload (`source`: SQ_INSTS_VALU / waves / steps; valid while kernel and profile "
                                 "stay in step) x waves x steps per launch / THIS run's HIP-event launch time; peak = 256 CU x 4 "
                                 "SIMD x 2.4 GHz / 2 cycles per wave64 VALU instruction (guides/MI355X_MICROARCH.md)"})
    if roof["achieved"] is None:   # no committed PMC summary for this shape: only the SURVEY 8d figure can be given
        roof.update({"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "note": "no rocprofv3 PMC summary of this workload under profiles/: algorithmic-HBM figure only (see hbm_algorithmic)"})
    return roof


def kernel_short(w, players):
    if not w["fused"]:
        return "k_rollout_single<%d>, 1 step/launch" % players
    occ3 = (players == 6 and w["n_local"] > 131072) or (players == 7 and w["n_local"] >= 262144) or (players == 8 and w["n_local"] >= 524288)
    name = {"random": "k_rollout_occ3" if occ3 else "k_rollout", "allin": "k_rollout_occ3_allin" if occ3 else "k_rollout_allin"}[w["policy"]]
    tab_min = int(os.environ.get("PK_ROLLOUT_TAB", "16"))     # pk_api.hip launch_rollout: the table-evaluator variant, where it applies
    if w["n_local"] <= 65536 and tab_min > 0 and w["stats"]["min"] >= tab_min and not occ3:
        if w["policy"] == "random" and players <= 6:
            name = "k_rollout_tab"
        elif w["policy"] == "allin" and players <= 10:
            name = "k_rollout_allin_tab"
    return "%s<%d> fused, %.0f steps/launch" % (name, players, w["kern_steps"])


def kernel_description(w):
    stats, K, chunk = w["stats"], w["K"], w["chunk"]
    if not w["fused"]:
        return "k_rollout_single (1 step/launch, the state round-trips HBM every step)"
    return ("k_rollout (fused; %d launches in the median sample, mean %.1f steps per launch, min %d, max %d over all samples: "
            "asynchronous calls of %d steps %s)"
            % (w["launches"], w["kern_steps"], stats["min"], stats["max"], min(chunk, K),
               "merged on the host while two launches are in flight (pk_set_coalesce)" if stats["max"] > min(chunk, K) else "launched one by one"))


def _leg(out, name, fn):
    """Runs one secondary leg; a failure becomes an entry with `error` instead of taking the other legs (or the headline) down."""
    try:
        out.append(fn())
    except Exception as e:   # noqa: BLE001
        out.append({"name": name, "error": "%s: %s" % (type(e).__name__, e)})



def step_workload(ctx, device, tables, players, steps=400, warmup=100, replay=False, fused_reset=True, async_hands=0, obs=None, obs_fused=True):
    """`Game.step` with CALLER-SUPPLIED actions on device buffers (reference pokerl/game.py:621-700) as the device-resident loop of
    examples/random_game.py:8-12: pick (pk_pick_actions_d) + step with the reset of a finished game in the same launch
    (pk_step_auto_d) -- two launches per step, the table state round-trips HBM in each; fused_reset=False: pk_step_d + pk_reset_d on
    the step's own flags, three launches.  replay: the actions a first pass recorded are replayed, so that the timed loop is the
    step kernel alone.  async_hands > 0: pk_step_async_d with that budget of hand ends per launch -- a table whose step rolls on through
    further hands stays in flight and the loop acts on the tables that are ready; the value counts DELIVERED steps (ready flags of the timed
    launches).  obs = "packed" / "dense": the loop also produces `game.active_state` (game.py:323-332: the StateView row of the player to act)
    after every step -- obs_fused: written by the step kernel itself from registers (pk_set_step_obs), else by a second launch
    (pk_get_obs_packed_d / pk_get_obs_d) that re-reads the tables.  HIP events on the handle's stream around the timed steps; the serials
    prove the steps were made."""
    import ctypes as C
    import numpy as np
    import pokerl_amd
    from pokerl_amd import _lib as L
    from pokerl_amd.hipmem import DeviceBuffer, DeviceEvent
    T = tables
    g = pokerl_amd.VecGame(T, num_players=players, device=device, table_id_base=ctx.rank * T)
    flags, terr = DeviceBuffer(T, device), DeviceBuffer(T, device)
    total = warmup + steps
    rec = DeviceBuffer(T * 4 * (total if replay else 1), device)
    ready = DeviceBuffer(T * (steps + 1), device) if async_hands > 0 else None     # one slice per timed launch (slice 0: warm-up)
    ev0, ev1 = DeviceEvent(), DeviceEvent()
    assert not (async_hands > 0 and (replay or not fused_reset)), "the asynchronous leg picks on the device and resets inside the launch"
    assert obs in (None, "packed", "dense"), obs
    row_bytes = 0 if obs is None else (16 + 8 * (3 * players + 1)) if obs == "packed" else 8 * (17 + 3 * players)     # PK_OBS_PACKED_BYTES / PK_OBS_DIM
    rows = DeviceBuffer(T * row_bytes, device) if obs else None
    lib = L.lib()
    if obs and obs_fused:
        g.set_step_obs(rows if obs == "dense" else None, rows if obs == "packed" else None)

    def observe():     # the second launch of the unfused form
        if obs and not obs_fused:
            L.check((lib.pk_get_obs_packed_d if obs == "packed" else lib.pk_get_obs_d)(g._h, -1, rows.ptr), g._h)

    def act(s):
        return C.c_void_p(rec.ptr.value + (s * T * 4 if replay else 0))

    def loop(lo, hi, pick):
        for s in range(lo, hi):
            if pick:
                g.pick_actions_d(act(s), 0)
            if async_hands > 0:
                g.step_async_d(act(s), flags, terr, C.c_void_p(ready.ptr.value + (s - warmup + 1 if s >= warmup else 0) * T), async_hands, True)
                observe()
                continue
            g.step_d(act(s), flags, terr, auto_reset=fused_reset)
            if not fused_reset:
                g.reset_d(flags, L.FLAG_GAME_OVER)
            observe()

    def drain():
        if async_hands > 0:
            g.pick_actions_d(act(0), 0)
            g.step_async_d(act(0), flags, terr, ready, 0, True)

    if replay:                       # record the action of every step, then start over on the same RNG streams
        g.reset(); loop(0, total, True); g.sync()
        g.set_serials(0, 0)
    g.reset()
    loop(0, warmup, not replay)
    drain(); g.sync()
    s0 = int(g.step_serial.sum())
    ctx.barrier()
    g.record_event(ev0.handle)
    t0 = time.perf_counter()
    loop(warmup, total, not replay)
    g.record_event(ev1.handle)
    g.sync(); ctx.barrier()
    dt = time.perf_counter() - t0
    dev_ms = DeviceEvent.elapsed_ms(ev0, ev1)
    delivered = None
    if async_hands > 0:     # delivered steps = the ready flags of the timed launches; then finish what is in flight so that the tables can be read
        delivered = int(ready.download(np.uint8, T * steps, T).sum(dtype=np.int64))
        drain(); g.sync()
    made = int(g.step_serial.sum()) - s0
    bad = int((terr.download(np.uint8, T) != 0).sum())
    if obs:            # the rows were really written: they are the getter's rows of the final state
        ref = DeviceBuffer(T * row_bytes, device)
        L.check((lib.pk_get_obs_packed_d if obs == "packed" else lib.pk_get_obs_d)(g._h, -1, ref.ptr), g._h)
        g.sync()
        same = rows.download(np.uint8, T * row_bytes).tobytes() == ref.download(np.uint8, T * row_bytes).tobytes()
        ref.free()
        assert same or async_hands > 0, "the observation rows of the last step differ from the getter's"     # (bounded launches: the drain's extra step has re-written some rows)
        if obs_fused:
            g.set_step_obs(None, None)
    g.close()
    for b in (flags, terr, rec, ready, rows):
        if b is not None:
            b.free()
    if async_hands > 0:     # (every delivered step was made; the drain's pick + step adds up to one more per table)
        assert 0.95 * T * steps <= delivered <= made <= delivered + 2 * T, (delivered, made, T * steps)
    else:
        assert made >= 0.999 * T * steps, (made, T * steps, bad)   # every table made every step (a PK_TERR_NO_WINNER step would not count)
    return dict(tables=T, players=players, steps=steps, warmup=warmup, replay=replay, seconds=dt, device_ms=dev_ms,
                game_steps=delivered if delivered is not None else made, async_hands=async_hands,
                ready_fraction_per_launch=(delivered / float(T * steps)) if delivered is not None else 1.0,
                tables_with_error_bits=bad, fused_reset=fused_reset, obs=obs, obs_fused=bool(obs and obs_fused), obs_row_bytes=row_bytes,
                launches_per_step=(1 if replay else 2) + (0 if fused_reset else 1) + (1 if (obs and not obs_fused) else 0))


def step_profile_summary(tables, players, bounded=False, obs=None, obs_fused=False):
    """The committed rocprofv3 summary of the Game.step loop (profiles/rNN_step_*_summary.json: kernel trace + PMC passes of
    tools/profile_step.sh), latest round; None if none."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_step_*_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("tables"), w.get("players"), bool(w.get("bounded", False)), w.get("obs"), bool(w.get("obs_fused", False))) == (tables, players, bool(bounded), obs, bool(obs_fused)):
            best = (d, os.path.basename(f))
    return best


def step_line(res, name):
    """Bench entry of a Game.step leg: HBM roofline on SURVEY 8d's algorithmic bytes (this path DOES move the table state every step),
    the measured traffic of the committed PMC passes beside it."""
    T, N = res["tables"], res["players"]
    ms_step = res["device_ms"] / res["steps"]
    rate = res["game_steps"] / (res["device_ms"] * 1e-3)
    row = res.get("obs_row_bytes", 0)
    alg = (b_step(N) + row) * res["game_steps"] / res["steps"]   # algorithmic bytes of one step of the whole batch (+ the observation row where the loop produces one)
    gbs = alg / (ms_step * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
            "kernel": " + ".join(([] if res["replay"] else ["k_pick"]) + [("k_step_async" if res.get("async_hands") else "k_step") + ("[+%s row]" % res["obs"] if res.get("obs_fused") else "")] +
                                 ([] if res["fused_reset"] else ["k_reset(masked)"]) + (["k_obs_packed" if res["obs"] == "packed" else "k_obs"] if (res.get("obs") and not res.get("obs_fused")) else [])),
            "obs_row_bytes": row,
            "kernel_ms": ms_step, "launches_timed": res["steps"] * res["launches_per_step"], "algorithmic_bytes_per_step": alg,
            "kernel_ms_source": "HIP events (pk_record_event) around the timed steps on the handle's stream / steps: ALL launches of one "
                                "loop iteration, launch gaps included"}
    prof = step_profile_summary(T, N, bool(res.get("async_hands")), res.get("obs"), bool(res.get("obs_fused")))
    if prof:
        d, src = prof
        mine = [k for k in d.get("loop_kernels", d.get("kernels", {})) if not (res["replay"] and k == "k_pick") and not (res["fused_reset"] and k == "k_reset")]   # the kernels of THIS leg's loop iteration
        roof.update({"traffic": sum(d["kernels"][k].get("hbm_traffic_bytes_per_launch", 0.0) for k in mine), "source": src, "profile_stale": profile_stale(d),
                     "k_step_ms_rocprof": d.get("k_step_avg_ms"), "k_step_ms_rocprof_min_median_max": d.get("k_step_min_median_max_ms"), "kernels_in_traffic": mine,
                     "traffic_unit": "HBM bytes per loop iteration (rocprofv3 PMC: 2*FETCH_SIZE + WRITE_SIZE, KiB), summed over its kernels"})
    return {"name": name, "metric": "Game.step env-steps/s (caller-supplied actions, device-resident)", "value": rate, "unit": "env-steps/s",
            "ms_per_step": res["seconds"] / res["steps"] * 1e3, "kernel": roof["kernel"], "kernel_ms": ms_step,
            "launches": res["steps"] * res["launches_per_step"], "device_ms": res["device_ms"], "seconds": res["seconds"],
            "tables_with_error_bits": res["tables_with_error_bits"], "ready_fraction_per_launch": res.get("ready_fraction_per_launch", 1.0), "roofline": roof}


# ---------------------------------------------------------------------------------------------- the ONE stdout line
LINE_LIMIT = 4096        # the driver keeps the last ~8 KB of stdout: the line must fit with room to spare
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")   # everything else: notes, per-sample arrays, the full roofline blocks


def _sig(x, digits=5):
    """Floats to `digits` significant digits (the line is for a parser; the exact values are in the detail file)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x == 0.0 or x != x or x in (float("inf"), float("-inf")):
        return x
    return float("%.*g" % (digits, x))


def _short_roofline(rf, with_alg=True):
    if not isinstance(rf, dict):
        return None
    out = {k: _sig(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "steps_per_launch")
           if k in rf}
    if rf.get("source"):
        out["source"] = rf["source"]
    if "profile_stale" in rf:
        out["profile_stale"] = bool(rf["profile_stale"])
    hb = rf.get("hbm_algorithmic")
    if with_alg and isinstance(hb, dict):
        out["hbm_algorithmic"] = {"frac": _sig(hb.get("frac"))}
    return out


def _short_leg(x):
    if "error" in x:
        return {"name": str(x.get("name", ""))[:40], "error": str(x["error"])[:80]}
    rf = x.get("roofline") or {}
    hb = rf.get("hbm") if isinstance(rf.get("hbm"), dict) else rf.get("hbm_algorithmic") if isinstance(rf.get("hbm_algorithmic"), dict) else None
    hbm_frac = rf.get("frac") if rf.get("bound") == "hbm" else (hb or {}).get("frac")
    out = {"name": str(x.get("short") or x.get("name", ""))[:40], "value": _sig(x.get("value")), "unit": x.get("unit"),
           "kernel_ms": _sig(x.get("kernel_ms")), "bound": rf.get("bound"), "frac": _sig(rf.get("frac")), "hbm_frac": _sig(hbm_frac)}
    if x.get("hand_evals_per_s") and x.get("showdown_heavy"):
        out["hand_evals_per_s"] = _sig(x["hand_evals_per_s"])
    if rf.get("profile_stale"):          # (only when true: a leg without the key rests on a summary of the running library's sources, or on none)
        out["profile_stale"] = True
    return out


def compact_line(full):
    """The driver-facing line (<= LINE_LIMIT bytes) out of the full result: the contract's keys, `roofline`, `cpu_baseline`, the
    evaluator and one short entry per extra leg.  Prose and arrays stay in DETAIL_FILE."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "reps", "samples", "timed_steps_per_table", "timed_s", "lib")
    line = {k: _sig(full[k]) for k in keep if k in full}     # (the self-check keys are absent from results of earlier rounds)
    c = full["config"]
    line["config"] = {"workload": c["workload"], "tables_per_gpu": c["tables_per_gpu"], "num_players": c["num_players"],
                      "policy": c["policy"], "kernel": c.get("kernel_short") or str(c.get("kernel", ""))[:60],
                      "launch_stats": c["launch_stats"], "parallelism": c["parallelism"]}
    line["hand_evals_per_s"] = _sig(full.get("hand_evals_per_s"))
    line["roofline"] = _short_roofline(full["roofline"])
    if "evaluator" in full:
        ev = full["evaluator"]
        line["evaluator"] = {"hand_evals_per_s": _sig(ev["hand_evals_per_s"]), "frac": _sig(ev["roofline"]["frac"]),
                             "bound": "hbm", "kernel_ms": _sig(ev["kernel_ms"])}
        if ev["roofline"].get("profile_stale"):
            line["evaluator"]["profile_stale"] = True
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                "sample": cb["sample"][:120],
                                "all_cores": {"value": _sig(cb["all_cores"]["value"]), "cores": cb["all_cores"]["cores"]}}
    if "extra_workloads" in full:
        line["extra_workloads"] = [_short_leg(x) for x in full["extra_workloads"]]
    if "dist" in full:
        line["dist"] = full["dist"]
    line["detail"] = os.path.basename(DETAIL_FILE)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) >= LINE_LIMIT and "extra_workloads" in line:       # never observed; the contract keys come first
        line["extra_workloads"] = [{"name": x["name"], "value": x.get("value"), "frac": x.get("frac")} for x in line["extra_workloads"]]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) < LINE_LIMIT, len(text)
    return text


def extra_workloads(ctx, device):
    """The other single-GPU BASELINE configs and the PokerGameEnv path, as SHORT driver-timed legs after the headline leg
    (~1 s of GPU work each, same event-bracketed timing, same counter assertion): configs[1], configs[4] (showdown-heavy:
    1.000 in-game evaluation per env-step), PokerGameEnv.step synchronous and asynchronous."""
    out = []

    def rollout_leg(name, short, tables, players, policy, **kw):
        def run():
            w = rollout_workload(ctx, device, tables, players, policy, **kw)
            return {"name": name, "short": short, "showdown_heavy": policy == "allin", "metric": "env-steps/s", "value": w["total_steps"] / w["seconds"], "unit": "env-steps/s",
                    "hand_evals_per_s": w["hand_evals_per_s"], "hands_per_s": w["hands_per_s"],
                    "ms_per_step": w["seconds"] / (w["K"] * w["reps"]) * 1e3, "steps_per_sample": w["K"] * w["reps"], "samples": len(w["sample_s"]),
                    "kernel": kernel_description(w), "kernel_ms": w["ms_launch"], "launches": w["launches"],
                    "launch_stats": w["stats"], "roofline": rollout_roofline(w, ctx.world)}
        _leg(out, name, run)

    def env_leg(name, short, **kw):
        def run():
            res = env_workload(ctx, device, **kw)
            line = env_line(res, ctx)
            return {"name": name, "short": short, "metric": "PokerGameEnv.step/s (delivered)", "value": line["value"], "unit": "env.step/s",
                    "game_steps_per_s": line["game_steps_per_s"], "game_steps_per_env_step": line["game_steps_per_env_step"],
                    "ready_fraction_per_launch": line["ready_fraction_per_launch"], "kernel": line["roofline"]["kernel"],
                    "kernel_ms": line["roofline"]["kernel_ms"], "launches": res["launches"], "device_ms": res["device_ms"],
                    "seconds": res["seconds"], "roofline": line["roofline"]}
        _leg(out, name, run)

    for cfg, (tables, players, policy) in ((1, (4096, 2, "random")), (4, (65536, 9, "allin"))):
        rollout_leg("BASELINE configs[%d]: %d tables x %d seats, %s agents" % (cfg, tables, players, policy),
                    "cfg%d %dx%d %s" % (cfg, tables, players, policy), tables, players, policy,
                    K=4096, warmup=512, min_steps=262144, samples=3)
    # the driver's own call pattern WITHOUT host-side merging: what a caller that observes the tables between its 20-step calls
    # gets (every getter flushes), beside the headline figure, which holds for a caller that does not
    rollout_leg("BASELINE configs[2] in 20-step calls, one launch per call (pk_set_coalesce(0)): the rate of a caller that "
                "observes the tables between calls", "cfg2 20-step calls, 1 launch/call", 65536, 6, "random", K=20, warmup=5,
                min_steps=131072, samples=3, coalesce=0)
    # Game.step itself (row a7): the caller's actions on device buffers, the state round-trips HBM every step
    _leg(out, "Game.step loop", lambda: step_line(step_workload(ctx, device, 65536, 6, steps=2000, warmup=200),
                                                  "Game.step, device-resident loop pk_pick_actions_d + pk_step_auto_d (game-over reset in the step's launch), 65 536 x 6")
         | {"short": "Game.step pick+step_auto_d 65536x6"})
    _leg(out, "Game.step replay", lambda: step_line(step_workload(ctx, device, 65536, 6, steps=2000, warmup=200, replay=True),
                                                    "Game.step, pk_step_auto_d alone on pre-picked (replayed) actions, 65 536 x 6")
         | {"short": "Game.step step_auto_d replay 65536x6"})
    # ... with `game.active_state` after every step (SURVEY f2 on this path): the packed StateView row of the player to act, by a second launch
    # that re-reads the tables (pk_get_obs_packed_d) and from the step kernel's registers (pk_set_step_obs)
    _leg(out, "Game.step + obs, 2 launches", lambda: step_line(step_workload(ctx, device, 65536, 6, steps=2000, warmup=200, replay=True, obs="packed", obs_fused=False),
                                                               "Game.step + packed observation row: pk_step_auto_d, then pk_get_obs_packed_d (two launches), replayed actions, 65 536 x 6")
         | {"short": "Game.step+obs_packed 2 launches 65536x6"})
    _leg(out, "Game.step + obs, fused", lambda: step_line(step_workload(ctx, device, 65536, 6, steps=2000, warmup=200, replay=True, obs="packed", obs_fused=True),
                                                          "Game.step + packed observation row written by the step kernel (pk_set_step_obs; one launch), replayed actions, 65 536 x 6")
         | {"short": "Game.step+obs_packed fused 65536x6"})
    # ... as bounded launches (pk_step_async_d): the few tables whose step rolls on through further hands stay in flight instead of
    # holding the launch; the loop acts on the tables that are ready, the value counts delivered steps
    _leg(out, "Game.step async", lambda: step_line(step_workload(ctx, device, 65536, 6, steps=2000, warmup=200, async_hands=1),
                                                   "Game.step, bounded launches: pk_pick_actions_d + pk_step_async_d (one hand end per launch, reset inside), 65 536 x 6")
         | {"short": "Game.step pick+step_async_d 65536x6"})
    # ... and at a batch that fills the chip: there the bytes, not the slowest table's serial chain, bound the launch
    _leg(out, "Game.step 1M", lambda: step_line(step_workload(ctx, device, 1048576, 6, steps=300, warmup=50, replay=True),
                                                "Game.step, pk_step_auto_d alone on pre-picked (replayed) actions, 1 048 576 x 6")
         | {"short": "Game.step step_auto_d replay 1048576x6"})
    env_leg("PokerGameEnv.step synchronous (pk_env_step_fused_d), 65 536 x 6", "env.step sync 65536x6", tables=65536, players=6, steps=1000, warmup=50)
    env_leg("PokerGameEnv.step asynchronous (pk_env_step_async_d, 8 betting passes per launch), one handle of 65 536 x 6",
            "env.step async8 65536x6", tables=65536, players=6, async_passes=8, steps=4000, warmup=300)
    env_leg("PokerGameEnv.step asynchronous, ONE handle of 524 288 x 6 in three sub-batches (pk_set_env_batches)",
            "env.step async8 524288x6 inner3", tables=524288, players=6, async_passes=8, inner=3, steps=1500, warmup=300)
    return out


def pk_lib_info():
    from pokerl_amd import _lib as L
    return L.lib().pk_build_info().decode()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4096, help="K: Game.step()s per table in one timed block")
    ap.add_argument("--warmup", type=int, default=512)
    ap.add_argument("--tables", type=int, default=65536, help="tables per GPU (weak scaling)")
    ap.add_argument("--players", type=int, default=6)
    ap.add_argument("--policy", choices=["random", "allin"], default="random")
    ap.add_argument("--chunk", type=int, default=4096, help="at most this many steps per fused launch")
    ap.add_argument("--reps", type=int, default=0,
                    help="blocks of --steps per timed sample (0 = as many as make a sample >= --min-steps steps, so that a "
                         "short --steps is not one launch-latency sample)")
    ap.add_argument("--min-steps", type=int, default=524288,
                    help="a timed sample runs at least this many steps per table (~1.1 s of GPU work at 65 536 x 6: seven "
                         "samples keep the GPU busy for ~8 s of a ~16 s run)")
    ap.add_argument("--samples", type=int, default=7, help="timed samples; the MEDIAN is reported")
    ap.add_argument("--unfused", action="store_true", help="one launch per step (state round-trips HBM every step)")
    ap.add_argument("--mode", choices=["game", "env", "step"], default="game")
    ap.add_argument("--step-replay", action="store_true", help="--mode step: the step kernel alone on replayed (pre-picked) actions")
    ap.add_argument("--step-unfused-reset", action="store_true", help="--mode step: pk_step_d + pk_reset_d(flags) instead of pk_step_auto_d")
    ap.add_argument("--step-async", type=int, default=0, metavar="HANDS", help="--mode step through pk_step_async_d with this budget of hand ends per launch")
    ap.add_argument("--step-obs", choices=["packed", "dense"], default=None, help="--mode step: the loop also produces the StateView row of the player to act after every step")
    ap.add_argument("--step-obs-separate", action="store_true", help="--mode step --step-obs: by a second launch (pk_get_obs(_packed)_d) instead of from the step kernel (pk_set_step_obs)")
    ap.add_argument("--env-batches", type=int, default=1, help="--mode env: independent batches in flight, one stream each")
    ap.add_argument("--env-async", type=int, default=0, metavar="PASSES",
                    help="--mode env through pk_env_step_async_d with this pass budget per launch (0: synchronous)")
    ap.add_argument("--env-unfused", action="store_true", help="--mode env with separate pick / step / reset / obs launches")
    ap.add_argument("--env-inner-batches", type=int, default=1,
                    help="--mode env --env-async: sub-batches INSIDE each handle (pk_set_env_batches): a call launches one "
                         "range of the handle's tables and delivers the range launched longest ago")
    ap.add_argument("--full-line", action="store_true",
                    help="print the FULL result (what bench_detail.json holds) on stdout instead of the compact line: for the profiling "
                         "tools under tools/, never for the driver (it keeps only the last ~8 KB of stdout)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-evaluator", action="store_true", help="skip the stand-alone evaluator kernel leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_workloads legs (the other BASELINE configs, PokerGameEnv)")
    ap.add_argument("--coalesce", type=int, default=-1,
                    help="host-side merging of asynchronous rollout calls into launches of up to this many steps "
                         "(pk_set_coalesce; -1: the library default 1024, 0: one launch per call)")
    args = ap.parse_args()

    ctx = DistContext()
    if ctx.world != max(1, args.gpus) and ctx.rank == 0:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, ctx.world), file=sys.stderr)
    import pokerl_amd  # noqa: F401  (fails loudly here if the HIP library is missing)
    device = 0 if os.environ.get("PK_BENCH_SAME_DEVICE") else ctx.local_rank  # rehearsal knob: all ranks on GPU 0
    if args.mode == "env":
        env_mode(args, ctx, device)
        ctx.close()
        return
    if args.mode == "step":   # Game.step with caller-supplied actions as its own line (tools/profile_step.sh profiles this command)
        res = step_workload(ctx, device, args.tables, args.players, args.steps, args.warmup, args.step_replay, not args.step_unfused_reset, args.step_async,
                            args.step_obs, not args.step_obs_separate)
        if ctx.rank == 0:
            print(json.dumps(step_line(res, "Game.step device-resident loop, %d x %d%s%s%s" % (args.tables, args.players, ", replayed actions" if args.step_replay else "",
                                                                                                ", bounded launches (%d hand end(s) each)" % args.step_async if args.step_async else "",
                                                                                                "" if not args.step_obs else ", %s observation rows %s" % (args.step_obs, "by a second launch" if args.step_obs_separate else "from the step kernel")))))
        ctx.close()
        return
    policy = 0 if args.policy == "random" else 1
    fused = not args.unfused
    w = rollout_workload(ctx, device, args.tables, args.players, args.policy, args.steps, args.warmup, args.chunk,
                         args.reps, args.min_steps, args.samples, fused, args.coalesce)
    K, reps = w["K"], w["reps"]
    dist_block = ctx.describe(device, w["n_local"])      # (collective: every rank takes part)
    ctx.barrier()
    if ctx.rank == 0:
        cfg_idx = baseline_config_index(args.tables, args.players, args.policy, ctx.world)
        out = {
            "metric": "env-steps/sec (whole node) + showdown hand-evals/sec, %s tables %d-max" % ("{:,}".format(args.tables).replace(",", " "), args.players),
            "value": w["total_steps"] / w["seconds"], "unit": "env-steps/s", "n_gpus": ctx.world, "steps": K,
            "warmup": args.warmup, "ms_per_step": w["seconds"] / (K * reps) * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "reps": reps, "samples": len(w["sample_s"]), "sample_seconds": w["sample_s"], "sample_device_ms": w["sample_dev_ms"],
            # self-check: ms_per_step (of the median sample) x timed_steps_per_table ~ timed_s (all samples) <= the run's wall clock
            "timed_steps_per_table": K * reps * len(w["sample_s"]), "timed_s": sum(w["sample_s"]), "lib": pk_lib_info(),
            "timing": "each sample = %d block(s) of %d steps per table, launched back to back and completed by a sync inside "
                      "the timed region; value = steps of one sample / MEDIAN sample time (MAX over ranks per sample)" % (reps, K),
            "config": {"workload": "%d tables/GPU x %d GPU(s), num_players=%d, %s agents in-kernel (Philox4x32-10), "
                                   "start_credits=100 blinds 1/2, auto-reset; %s"
                                   % (args.tables, ctx.world, args.players, args.policy,
                                      "BASELINE configs[%d]" % cfg_idx if cfg_idx is not None else "not a BASELINE config"),
                       "tables_per_gpu": args.tables, "num_players": args.players, "policy": args.policy,
                       "kernel": kernel_description(w), "kernel_short": kernel_short(w, args.players), "launch_stats": w["stats"], "launch_stats_before_timed_region": w["stats_warm"],
                       "parallelism": "env-parallel, %d shard(s), no collective on the step path" % ctx.world},
            "hand_evals_per_s": w["hand_evals_per_s"], "hands_per_s": w["hands_per_s"], "games_per_s": w["games_per_s"],
            "roofline": rollout_roofline(w, ctx.world),
        }
        if not args.no_evaluator:
            out["evaluator"] = evaluator_leg(device)
        if not args.no_extra and ctx.world == 1 and fused:
            try:    # the headline line must survive whatever happens to a secondary leg
                out["extra_workloads"] = extra_workloads(ctx, device)
            except Exception as e:   # noqa: BLE001
                out["extra_workloads"] = [{"name": "extra_workloads", "error": "%s: %s" % (type(e).__name__, e)}]
        if not args.no_cpu_baseline and ctx.world == 1:
            out["cpu_baseline"] = cpu_baseline(args.players, policy)
        if dist_block is not None:
            out["dist"] = dist_block
        with open(DETAIL_FILE, "w") as f:      # everything: notes, per-sample arrays, the full roofline block of every leg
            json.dump(out, f, indent=1)
        print(json.dumps(out) if args.full_line else compact_line(out))   # ONE line of < 4 KB (the driver keeps the last ~8 KB of stdout)
    ctx.close()


if __name__ == "__main__":
    main()
