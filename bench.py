#!/usr/bin/env python3
"""bench.py -- participant-steps/s (physics + collision + status) of the batched env.step().

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over every participant of the rank's pool: the fused step kernel (integrate +
collide + status epilogue) including the device-side auto-reset of finished envs, and, for N > 1, the asynchronous
RCCL all-gather of the 8-byte per-env result records.  Inputs (state, actions, geometry) are resident in HBM before
the timed region starts.  Weak scaling: every rank owns --envs environments (default 4096 x 64 participants, the
metric workload).

How the K timed steps are enqueued (--mode):
  chain (default)  t2d_step_n: a rollout fragment of up to 32 steps is ONE launch -- workgroup (g, k) takes step k of the
                   envs of workgroup g and is ordered after (g, k - 1) through a word in device memory, so no launch
                   boundary separates two steps; every step does the full work and every per-step result equals that of
                   K separate launches (tests/test_gpu_chain.py).
  step             one t2d_step launch per step (rounds 1-2's headline).
The line reports the other mode too (`alternates`), so both are driver-timed.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the step kernel is bound by fp64 VALU
issue, not by HBM: the fraction of the VALU issue roof -- instructions per step from the committed rocprofv3 SQ pass of
the same sources, checked by a source hash, over the step duration measured here with HIP events -- and the HBM figure
beside it), `configs` (BASELINE.json's other configurations), `next_rows` (IDM, lidar, the ParkingEnv vector step),
`gather` (N > 1) and `cpu_baseline` (the C oracle -- a port of the reference's algorithm -- on the host cores).
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; streams sharing a queue serialise.  Env groups
# (--groups) need one queue each next to torch's own streams -- must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue roof: a SIMD issues one wave64 VALU instruction (fp64 included: full rate on CDNA4) per 4 cycles;
# 256 CUs x 4 SIMDs at the 2.4 GHz peak clock of the same guide
VALU_ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 4.0
EFFECTIVE_CLOCK_GHZ = 2.26     # shader clock inside the chained step kernel (s_memtime / s_memrealtime; profiles/r06_chain_timing_frag20.json)
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9   # SIMD-cycles per second at the 2.4 GHz peak clock
INTEGRATOR_BYTES = 44          # algorithmic bytes per participant-step
COLLIDE_BYTES = 20             # + per-env geometry (computed from the scene)
ACTION_SETS = 32               # action ring resident in HBM (one set per step of a chained fragment)
DEFAULTS = {"metric": (4096, 64), "cfg5": (1024, 64), "cfg2": (4096, 1), "cfg3": (1024, 64), "cfg4": (512, 32)}


def build_scene(name, n_env, agents, seed):
    from tactics2d_amd import scenarios as S
    if name == "metric" or name == "cfg5":
        return S.mixed(n_env, agents, seed=3 + 1000 * seed)
    if name == "cfg2":
        return S.parking(n_env, seed0=seed * n_env)
    if name == "cfg3":
        return S.highway(n_env, agents, seed=1 + 1000 * seed)
    if name == "cfg4":
        return S.intersection(n_env, agents, seed=2 + 1000 * seed)
    raise SystemExit(f"unknown config {name}")


# ------------------------------------------------------------------------------------------------- CPU baseline
def _cpu_leg(scene, n_env, threads, target_seconds, min_steps=1):
    """Step the first n_env envs of the scene with the oracle on `threads` host threads:
    integrate -> fp32 store -> collide -> status, per step.  Returns (participant-steps/s, steps, s)."""
    from oracle import oracle as O
    A = scene.A
    n = n_env * A
    sl = slice(0, n)

    def cut(csr):
        if csr is None:
            return None
        eo, vo, xy = csr
        p1 = eo[n_env]
        return eo[:n_env + 1].copy(), vo[:p1 + 1].copy(), xy[:vo[p1]].copy()

    static, lanes = cut(scene.static), cut(scene.lanes)
    boundary = None if scene.boundary is None else scene.boundary[:n_env]
    x, y, h, v = (a[sl].copy() for a in (scene.x, scene.y, scene.heading, scene.speed))
    vx = (v.astype(np.float64) * np.cos(h.astype(np.float64))).astype(np.float32)
    vy = (v.astype(np.float64) * np.sin(h.astype(np.float64))).astype(np.float32)
    tid, act = scene.type_id[sl], scene.active[sl]
    cfg = O.make_config(**scene.status)
    ep = O.EpisodeState(n_env, None if scene.target is None else scene.target[:n_env], None,
                        np.stack([scene.x[sl].reshape(n_env, A)[:, 0], scene.y[sl].reshape(n_env, A)[:, 0]], 1))
    cnt = np.zeros(n_env, np.int32); frame = np.zeros(n_env, np.int32)
    rng = np.random.default_rng(123)
    full_a0, full_a1 = scene.sample_actions(rng)
    a0, a1 = full_a0[sl], full_a1[sl]
    is_dyn = scene.rows[tid, 0] == 1
    steps = 0
    O.set_threads(threads)
    try:
        t0 = time.perf_counter()
        while True:
            o = O.integrate(scene.rows, x, y, h, v, vx, vy, a0, a1, tid, act, scene.interval_ms)
            x, y, h, v = (np.float32(o[:, k]) for k in range(4))
            vx = np.where(is_dyn, vx, np.float32(o[:, 4])); vy = np.where(is_dyn, vy, np.float32(o[:, 5]))
            f, _ = O.collide(scene.rows, n_env, A, x, y, h, tid, act, static, boundary, None, lanes, 0)
            O.status_ex(cfg, A, f, scene.interval_ms, cnt, frame, scene.rows, x, y, h, tid, ep)
            steps += 1
            el = time.perf_counter() - t0
            if (el >= target_seconds and steps >= min_steps) or steps >= 2000:
                break
    finally:
        O.set_threads(1)
    return n * steps / el, steps, el


def cpu_baseline(scene, target_seconds=10.0):
    """The oracle (C port of the reference algorithm, fp64 scalar) on a bounded sample of the SAME scene, SURVEY.md 8(d):
    (1) one core, first 96 envs; (2) the same code with its batch loops spread over the host cores (OpenMP), whole scene.
    `value` is the all-cores rate.  The thread count is chosen from a trial of >= 5 steps per candidate (after one
    untimed step that wakes the thread team); the candidates always include the cgroup CPU quota and twice it, and the
    whole trial table is reported."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # before libgomp initialises: idle threads sleep
    from oracle import oracle as O
    O.build()
    A = scene.A
    one, steps1, el1 = _cpu_leg(scene, min(scene.n_env, 96), 1, target_seconds)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    out = dict(value=one, unit="participant-steps/s", cores=1, kind="port",
               sample=f"first {min(scene.n_env, 96)} envs x {A} participants of the same scene, {steps1} steps, "
                      f"{el1:.1f} s on 1 core; C oracle oracle/t2d_oracle.c (fp64 scalar restatement "
                      f"of the reference)")
    try:   # the reference's own call pattern: one Python-level step per participant, numpy scalar ufuncs
        from oracle import py_loop
        kin = scene.rows[scene.rows[:, 0] == 0]
        if len(kin):
            out["python_loop_value"] = py_loop.time_loop(kin[0], 2000)
            out["python_loop_note"] = ("reference-style per-participant Python loop (SingleTrackKinematics.step restated "
                                       "with numpy scalar calls, physics only), 2000 steps on 1 core")
    except Exception as e:  # reported, never fatal for the bench line
        out["python_loop_note"] = f"not run: {e}"
    if O.has_openmp() and cores > 1:
        # how many threads actually help is a property of the box (cgroup CPU quotas are invisible to sched_getaffinity)
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                quota = max(1, int(round(int(q) / int(per))))
        except Exception:
            pass
        cand = {c for c in (4, 8, 16, 32, 64, 128, cores) if c <= cores}
        if quota:
            cand |= {min(cores, quota), min(cores, 2 * quota)}
        trial = {}
        for c in sorted(cand):
            _cpu_leg(scene, scene.n_env, c, 0.0)                       # untimed: the thread team of this size wakes up
            trial[c] = _cpu_leg(scene, scene.n_env, c, 0.0, min_steps=5)[0]
        # the reported run uses as many threads as the cgroup lets run at once (round 4 picked the winner of the 5-step trial:
        # 32 threads under a quota of 16, whose 10 s run was then throttled to half its trial rate); no quota: the trial's best
        best = min(cores, quota) if quota else max(trial, key=trial.get)
        allc, stepsN, elN = _cpu_leg(scene, scene.n_env, best, target_seconds)
        out.update(value=allc, cores=best, one_core_value=one, cgroup_cpu_quota=quota, logical_cpus=cores,
                   thread_trials={str(c): v for c, v in sorted(trial.items())},
                   sample=f"all {scene.n_env} envs x {A} participants of the same scene, {stepsN} steps, "
                          f"{elN:.1f} s on {best} OpenMP threads over envs (= the cgroup CPU quota where there is one, else the best "
                          f"of the trial table; the table's entries are 5-step bursts and are not sustained rates; {cores} logical "
                          f"CPUs visible, cgroup quota {quota}); one_core_value: first "
                          f"{min(scene.n_env, 96)} envs, {steps1} steps, {el1:.1f} s on 1 core; C oracle "
                          f"oracle/t2d_oracle.c (fp64 scalar restatement of the reference)")
    return out


# ------------------------------------------------------------------------------------------------- GPU helpers
class Runner:
    """One pool + a resident action ring + the two ways of enqueuing steps."""

    def __init__(self, scene, dev, variant, auto_reset=True, outputs="all", seed=5, idm=False):
        import torch
        from tactics2d_amd import layout as L
        from tactics2d_amd.pool import ParticipantPool
        self.scene, self.dev = scene, dev
        self.pool = ParticipantPool(scene.n_env, scene.A, dev.index)
        scene.load(self.pool)
        self.pool.set_integrator_variant(variant)
        if auto_reset:
            self.pool.set_auto_reset(True)   # finished envs restart inside the step launch (no extra kernels)
        if outputs == "state":
            self.pool.set_outputs(velocity=False, applied=False)
        if idm:
            from tactics2d_amd.controller import IDMController, install
            cid = np.full((scene.n_env, scene.A), L.IDM_NONE, np.uint8)
            veh = (scene.rows[scene.type_id, L.P_MODEL] != L.MODEL_POINTMASS).reshape(scene.n_env, scene.A)
            cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
            install(self.pool, [IDMController(desired_speed=25.0, horizon=120.0)], cid.reshape(-1))
        rng = np.random.default_rng(seed)
        sets = [scene.sample_actions(rng) for _ in range(ACTION_SETS)]
        self.a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()   # [sets][N]
        self.a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
        self.stream = torch.cuda.Stream(device=dev)
        self.k = 0
        self.align = False
        self.ring_bound = False    # the whole action ring is what the pool reads (steps_chain binds it once, steps_single rebinds per step)

    def steps_single(self, n, after_step=None):
        N = self.scene.n
        for _ in range(n):
            s = self.k % ACTION_SETS
            self.pool.bind_actions(self.a0.data_ptr() + 4 * N * s, self.a1.data_ptr() + 4 * N * s)
            self.ring_bound = False
            self.pool.step(self.scene.interval_ms, self.stream.cuda_stream)
            self.k += 1
            if after_step:
                after_step()

    def steps_chain(self, n, frag, after_fragment=None):
        """n steps as t2d_step_n fragments of `frag` steps; step j of a fragment reads action set j of the ring"""
        if not self.ring_bound:   # (bound once, like a caller with a resident action ring would: not a call per fragment)
            self.pool.bind_actions(self.a0.data_ptr(), self.a1.data_ptr(), extent=self.a0.numel())
            self.ring_bound = True
        done = 0
        while done < n:
            f = min(frag, n - done)
            if self.align:   # fragments end where the pool's step count reaches a multiple of `frag` (where a gather is due)
                f = min(frag - self.pool.step_count() % frag, n - done)
            self.pool.step_n(f, self.scene.interval_ms, self.scene.n, self.stream.cuda_stream)
            done += f
            self.k += f
            if after_fragment:
                after_fragment()

    def run(self, mode, n, frag, hook=None):
        if mode == "chain":
            self.steps_chain(n, frag, hook)
        else:
            self.steps_single(n, hook)

    def close(self):
        self.pool.close()


def integrator_per_model(variant, n_env=65536, A=64):
    """north_star's roofline kernel at a size where it STREAMS (4 M participants; the metric's 262 144 are one wave round: launch
    ramp, not bandwidth): t2d_integrate per physics model, us per launch and the fraction of the 8 TB/s peak on SURVEY 8(d)'s 44-B
    figure.  Pools of >= 2 M participants without a dynamics participant take four participants per lane."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    rows, _ = S.full_type_table()
    models = rows[:, L.P_MODEL].astype(int)
    n = n_env * A
    out = {}
    for label, model in (("kinematics", L.MODEL_KINEMATICS), ("dynamics", L.MODEL_DYNAMICS), ("pointmass", L.MODEL_POINTMASS)):
        ids = np.nonzero(models == model)[0]
        rng = np.random.default_rng(1)
        tid = ids[rng.integers(0, ids.size, n)].astype(np.uint8)
        pool = ParticipantPool(n_env, A)
        pool.set_param_table(rows)
        pool.set_integrator_variant(variant)
        pool.reset(np.float32(rng.uniform(-100, 100, n)), np.float32(rng.uniform(-100, 100, n)), np.float32(rng.uniform(0, 6.28, n)),
                   np.float32(rng.uniform(0.5, 1.4, n) if model == L.MODEL_POINTMASS else rng.uniform(2.0, 7.5, n)), tid)
        pool.set_actions(np.float32(rng.uniform(-1.0, 1.0, n)), np.float32(rng.uniform(-0.08, 0.08, n)))
        pool.snapshot()
        for _ in range(3):
            pool.integrate(100)
        pool.profile_enable(True)
        for _ in range(12):
            pool.restore()
            pool.integrate(100)
        ms, launches = pool.profile_read(0)
        us = 1e3 * ms / launches
        out[label] = dict(participants=n, avg_us=us, achieved_gbs=INTEGRATOR_BYTES * n / (us * 1e-6) / 1e9,
                          frac=INTEGRATOR_BYTES * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        pool.close()
    out["note"] = ("stand-alone t2d_integrate on pools of one model, every launch from the same snapshot, HIP events around each launch "
                   "(~2 us of the pair included); 44 algorithmic bytes per participant -- the kernel really moves 56-60")
    return out


def power_clock_under_load(run, mode, frag, seconds=0.6):
    """Socket power and shader clock the SMU reports WHILE the headline launches run (amdsmi's metrics table, sampled by a
    thread; outside `value`): whether the step's time is set by a power budget.  None when amdsmi is not importable."""
    import threading
    import torch
    try:
        import amdsmi
        amdsmi.amdsmi_init()
        h = amdsmi.amdsmi_get_processor_handles()[torch.cuda.current_device()]
        cap = amdsmi.amdsmi_get_power_cap_info(h).get("power_cap")
    except Exception as exc:   # noqa: BLE001
        return dict(error=f"amdsmi unavailable: {exc}")
    rows, stop = [], [False]

    def loop():
        while not stop[0]:
            try:
                m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 60000]
                rows.append((m.get("current_socket_power"), float(np.mean(clk)) if clk else None, m.get("ppt_residency_acc")))
            except Exception:   # noqa: BLE001
                break
    th = threading.Thread(target=loop, daemon=True)
    t0 = time.perf_counter()
    run.run(mode, 64, frag)
    torch.cuda.synchronize()
    th.start()
    n = 0
    while time.perf_counter() - t0 < seconds:
        run.run(mode, 10 * frag, frag)
        n += 1
        if n % 4 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    tail = rows[len(rows) // 3:]
    pw = [r[0] for r in tail if isinstance(r[0], (int, float)) and r[0] < 60000]
    ck = [r[1] for r in tail if r[1]]
    ppt = [r[2] for r in rows if isinstance(r[2], (int, float))]
    if not pw or not ck:
        return dict(error="the SMU's metrics table held no power / clock sample", samples=len(rows))
    return dict(power_w=float(np.mean(pw)), power_cap_w=(cap / 1e6 if isinstance(cap, (int, float)) and cap > 1e5 else cap),
                clock_mhz_under_load=float(np.mean(ck)), ppt_residency_delta=(ppt[-1] - ppt[0] if len(ppt) > 1 else None),
                samples=len(tail), seconds=seconds,
                note="SMU metrics table (amdsmi) sampled while the headline launches run, outside `value`; current_gfxclks averaged "
                     "over the XCDs.  The EFFECTIVE shader clock (s_memtime ticks per 100 MHz tick inside the kernel, = GRBM_GUI_ACTIVE "
                     "/ duration) is ~5 % below what the SMU reports: profiles/r06_chain_timing_frag20.json")


STRONG_CASES = (("metric", 4096, 64), ("cfg4", 2048, 32), ("cfg5", 8192, 64))   # BASELINE.json: the TOTALS its metric / configs name


def strong_scaling(rank, world, dev, args, backend, native_gather, every, clock_warm):
    """The FIXED-TOTAL reading of BASELINE.json's metric ("at 4096 envs x 64 agents, 1/2/4/8 GPU") and of its sharded configs
    (cfg4 = 2048 x 32 over 4 GPUs, cfg5 = 8192 x 64 over 8): the total is cut into `world` contiguous env blocks, each rank steps
    its block, the per-env records travel in the same all-gather as in the weak run and are INSIDE the timed region.  Same
    protocol as `value`: barrier + synchronise on both sides, max over ranks.  Returns {case: {...}} on every rank.
    What it will show: a 512-env shard is a one-wave-per-SIMD pool -- a latency chain per step that more GPUs do
    not shorten -- so the fixed-total curve flattens near 2-3x at 8 GPUs while the weak curve stays linear."""
    import torch
    from tactics2d_amd import dist as D, layout as L
    out = {}
    for name, total, agents in STRONG_CASES:
        if total % world:
            out[f"{name}_{total}x{agents}"] = dict(skipped=f"{total} envs do not divide over {world} ranks")
            continue
        per = total // world
        scene = build_scene(name, total, agents, seed=0).shard(rank * per, (rank + 1) * per)
        run = Runner(scene, dev, args.variant, auto_reset=not args.no_reset, outputs=args.outputs, seed=2000 + rank)
        chained = args.mode == "chain"
        ev = max(e for e in (1, 2, 4, 8, 16, 32) if e <= max(1, min(every, args.steps, L.RECORD_RING // 2)))
        frag = ev if world > 1 or os.environ.get("T2D_FORCE_GATHER") else max(1, min(args.fragment, L.RECORD_RING, ACTION_SETS))
        gather = None
        if world > 1 or os.environ.get("T2D_FORCE_GATHER"):
            if native_gather:
                D.NativeGather.bootstrap(run.pool, rank, world)
                gather = D.NativeGather(run.pool, world, every=ev, device=dev)
            else:
                rec = torch.as_tensor(run.pool.device_array(L.F_RECORD), device=dev).view(torch.int32)
                gather = D.ResultGather(rec, world, every=ev)
        run.align = gather is not None

        def hook(gather=gather, run=run):
            if gather is None:
                return
            if native_gather:
                gather.launch(None, run.stream.cuda_stream)
            else:
                with torch.cuda.stream(run.stream):
                    gather.launch(run.k - 1)

        def barrier():
            torch.cuda.synchronize()
            if world > 1:
                torch.distributed.barrier()
                torch.cuda.synchronize()

        mode = "chain" if chained else "step"
        clock_warm()
        run.run(mode, args.warmup, frag, hook)
        if gather is not None:
            with torch.cuda.stream(run.stream):
                gather.wait()
        barrier()
        t0 = time.perf_counter()
        run.run(mode, args.steps, frag, hook)
        if gather is not None:
            with torch.cuda.stream(run.stream):
                gather.wait()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            torch.distributed.all_gather(allt, t)
            elapsed = max(float(x.item()) for x in allt)
        out[f"{name}_{total}x{agents}"] = dict(
            value=total * agents * args.steps / elapsed, unit="participant-steps/s", total_envs=total, envs_per_gpu=per,
            participants_per_env=agents, us_per_step=1e6 * elapsed / args.steps, step_form=run.pool.step_form(frag) if chained else "step",
            fragment=frag, gather_every=(ev if gather is not None else None), gather_native=(bool(native_gather) if gather is not None else None))
        run.close()
    return out


def timed(runner, mode, steps, warmup, frag, clock_warm=None, reps=1):
    """wall time and HIP-event span of `steps` steps after `warmup` untimed ones: (us per step, event-span us per step)"""
    import torch
    best = None
    for _ in range(reps):
        if clock_warm:
            clock_warm()
        runner.run(mode, warmup, frag)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(runner.stream)
        t0 = time.perf_counter()
        runner.run(mode, steps, frag)
        e1.record(runner.stream)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        r = (1e6 * el / steps, 1e3 * e0.elapsed_time(e1) / steps)
        best = r if best is None or r[0] < best[0] else best
    return best


def time_config(name, steps, warmup, dev, variant, frag, clock_warm):
    """One BASELINE.json configuration at its per-GPU size, device-resident actions, auto-reset on: per step as separate
    launches and as chained fragments (cfg2's single-ego kernel takes separate launches either way).  cfg4 / cfg5 are the
    per-GPU shards of the 4- / 8-GPU configurations."""
    n_env, agents = DEFAULTS[name]
    scene = build_scene(name, n_env, agents, seed=0)
    r = Runner(scene, dev, variant)
    us_s, _ = timed(r, "step", steps, warmup, frag, clock_warm)
    us_c, _ = timed(r, "chain", steps, warmup, frag, clock_warm)
    forms = dict(separate_launches=r.pool.step_form(1), chained=r.pool.step_form(frag))   # which kernel form each took (t2d_step_form)
    r.close()
    best = min(us_s, us_c)
    return dict(envs=n_env, participants_per_env=agents, value=scene.n / (best * 1e-6), unit="participant-steps/s",
                us_per_step=best, us_per_step_separate_launches=us_s, us_per_step_chained=us_c, kernel_form=forms, steps=steps, warmup=warmup)


def closed_loop(scene, dev, variant, steps, warmup, clock_warm, outputs):
    """The loop the reference's callers run -- action = policy(obs); env.step(action), envs/parking.py:219-256 -- kept on the
    device: per env group and step a policy kernel that reads the state the previous step left behind and writes an [n, 2]
    (steering, accel) tensor -> t2d_step reading it in place, nothing synchronising with the host (tactics2d_amd/csrc/
    t2d_loop.hip: a stand-in policy of a few flops per participant, so what is timed is the step path under the real
    dependency).  One pool on one stream first (the plain closed loop), then G env groups on G streams -- their launches
    overlap: what gives a closed-loop caller part of the overlap t2d_step_n gives an open-loop one.  Same results whatever G
    (tests/test_gpu_closed_loop.py)."""
    import torch
    from tactics2d_amd.debug import ClosedLoop, env_groups   # (the stand-in policy + its runner: libt2d_hip_debug.so, include/t2d_debug.h)
    N = scene.n
    rows = []
    long_steps = max(steps, 400)

    def measure(loop, n):
        clock_warm()
        loop.run(max(warmup, 32))
        torch.cuda.synchronize()
        t = time.perf_counter()
        loop.run(n)
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t) / n

    for G, launcher in ((1, "thread"), (2, "threads"), (4, "threads")):
        if scene.n_env % G:
            continue
        eg = env_groups(scene, G, dev.index)
        eg.configure(lambda p: (p.set_integrator_variant(variant), p.set_auto_reset(True),
                                p.set_outputs(velocity=outputs == "all", applied=outputs == "all")))
        try:
            loop = ClosedLoop(eg, launcher, scene.interval_ms)
            us = min(measure(loop, steps) for _ in range(2))
            us_long = measure(loop, long_steps)
            loop.close()
            rows.append(dict(groups=G, launcher=launcher, us_per_step=us, us_per_step_long_run=us_long, long_run_steps=long_steps))
        except Exception as e:   # noqa: BLE001 -- reported in the line, never fatal for it
            rows.append(dict(groups=G, launcher=launcher, error=str(e)))
        eg.close()
    ok = [r for r in rows if "us_per_step" in r]
    if not ok:
        return dict(error="no closed-loop configuration ran", candidates=rows)
    best = min(ok, key=lambda r: r["us_per_step"])
    single = next((r for r in ok if r["groups"] == 1), None)
    return dict(us_per_step=best["us_per_step"], value=N / (best["us_per_step"] * 1e-6), unit="participant-steps/s", groups=best["groups"],
                launcher=best["launcher"], steps=steps, warmup=max(warmup, 32),
                us_per_step_long_run=best["us_per_step_long_run"], long_run_steps=long_steps,
                us_per_step_one_pool_one_stream=(single or {}).get("us_per_step"),
                form="per env group and step: policy kernel -> t2d_step (one fused launch), the group's own stream, no host synchronisation; "
                     "the groups' launches overlap (GPU_MAX_HW_QUEUES=%s)" % os.environ.get("GPU_MAX_HW_QUEUES"),
                producer="t2d_debug_feedback_policy: a stand-in policy, one launch per group and step -- reads x, y, heading, speed of step "
                         "k - 1, writes [n, 2] (steering, accel) in the reference's action layout, bound with t2d_bind_actions_strided",
                bit_identical_to="the same policy and t2d_step calls on one pool holding all the envs (tests/test_gpu_closed_loop.py)",
                candidates=rows,
                note="us_per_step = wall time of `steps` iterations of all groups incl. the final synchronise, best of 2 after the clock ramp; "
                     "which group count wins depends on how the runtime maps streams to hardware queues on the box (profiles/r04_closed_loop_sweep_*.json: "
                     "2 groups overlap with 8 queues, 4 groups with 4; the others serialise)")


def next_rows(dev, clock_warm, metric_scene):
    """The rows SURVEY 8f adds around the step (IDM, lidar, the ParkingEnv vector step): >= 200 back-to-back launches each
    after the clock ramp, wall time including the final synchronise."""
    import torch
    from tactics2d_amd import scenarios as S
    from tactics2d_amd.envs import VecParkingEnv
    out = {}

    def loop(fn, n=300, warm=60):
        clock_warm()
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t) / n

    r = Runner(metric_scene, dev, "fast", idm=True)
    out["idm_kernel_us"] = loop(lambda: r.pool.idm_actions(None, r.stream.cuda_stream))
    out["idm_note"] = (f"t2d_idm_actions alone, {metric_scene.n_env} x {metric_scene.A}: every vehicle but the ego IDM-controlled "
                       f"(desired speed 25 m/s, horizon 120 m)")
    r.close()
    # the same controllers on a per-GPU shard of config 3 (1024 x 64 highway, every vehicle but the egos IDM-controlled): one
    # t2d_step per step = idm_kernel + step launch, against t2d_step_n fragments, whose integrator waves run the controllers
    hw = build_scene("cfg3", *DEFAULTS["cfg3"], seed=0)
    r = Runner(hw, dev, "fast", idm=True)
    r.pool.bind_actions(r.a0.data_ptr(), r.a1.data_ptr())
    r.pool.set_step_chaining(0)   # idm_kernel a launch of its own ahead of every step launch
    out["idm_pool_step_us_separate_launches"] = loop(lambda: r.pool.step(hw.interval_ms, r.stream.cuda_stream))
    r.pool.set_step_chaining(1)
    out["idm_pool_step_us_one_launch"] = loop(lambda: r.pool.step(hw.interval_ms, r.stream.cuda_stream))
    out["idm_pool_step_us_fragments"] = loop(lambda: r.pool.step_n(20, hw.interval_ms, 0, r.stream.cuda_stream), n=30, warm=6) / 20.0
    out["idm_pool_note"] = (f"{hw.n_env} x {hw.A} highway envs with 63 IDM agents each, device-resident actions for the egos, auto-reset on: "
                            f"per step as idm_kernel + step launch, as one step launch with the controllers in its front "
                            f"({r.pool.step_form(1)}) and as t2d_step_n fragments of 20 ({r.pool.step_form(20)}: the integrator waves run them)")
    r.close()
    sc = S.parking(4096)
    r = Runner(sc, dev, "fast")
    r.pool.lidar_config(360, 20.0, False)
    out["lidar_kernel_us"] = loop(lambda: r.pool.lidar_scan(None, r.stream.cuda_stream))
    out["lidar_note"] = "t2d_lidar_scan alone, 4096 parking envs x 360 beams, 8 static quads (32 edges) per env"
    r.close()
    for source in ("layout", "generator"):
        env = VecParkingEnv(4096, max_step=200, auto_reset=True, seed=1, scene_source=source)
        env.reset()
        lo = torch.tensor([-0.524, -2.0], device=dev); hi = torch.tensor([0.524, 2.0], device=dev)
        acts = [lo + (hi - lo) * torch.rand((4096, 2), device=dev) for _ in range(8)]
        k = [0]

        def step():
            env.step_torch(acts[k[0] & 7]); k[0] += 1
        out[f"vec_parking_env_step_us_{source}_scenes"] = loop(step)
        env.close()
    out["vec_parking_env_note"] = ("VecParkingEnv.step_torch at 4096 envs: ego step + 360-beam lidar, device-resident actions, no host "
                                   "copy or synchronisation; 'generator' = every finished episode continues in a newly generated lot")
    # the Gym-API HOST path (what the reference's caller sees: numpy actions in, numpy 5-tuple out; envs/parking.py:219-256):
    # one t2d_step_host call per step -- actions staged + box-checked, step, (scan,) pack, one frame back
    import numpy as np
    from tactics2d_amd.envs import ParkingEnv

    def host_loop(env, n_envs, n=300, warm=40):
        rng = np.random.default_rng(0)
        acts = [env.action_space.sample(rng, n_envs) for _ in range(8)]
        clock_warm()
        for k in range(warm):
            env.step(acts[k & 7])
        t = time.perf_counter()
        for k in range(n):
            env.step(acts[k & 7])
        return 1e6 * (time.perf_counter() - t) / n

    for key, kw in (("vec_parking_env_step_numpy_us", dict(info_lidar=False)),
                    ("vec_parking_env_step_numpy_us_with_lidar_in_info", dict()),
                    ("vec_parking_env_step_numpy_us_with_120_beams_in_info", dict(lidar_beams=120)),
                    ("vec_parking_env_step_numpy_us_generator_scenes", dict(info_lidar=False, scene_source="generator"))):
        env = VecParkingEnv(4096, max_step=200, auto_reset=True, seed=1, **kw)
        env.reset()
        out[key] = host_loop(env, 4096)
        env.close()
    env = ParkingEnv(max_step=int(2e4), seed=0)
    env.reset()
    rng = np.random.default_rng(0)
    acts = [env.action_space.sample(rng) * 0.2 for _ in range(64)]
    for k in range(200):
        env.step(acts[k & 63])
    t = time.perf_counter()
    for k in range(3000):
        o, rew, te, tr, info = env.step(acts[k & 63])
        if te or tr:
            env.reset()
    us = 1e6 * (time.perf_counter() - t) / 3000
    env.close()
    out["parking_env_single_step_us"] = us
    out["parking_env_single_steps_per_s"] = 1e6 / us
    # the PCIe-inclusive rate of the METRIC scene: actions of all 262 144 participants from host memory (2 MB per step), the
    # ego records back, through BatchedScenarioManager.step_host (copy commands: the pool is far too large for mapped memory)
    from tactics2d_amd.traffic import BatchedScenarioManager
    m = BatchedScenarioManager(metric_scene.n_env, metric_scene.A, max_step=metric_scene.status.get("max_step", 2000),
                               step_size=metric_scene.interval_ms)
    metric_scene.load(m.pool)
    m.pool.set_auto_reset(True)
    rng = np.random.default_rng(1)
    hacts = []
    for _ in range(4):
        a0, a1 = metric_scene.sample_actions(rng)
        hacts.append(np.ascontiguousarray(np.stack([a1, a0], 1), np.float32))
    clock_warm()
    for k in range(20):
        m.step_host(hacts[k & 3], fresh=False)
    t = time.perf_counter()
    for k in range(200):
        m.step_host(hacts[k & 3], fresh=False)
    us = 1e6 * (time.perf_counter() - t) / 200
    # ... and with the actions written straight into the pool's pinned staging buffer (what a policy running on the host
    # would do): no staging copy
    buf = m.pool.host_action_buffer()
    for k in range(20):
        buf[:] = hacts[k & 3]
        m.step_host(buf, fresh=False)
    t_fill = t_all = 0.0
    for k in range(200):
        t0 = time.perf_counter()
        buf[:] = hacts[k & 3]           # (stands for the policy writing its output: not part of the step)
        t1 = time.perf_counter()
        m.step_host(buf, fresh=False)
        t_all += time.perf_counter() - t1
    out["metric_step_host_us_actions_in_pinned_buffer"] = 1e6 * t_all / 200
    m.close()
    out["metric_step_host_us"] = us
    out["metric_step_host_value"] = metric_scene.n / (us * 1e-6)
    out["metric_step_host_note"] = (f"the metric scene ({metric_scene.n_env} x {metric_scene.A}) stepped from HOST actions: 2 MB of actions up "
                                    "per step (pinned staging + one async copy), one t2d_step launch, the egos' packed frame down, one "
                                    "synchronisation per step -- the PCIe-inclusive rate of the path (never `value`)")
    out["host_path_note"] = ("VecParkingEnv.step at 4096 envs, host to host: numpy actions in (Box.contains checked while they are staged), "
                             "(obs, reward, terminated, truncated, infos) out as views of a pinned frame nobody holds any more -- one "
                             "library call, no per-field copies; '_with_120_beams_in_info' = VecParkingEnv(lidar_beams=120): the every-third-beam scan the "
                             "tutorial policy consumes (docs/tutorial/train_parking_demo.ipynb), bit-identical to [::3] of the 360-beam one; "
                             "'_with_lidar_in_info' adds the 360-beam scan to the frame (5.9 MB per "
                             "step over PCIe).  parking_env_single_*: BASELINE config 1, one ParkingEnv stepped through the reference's "
                             "5-tuple API (lidar in info), resets included; compare cpu_baseline.python_loop_value (physics only)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="metric", help="metric | cfg2 | cfg3 | cfg4 | cfg5")
    ap.add_argument("--envs", type=int, default=None, help="environments PER GPU")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--variant", default="fast", choices=["fast", "exact"])
    ap.add_argument("--mode", default="chain", choices=["chain", "step"],
                    help="chain: t2d_step_n fragments (one launch per fragment); step: one t2d_step launch per step")
    ap.add_argument("--fragment", type=int, default=32, help="steps per t2d_step_n call in chain mode (<= 32)")
    ap.add_argument("--outputs", default="all", choices=["state", "all"],
                    help="all (default): the reference's State -- x, y, heading, speed, vx / vy, the applied action -- plus flags and the env "
                         "records; state: without the derived vx / vy of the single-track models and the applied action (t2d_set_outputs)")
    ap.add_argument("--gather-every", type=int, default=32, help="N > 1: steps per all-gather of the result records (1 = every step; "
                    "it divides the record ring of 64 slots and is at most half of it)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP events")
    ap.add_argument("--no-configs", action="store_true", help="skip timing BASELINE.json's other configurations")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the IDM / lidar / ParkingEnv timings")
    ap.add_argument("--no-alternates", action="store_true", help="skip timing the other step mode and the all-outputs form")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip timing the closed loop (policy kernel -> t2d_step per env group)")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the fixed-total (strong-scaling) cases")
    ap.add_argument("--no-reset", action="store_true", help="skip the device-side auto-reset")
    ap.add_argument("--idm", action="store_true", help="non-ego vehicles driven by on-device IDM controllers (row f3); "
                    "adds the idm kernel to every step (not the metric configuration)")
    ap.add_argument("--clock-warm", type=int, default=3000, help="untimed steps of a scratch pool (same scene) before the "
                    "warm-up steps, so that the GPU has left its idle clocks when the timed region starts (0 = off)")
    args = ap.parse_args()

    import torch
    from tactics2d_amd import build as B, dist as D, layout as L

    # (T2D_DIST_BACKEND / T2D_FORCE_DEVICE exist to exercise the N > 1 code path on a one-GPU box: gloo, every rank on
    # the same device; never set in a real run)
    backend = os.environ.get("T2D_DIST_BACKEND", "nccl")
    rank, local_rank, world = D.init_process_group(backend)
    if "T2D_FORCE_DEVICE" in os.environ:
        local_rank = int(os.environ["T2D_FORCE_DEVICE"])
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); "
                         "there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    n_env, agents = DEFAULTS[args.config]
    n_env = args.envs or n_env
    agents = args.agents or agents
    scene = build_scene(args.config, n_env, agents, seed=rank)
    N = scene.n
    frag = max(1, min(args.fragment, L.RECORD_RING, ACTION_SETS))
    run = Runner(scene, dev, args.variant, auto_reset=not args.no_reset, outputs=args.outputs, seed=1000 + rank, idm=args.idm)
    geo_record_bytes = run.pool.geometry_bytes_per_launch()
    chained_ok = args.mode == "chain" and not args.idm

    # N > 1: the per-env result records of `gather_every` consecutive steps travel in ONE all-gather (a rollout fragment).
    # With RCCL (the real run) the library issues it itself -- t2d_gather: ncclAllGather reading the record ring in place,
    # on a stream of the pool's own, ordered after the steps by events; torch.distributed only ships the communicator id.
    # Without RCCL (gloo, the one-GPU rehearsal) the same exchange goes through torch.distributed.
    gather, gather_note = None, None
    use_gather = world > 1 or bool(os.environ.get("T2D_FORCE_GATHER"))
    every = max(1, args.gather_every)
    if use_gather:   # at least one gather inside the timed region, whatever --steps is (fragments end at multiples of `every`)
        every = max(e for e in (1, 2, 4, 8, 16, 32) if e <= max(1, min(every, args.steps, L.RECORD_RING // 2)))
    native_gather = backend == "nccl" and not os.environ.get("T2D_GATHER_TORCH")
    if use_gather:
        if L.RECORD_RING % every or every > L.RECORD_RING // 2:
            raise SystemExit(f"--gather-every {every} must divide {L.RECORD_RING} and be <= {L.RECORD_RING // 2}")
        frag = every   # a chained fragment = the steps one gather ships
        if native_gather:
            # every rank must end up on the same path: if the library's communicator cannot be created on ANY rank
            # (no librccl to dlopen, ncclCommInitRank failing), all of them fall back to torch.distributed -- and say so
            ok = 1
            try:
                D.NativeGather.bootstrap(run.pool, rank, world)
            except Exception as exc:   # noqa: BLE001 -- reported in the JSON line, not swallowed
                ok, gather_note = 0, f"t2d_comm_init failed on rank {rank}: {exc}"
            if world > 1:
                flag = torch.tensor([ok], dtype=torch.int32, device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = int(flag.item())
            if not ok:
                native_gather = False
                gather_note = gather_note or "t2d_comm_init failed on another rank"
                print("warning: native gather unavailable, using torch.distributed:", gather_note, file=sys.stderr)
        if native_gather:
            gather = D.NativeGather(run.pool, world, every=every, device=dev)
        else:
            rec = torch.as_tensor(run.pool.device_array(L.F_RECORD), device=dev).view(torch.int32)
            gather = D.ResultGather(rec, world, every=every)
    run.align = gather is not None
    n_gathers = [0]

    def hook():   # after every step (step mode) or fragment (chain mode)
        if gather is None:
            return
        if native_gather:
            k = gather.launch(None, run.stream.cuda_stream)
        else:
            with torch.cuda.stream(run.stream):
                k = gather.launch(run.k - 1)
        n_gathers[0] += k is not None

    def drain():
        if gather is not None:
            with torch.cuda.stream(run.stream):
                gather.wait()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # The GPU leaves its idle power state only after ~10 ms of sustained load (measured: the same 20 timed steps take
    # 32.4 us each straight after start-up and 29.9 us after 15 ms of any load), and falls back to it whenever the host
    # spends some tens of ms building the next pool.  The driver's run is 25 steps = 0.5 ms, so the clocks are ramped
    # before every timed region -- on a SCRATCH pool holding the same scene: the measured pools' states, step counts
    # and actions are untouched, the timed regions are unchanged.  Stated in config.untimed_prewarm.
    warm = Runner(scene, dev, args.variant, auto_reset=not args.no_reset, outputs=args.outputs, seed=7) if args.clock_warm else None

    def clock_warm():
        if warm is not None:
            warm.steps_single(args.clock_warm)
            torch.cuda.synchronize()

    mode = "chain" if chained_ok else "step"
    clock_warm()
    run.run(mode, args.warmup, frag, hook)
    drain()
    barrier()
    # ---- timed region: EXACTLY --steps steps, nothing but the step launches (and, N > 1, the gathers) in it; the HIP
    # events on the launch stream bracket the same region for the roofline's figure ----------------------------------
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(run.stream)
    g0 = n_gathers[0]
    t0 = time.perf_counter()
    run.run(mode, args.steps, frag, hook)
    host_enqueue_us = 1e6 * (time.perf_counter() - t0) / args.steps
    drain()
    ev1.record(run.stream)
    barrier()
    elapsed = time.perf_counter() - t0
    span_ms = ev0.elapsed_time(ev1)
    gathers_timed = n_gathers[0] - g0
    rank_ms = 1e3 * elapsed / args.steps
    per_rank = None
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(t) for _ in range(world)]
        torch.distributed.all_gather(allt, t)
        per_rank = [1e3 * float(x.item()) / args.steps for x in allt]
        elapsed = max(float(x.item()) for x in allt)

    # ---- per-kernel pass (outside `value`): the same steps again with HIP events recorded on the launch stream around
    # every launch (the event packets cost ~2 us per launch).  At least PROF_MIN steps whatever --steps is, every
    # kernel form run PREWARM steps untimed first (a first launch -- code object load, cold instruction cache -- is not
    # a launch duration). ----------------------------------------------------------------------------------------------
    kern = {}
    PREWARM, PROF_MIN = 32, 128
    n_prof = 0

    def profiled(mode_, n):
        run.pool.profile_enable(False)
        run.run(mode_, PREWARM, frag)
        barrier()
        run.pool.profile_enable(True)    # (the events stay readable until the next profile_enable call)
        run.run(mode_, n, frag)
        barrier()

    def read_kernels(ids, steps_per_launch=1):
        for kid, name in ids:
            ms, launches = run.pool.profile_read(kid)
            if launches:
                kern[name] = dict(avg_us=1e3 * ms / launches, launches=int(launches), steps_per_launch=steps_per_launch,
                                  avg_us_per_step=1e3 * ms / launches / steps_per_launch)

    if not args.no_profile and gather is None:
        n_prof = min(max(args.steps, PROF_MIN), 2000)
        n_prof -= n_prof % frag
        if chained_ok:
            profiled("chain", n_prof)
            read_kernels(((7, "step_kernel_chained"),), frag)
        profiled("step", min(n_prof, 512))
        read_kernels(((2, "step_kernel"), (4, "idm_kernel")))
        run.pool.set_fused_step(False)   # the two stand-alone kernels (the integrator is north_star's roofline kernel)
        profiled("step", min(n_prof, 200))
        read_kernels(((0, "integrate_kernel"), (1, "collide_kernel")))
        run.pool.set_fused_step(True)
        run.pool.profile_enable(False)

    pclk = None
    if world == 1 and gather is None and not args.no_profile:
        pclk = power_clock_under_load(run, mode, frag)
    integ_models = None
    if world == 1 and gather is None and not args.no_profile and args.config == "metric" and not args.no_next_rows:
        try:
            integ_models = integrator_per_model(args.variant)
        except Exception as exc:   # noqa: BLE001 -- a reported side measurement, never fatal for the line
            integ_models = dict(error=str(exc))

    # ---- the other ways of running the same steps, driver-timed like `value` -----------------------------------------
    alternates = None
    if world == 1 and gather is None and not args.no_alternates:
        alternates = {}
        other = "step" if mode == "chain" else "chain"
        if other == "step" or not args.idm:
            us, span = timed(run, other, args.steps, args.warmup, frag, clock_warm)
            alternates["separate_launches" if other == "step" else "chained"] = dict(
                us_per_step=us, event_span_us_per_step=span, value=N / (us * 1e-6),
                note=("one t2d_step launch per step (rounds 1-2's headline form)" if other == "step" else
                      f"t2d_step_n fragments of {frag} steps: one launch per fragment"))
        run.pool.set_outputs(velocity=args.outputs != "all", applied=args.outputs != "all")
        us, span = timed(run, mode, args.steps, args.warmup, frag, clock_warm)
        alternates["outputs_" + ("all" if args.outputs != "all" else "state")] = dict(
            us_per_step=us, event_span_us_per_step=span, value=N / (us * 1e-6),
            note="same run with the derived vx / vy of the single-track models and the applied action " +
                 ("stored as well (t2d_set_outputs(T2D_OUT_ALL): the reference's State)" if args.outputs != "all" else "not stored"))
        run.pool.set_outputs(velocity=args.outputs == "all", applied=args.outputs == "all")

    # ---- the same timed region once more WITHOUT the clock ramp: the GPU left idle for 0.4 s, then --warmup steps, then the
    # timed steps -- what the ramp contributes is driver-timed here, not quoted ---------------------------------------------
    no_ramp = None
    if world == 1 and gather is None and args.clock_warm:
        time.sleep(0.4)
        us, span = timed(run, mode, args.steps, args.warmup, frag, None)
        no_ramp = dict(value=N / (us * 1e-6), us_per_step=us, event_span_us_per_step=span,
                       note=f"same {args.steps} steps after 0.4 s of idle GPU and {args.warmup} warm-up steps, no untimed clock ramp")

    # state sanity after the run (not timed): flags/status distribution
    flags = run.pool.download(L.F_FLAGS)
    status = run.pool.download(L.F_STATUS)
    finite = bool(np.isfinite(run.pool.download(L.F_X)).all())
    comm = run.pool.comm_info() if gather is not None and native_gather else None
    run.close()

    # ---- BASELINE.json's other configurations + the next rows, timed by the same process ------------------
    configs = nrows = None
    if world == 1 and rank == 0 and args.config == "metric":
        if not args.no_configs:
            configs = {}
            for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
                configs[name] = time_config(name, max(args.steps, 128), max(args.warmup, 32), dev, args.variant, frag, clock_warm)
            configs["note"] = ("per-GPU sizes (cfg4 = 2048 x 32 over 4 GPUs, cfg5 = 8192 x 64 over 8 GPUs), device-resident actions, auto-reset "
                               "on, >= 128 timed steps after >= 32 warm-up steps each, as separate launches and as chained fragments "
                               f"of {frag} steps; us_per_step = the better of the two; wall time incl. the final synchronise")
        if not args.no_next_rows:
            nrows = next_rows(dev, clock_warm, scene)
    cloop = None
    if world == 1 and rank == 0 and not args.no_closed_loop and not args.idm:
        cloop = closed_loop(scene, dev, args.variant, args.steps, args.warmup, clock_warm, args.outputs)
    # ---- N > 1: the fixed-total (strong-scaling) reading of the same metric and of the sharded configs, every rank takes part --
    strong = None
    if (world > 1 or os.environ.get("T2D_FORCE_STRONG")) and args.config == "metric" and not args.idm and not args.no_strong:
        try:
            strong = strong_scaling(rank, world, dev, args, backend, native_gather, every, clock_warm)
            strong["note"] = ("fixed TOTAL sizes cut into contiguous env blocks over the ranks (weak `value` above: 4096 envs PER rank); the "
                              "all-gather of the records is inside the timed region; value = total participant-steps / max-over-ranks time")
        except Exception as exc:   # noqa: BLE001 -- reported in the line; `value` (the contract) is already measured
            strong = dict(error=f"rank {rank}: {exc}")
    if warm is not None:
        warm.close()

    if rank == 0:
        value = world * N * args.steps / elapsed
        # geometry the step reads per launch: the packed per-workgroup records (fp32 vertices and boxes, the fp64
        # boundary pieces of the lane unions, index ranges) as the library lays them out + the 16-B map boundary per env
        geo_bytes = geo_record_bytes + 16 * n_env
        step_bytes = (INTEGRATOR_BYTES + 4) * N + geo_bytes   # fused: the integrator's 44 B + the 4-B flag word
        step_us = span_ms * 1e3 / args.steps                  # HIP-event span of the timed region per step
        hbm = dict(bound="hbm", achieved=step_bytes / (step_us * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                   algorithmic_bytes_per_step=step_bytes,
                   note="what north_star names; the kernel is not under this roof (20 fp64 Euler sub-steps per "
                        "participant-step: ~2000 VALU instructions per wave for 48 B per lane)")
        hbm["frac"] = hbm["achieved"] / HBM_PEAK_GBS
        # PMC counters cannot be read from inside the process: instruction counts and HBM traffic come from the committed
        # rocprofv3 passes (scripts/profile_round.sh) -- valid only for the sources they were taken of
        tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
        tj = json.load(open(tf)) if os.path.exists(tf) else {}
        same_cfg = (tj.get("config"), tj.get("envs_per_gpu"), tj.get("participants_per_env")) == (args.config, n_env, agents)
        stale = tj.get("source_sha256") != B.source_hash()
        src = f"profiles/traffic_latest.json ({tj.get('tag')})"
        dom = "step_kernel_chained" if mode == "chain" else "step_kernel"
        sq = tj.get("sq_counters_per_step", {}).get(dom) if same_cfg else None
        traffic = tj.get("hbm_bytes_per_step", {}).get(dom) if same_cfg and not stale else None
        # What bounds the chained step, as MEASURED in round 6 (profiles/r06_chain_timing_frag20.json, r06_power_clock.json,
        # r06_sq_wait_chain.json): VALU issue -- but of a SIMD that holds four wave slots of which, on average, only 2.8 are in a
        # phase that issues: per slot and step 7 % is the dispatch gap between two workgroups, 5 % the hand-off wait, 12 % the
        # start-up (tables + geometry record -> LDS) and 6 % the store drain.  Not power: 947 W of a 1400 W cap, no PPT residency.
        roof = dict(bound="valu_issue_at_2.8_of_4_wave_slots_issuing", kernel=dom, peak=VALU_ISSUE_PEAK_GINST, unit="G wave-instructions/s",
                    peak_is="256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction (every class but plain 32-bit runs: "
                            "profiles/valu_issue_cycles.json); at the EFFECTIVE clock measured inside the kernel (2.26 GHz: s_memtime per "
                            "100 MHz tick) the peak is peak_at_measured_clock",
                    peak_at_measured_clock=VALU_ISSUE_PEAK_GINST * EFFECTIVE_CLOCK_GHZ / 2.4, measured_clock_ghz=EFFECTIVE_CLOCK_GHZ,
                    attribution="profiles/r06_chain_timing_frag20.json: per wave slot and step -- dispatch gap 1.2 us, hand-off wait 0.8, "
                                "start-up 2.1, integrator 4.2, event phases 7.5, store drain + word 1.1 (sums to the step); "
                                "profiles/r06_sq_wait_chain.json: a wave is parked 43 % of its life, issue-stalled 21 %, issuing 36 %",
                    power_clock=pclk,
                    step_us=step_us, counters_source=src, counters_stale=bool(stale) if tj else None,
                    traffic=traffic, hbm=hbm, timed_region_event_span_ms=span_ms,
                    how="step_us (what `achieved` divides by) = HIP-event span of the timed region on the launch stream / its steps "
                        "(the launches are back to back; in chain mode a launch holds up to 32 steps); valu_insts_per_step from the "
                        "rocprofv3 SQ pass of the same command and sources (a property of kernel + data, like the algorithmic bytes); "
                        "kernels[...] = HIP events around every launch in a second pass outside `value`",
                    kernels=kern or None)
        if sq and not stale:
            insts = float(sq["SQ_INSTS_VALU"])
            roof.update(achieved=insts / (step_us * 1e-6) / 1e9, valu_insts_per_step=insts,
                        valu_insts_per_wave=insts / sq["SQ_WAVES"], insts_per_wave=sq.get("SQ_INSTS", 0) / sq["SQ_WAVES"])
            roof["frac_at_4_cycles_per_instruction"] = roof["achieved"] / VALU_ISSUE_PEAK_GINST
            roof["frac"] = roof["frac_at_4_cycles_per_instruction"]
            roof["frac_at_measured_clock"] = roof["achieved"] / roof["peak_at_measured_clock"]
            # class-resolved: what the step's VALU instructions cost a SIMD by the MEASURED issue cost of their class
            # (profiles/valu_issue_cycles.json <- scripts/valu_roof.hip on this GPU) over the SIMD-cycles of the step
            cyc_f = os.path.join(ROOT, "profiles", "valu_issue_cycles.json")
            if os.path.exists(cyc_f) and "SQ_INSTS_VALU_FMA_F64" in sq:
                cyc = json.load(open(cyc_f))["cycles"]
                g = lambda k: float(sq.get(k, 0.0))
                fp64 = g("SQ_INSTS_VALU_FMA_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_ADD_F64")
                tr64, tr32 = g("SQ_INSTS_VALU_TRANS_F64"), g("SQ_INSTS_VALU_TRANS_F32")
                cvt, i64 = g("SQ_INSTS_VALU_CVT"), g("SQ_INSTS_VALU_INT64")
                i32 = g("SQ_INSTS_VALU_INT32")
                f32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32")
                fma32 = g("SQ_INSTS_VALU_FMA_F32")
                other = max(0.0, insts - (fp64 + tr64 + tr32 + cvt + i64 + i32 + f32 + fma32))   # moves, selects, compares, lane ops
                fixed = (fp64 * cyc["fp64_add_mul_fma"] + tr64 * cyc["trans_f64"] + tr32 * cyc["trans_f32"] + cvt * cyc["cvt"] +
                         i64 * cyc["int64"] + fma32 * cyc["f32_fma"])
                simple = i32 + f32 + other
                simd_cycles = step_us * 1e-6 * SIMD_CYCLES_PER_S
                hi = (fixed + simple * cyc["simple_32bit_between_fp64"]) / simd_cycles
                lo = (fixed + simple * cyc["int32_simple_back_to_back"]) / simd_cycles
                # (`frac` stays on one definition across rounds -- instructions x 4 cycles -- the class-weighted estimate has its own key)
                roof.update(frac_class_weighted=hi, frac_lower_bound=lo,
                            issue_cycles_per_class=dict(source="profiles/valu_issue_cycles.json", **{k: cyc[k] for k in (
                                "fp64_add_mul_fma", "trans_f64", "trans_f32", "cvt", "int64", "f32_fma", "simple_32bit_between_fp64",
                                "int32_simple_back_to_back", "int32_other", "f32_other", "pk_f32", "salu")}),
                            valu_insts_per_wave_by_class={k: v / sq["SQ_WAVES"] for k, v in dict(
                                fp64_add_mul_fma=fp64, trans_f64=tr64, trans_f32=tr32, cvt=cvt, int64=i64, int32=i32, f32_add_mul=f32,
                                f32_fma=fma32, other_moves_selects_compares=other).items()},
                            frac_is="instructions per step x 4 cycles / SIMD-cycles of the step (the definition of rounds 1-3)",
                            frac_class_weighted_is="sum over classes of (instructions per step x measured SIMD cycles per wave64 instruction of the class, 4 waves "
                                    "per SIMD) / (256 CUs x 4 SIMDs x 2.4 GHz x step_us); 32-bit integer / fp32 add-mul / move-select-compare "
                                    "instructions at the 4.0 cycles one of them costs BETWEEN fp64 instructions (the stream of this kernel); "
                                    "frac_lower_bound prices all of them at the 2.19 cycles of an unbroken run of plain 32-bit operations")
        else:   # never print a numerator that belongs to another binary
            roof.update(achieved=None, frac=None,
                        note="no instruction counts for these sources: re-run scripts/profile_round.sh; the HBM figure stands")
        # north_star's own roofline kernel: the stand-alone integrator (t2d_integrate), algorithmic 44 B per participant over its
        # launch duration (HIP events of the per-kernel pass) against the 8 TB/s peak -- the target there is >= 0.40
        integ = None
        if kern.get("integrate_kernel"):
            iu = kern["integrate_kernel"]["avg_us"]
            integ = dict(kernel="integrate_kernel<fast>", bound="hbm", avg_us=iu, algorithmic_bytes=INTEGRATOR_BYTES * N,
                         achieved=INTEGRATOR_BYTES * N / (iu * 1e-6) / 1e9, peak=HBM_PEAK_GBS, unit="GB/s",
                         north_star_target_frac=0.40,
                         note="20 fp64 Euler sub-steps per participant for 44 B: VALU-issue bound (DESIGN.md 4.1); avg_us includes the "
                              "~2 us of the HIP-event pair around each launch")
            integ["frac"] = integ["achieved"] / HBM_PEAK_GBS
            isq = tj.get("integrator_sq_per_launch") if same_cfg and not stale else None
            if isq and isq.get("SQ_WAVES"):
                integ.update(valu_insts_per_wave=isq["SQ_INSTS_VALU"] / isq["SQ_WAVES"],
                             valu_busy_frac=isq.get("_valu_busy_frac"), rocprofv3_avg_us=tj.get("kernel_trace_avg_us_per_step", {}).get("step", {}).get("integrate_kernel"))
        if integ is not None and integ_models is not None:
            integ["per_model_at_4M_participants"] = integ_models
        roof["integrator"] = integ
        gather_obj = None
        if gather is not None:
            gather_obj = dict(native=bool(native_gather), every=every, every_requested=args.gather_every, gathers_in_timed_region=gathers_timed,
                              rccl_world=(comm[1] if comm else None), rccl_rank0=(comm[2] if comm else None),
                              rccl_communicator=(bool(comm[0]) if comm else False),
                              ms_per_step_by_rank=per_rank, ms_per_step_min=(min(per_rank) if per_rank else rank_ms),
                              ms_per_step_max=(max(per_rank) if per_rank else rank_ms),
                              how=("t2d_gather: RCCL all-gather issued by the library from the record ring, on a stream of the pool's "
                                   "own; rccl_world = ncclCommCount of that communicator" if native_gather else
                                   "torch.distributed all_gather_into_tensor" + (f" ({gather_note})" if gather_note else " (no RCCL on this backend)")))
        how_steps = (f"t2d_step_n fragments of {frag} steps, one launch each" if mode == "chain" else "one t2d_step launch per step")
        out = dict(metric="participant-steps/sec (physics+collision)", value=value,
                   unit="participant-steps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=f"{scene.name}: {n_env} envs x {agents} participants per GPU, "
                                        f"interval 100 ms / delta_t 5 ms (20 Euler sub-steps), "
                                        f"integrator variant {args.variant}, fused step kernel, auto-reset "
                                        f"{'off' if args.no_reset else 'on'}" + (", IDM agents on" if args.idm else "") +
                                        f"; steps enqueued as {how_steps}; outputs stored: {args.outputs}",
                               config=args.config, envs_per_gpu=n_env, participants_per_env=agents, mode=mode,
                               fragment=(min(frag, args.steps) if mode == "chain" else 1), fragment_max=frag,
                               outputs=args.outputs, host_enqueue_us_per_step=host_enqueue_us,
                               untimed_prewarm=f"{args.clock_warm} steps of a scratch pool with the same scene before every timed region (GPU clock "
                                               f"ramp, the measured pool untouched), then {args.warmup} warm-up steps; per-kernel pass: "
                                               f"{PREWARM} untimed steps of each kernel form, then up to {n_prof} timed ones",
                               parallelism=(f"env-sharded x{world}, one async all-gather of the 8 B/env result records per {every} steps"
                                            if world > 1 else "single GPU")),
                   roofline=roof, closed_loop=cloop, value_without_clock_ramp=no_ramp, alternates=alternates, gather=gather_obj,
                   strong=strong,
                   configs=configs, next_rows=nrows,
                   check=dict(state_finite=finite,
                              flag_rates=[float((flags & b).astype(bool).mean()) for b in (1, 2, 4, 8)],
                              truncated_frac=float(status[:, 3].mean())))
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (the other ranks would sit in the barrier)
            out["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(out))
    bad_world = (world > 1 and backend == "nccl" and not os.environ.get("T2D_GATHER_TORCH") and
                 (not native_gather or comm is None or comm[1] != world))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    if bad_world:   # a multi-GPU run whose records did not travel through the library's own RCCL communicator of N ranks is not
        # the path this bench claims to measure: fail loudly (T2D_GATHER_TORCH=1 asks for the torch.distributed path on purpose)
        print(f"error: rank {rank}: the native RCCL gather is not in place (native={native_gather}, communicator={comm}); "
              f"expected a communicator of {world} ranks", file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
