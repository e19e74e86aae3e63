#!/usr/bin/env python3
"""bench.py -- participant-steps/s (physics + collision + status) of the batched env.step().

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one t2d_step (integrate + collide + status epilogue) over every participant of the
rank's pool, including the fused device-side auto-reset of finished envs (t2d_set_auto_reset), and,
for N > 1, the asynchronous RCCL all-gather of the 8-byte per-env result records.  Inputs
(state, actions, geometry) are resident in HBM before the timed region starts.  Weak scaling:
every rank owns --envs environments (default 4096 x 64 participants, the metric workload).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (the step kernel
is bound by fp64 VALU issue, not by HBM: the line gives the fraction of the VALU issue roof --
instructions per launch from the committed rocprofv3 SQ pass over the launch duration measured here
with HIP events -- and the HBM figure beside it), `configs` (BASELINE.json's other configurations,
timed in the same run) and `cpu_baseline` (the C oracle -- a port of the reference's algorithm --
timed on one host core and on all host cores, OpenMP over envs, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; streams sharing a queue serialise.  The env
# groups below need one queue each next to torch's own streams -- must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue roof: a SIMD issues one wave64 VALU instruction (fp64 included: full rate on CDNA4) per 4 cycles;
# 256 CUs x 4 SIMDs at the 2.4 GHz peak clock of the same guide
VALU_ISSUE_PEAK_GINST = 256 * 4 * 2.4 / 4.0
INTEGRATOR_BYTES = 44          # algorithmic bytes per participant-step
COLLIDE_BYTES = 20             # + per-env geometry (computed from the scene)


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


def _cpu_leg(scene, n_env, threads, target_seconds):
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
            if el >= target_seconds or steps >= 2000:
                break
    finally:
        O.set_threads(1)
    return n * steps / el, steps, el


def cpu_baseline(scene, target_seconds=10.0):
    """The oracle (C port of the reference algorithm, fp64 scalar) on a bounded sample of the SAME
    scene, SURVEY.md 8(d): (1) one core, first 96 envs; (2) best effort: the same code with its
    batch loops spread over all host cores (OpenMP), whole scene.  `value` is the all-cores rate."""
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
        # how many threads actually help is a property of the box (cgroup CPU quotas are invisible to
        # sched_getaffinity): try a few counts on 2 steps each, keep the best, then run the timed leg
        cand = sorted({c for c in (4, 8, 16, 32, 64, 128, cores) if c <= cores})
        quota = None
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                quota = max(1, int(round(int(q) / int(per))))
                cand = sorted(set(cand) | {min(cores, quota)})
        except Exception:
            pass
        trial = {c: _cpu_leg(scene, scene.n_env, c, 0.0)[0] for c in cand}   # target 0 s -> exactly one step each
        best = max(trial, key=trial.get)
        allc, stepsN, elN = _cpu_leg(scene, scene.n_env, best, target_seconds)
        out.update(value=allc, cores=best, one_core_value=one,
                   sample=f"all {scene.n_env} envs x {A} participants of the same scene, {stepsN} steps, "
                          f"{elN:.1f} s on {best} OpenMP threads over envs (best of {sorted(trial)} tried on one step each; "
                          f"{cores} logical CPUs visible, cgroup quota {quota}); one_core_value: first "
                          f"{min(scene.n_env, 96)} envs, {steps1} steps, {el1:.1f} s on 1 core; C oracle "
                          f"oracle/t2d_oracle.c (fp64 scalar restatement of the reference)")
    return out


DEFAULTS = {"metric": (4096, 64), "cfg5": (1024, 64), "cfg2": (4096, 1), "cfg3": (1024, 64), "cfg4": (512, 32)}
GATHER_EVERY = 16


def time_config(name, steps, warmup, dev, variant="fast", clock_warm=None):
    """One BASELINE.json configuration at its per-GPU size, single launch per step, device-resident actions, auto-reset
    on: (participant-steps/s, us per step).  cfg4 / cfg5 are the per-GPU shards of the 4- / 8-GPU configurations."""
    import torch
    from tactics2d_amd.pool import ParticipantPool
    n_env, agents = DEFAULTS[name]
    scene = build_scene(name, n_env, agents, seed=0)
    pool = ParticipantPool(scene.n_env, scene.A, dev.index)
    scene.load(pool)
    pool.set_integrator_variant(variant)
    pool.set_auto_reset(True)
    rng = np.random.default_rng(5)
    ring = []
    for _ in range(4):
        a0, a1 = scene.sample_actions(rng)
        ring.append((torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)))
    st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()

    def run(n):
        for k in range(n):
            a0, a1 = ring[k & 3]
            pool.bind_actions(a0.data_ptr(), a1.data_ptr())
            pool.step(scene.interval_ms, st.cuda_stream)
    if clock_warm:
        clock_warm()   # the GPU fell back to its idle clocks while the host built this scene
    run(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    pool.close()
    return dict(envs=n_env, participants_per_env=agents, value=scene.n * steps / el, unit="participant-steps/s",
                us_per_step=1e6 * el / steps, steps=steps, warmup=warmup)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="metric", help="metric | cfg2 | cfg3 | cfg4 | cfg5")
    ap.add_argument("--envs", type=int, default=None, help="environments PER GPU")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--variant", default="fast", choices=["fast", "exact"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP events")
    ap.add_argument("--no-configs", action="store_true", help="skip timing BASELINE.json's other configurations")
    ap.add_argument("--no-reset", action="store_true", help="skip the device-side auto-reset")
    ap.add_argument("--idm", action="store_true", help="non-ego vehicles driven by on-device IDM controllers (row f3); "
                    "adds the idm kernel to every step (not the metric configuration)")
    ap.add_argument("--groups", type=int, default=0, help="env groups on separate HIP streams per GPU "
                    "(0 = auto: 4 when the env count allows, else 1)")
    ap.add_argument("--clock-warm", type=int, default=500, help="untimed steps of a scratch pool (same scene) before the "
                    "warm-up steps, so that the GPU has left its idle clocks when the timed region starts (0 = off)")
    ap.add_argument("--split", action="store_true", help="two-kernel step (integrate + check_status) instead of the fused launch")
    args = ap.parse_args()

    import torch
    from tactics2d_amd import dist as D, layout as L

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
    # env groups: independent envs cut into G pools on G HIP streams, so that one group's start-up latency and
    # tail overlap the others' busy middle, and step k+1 of a group starts while step k of the next one still
    # runs (tactics2d_amd/pipeline.py).  G = 1 is the plain single-launch step.
    # The headline line is measured with ONE group (one launch per step: its per-launch HIP-event durations are
    # what rocprofv3 sees for the same command).  The pipelined variant (4 groups; worth it once a group still
    # fills the GPU's wave slots, >= 32 Ki participants) is timed afterwards and reported as `pipelined`.
    G = args.groups if args.groups else 1
    pipelined_G = 4 if (not args.groups and world == 1 and n_env % 4 == 0 and n_env * agents >= 131072) else 0
    from tactics2d_amd.pipeline import EnvGroups
    eg = EnvGroups(scene, G, device_id=local_rank)
    N = scene.n

    geo_record_bytes = sum(p.geometry_bytes_per_launch() for p in eg.pools)

    def setup(p):
        p.set_integrator_variant(args.variant)
        p.set_fused_step(not args.split)
        if not args.no_reset:
            p.set_auto_reset(True)   # finished envs restart inside the step launch (no extra kernels)
    eg.configure(setup)
    if args.idm:
        from tactics2d_amd.controller import IDMController, install
        for (lo, hi), p in zip(eg.bounds, eg.pools):
            sub = slice(lo * agents, hi * agents)
            cid = np.full((hi - lo, agents), L.IDM_NONE, np.uint8)
            veh = (scene.rows[scene.type_id[sub], L.P_MODEL] != L.MODEL_POINTMASS).reshape(hi - lo, agents)
            cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
            install(p, [IDMController(desired_speed=25.0, horizon=120.0)], cid.reshape(-1))

    # actions: a ring of pre-generated batches resident in HBM, bound zero-copy (each group its slice)
    rng = np.random.default_rng(1000 + rank)
    ring = []
    for _ in range(4):
        a0, a1 = scene.sample_actions(rng)
        ring.append((torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)))
    # N > 1: the per-env result records of 16 consecutive steps travel in ONE all-gather per group (a rollout fragment).
    # With RCCL (the real run) the library issues it itself -- t2d_gather: ncclAllGather reading the record ring in
    # place, on a stream of the pool's own, ordered after the group's steps by events; torch.distributed only ships the
    # communicator id.  Without RCCL (gloo, the one-GPU rehearsal) the same exchange goes through torch.distributed.
    gathers = []
    native_gather = backend == "nccl" and not os.environ.get("T2D_GATHER_TORCH")
    gather_note = None
    if (world > 1 or os.environ.get("T2D_FORCE_GATHER")) and native_gather:
        # every rank must end up on the same path: if the library's communicator cannot be created on ANY rank
        # (no librccl to dlopen, ncclCommInitRank failing), all of them fall back to torch.distributed -- and say so
        ok = 1
        try:
            for p in eg.pools:
                D.NativeGather.bootstrap(p, rank, world)
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
    if world > 1 or os.environ.get("T2D_FORCE_GATHER"):
        for p in eg.pools:
            if native_gather:
                gathers.append(D.NativeGather(p, world, every=GATHER_EVERY, device=dev))
            else:
                rec = torch.as_tensor(p.device_array(L.F_RECORD), device=dev).view(torch.int32)
                gathers.append(D.ResultGather(rec, world, every=GATHER_EVERY))
    step_no = [0]
    torch.cuda.synchronize()

    def one_step(k):
        a0, a1 = ring[k & 3]
        eg.bind_actions(a0, a1)
        eg.step(scene.interval_ms)
        if gathers:
            for g, s in zip(gathers, eg.streams):
                if native_gather:
                    g.launch(step_no[0], s.cuda_stream)
                else:
                    with torch.cuda.stream(s):
                        g.launch(step_no[0])
        step_no[0] += 1

    def drain():
        for g, s in zip(gathers, eg.streams):
            with torch.cuda.stream(s):
                g.wait()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # The GPU leaves its idle power state only after ~10 ms of sustained load (measured: the same 20 timed steps take
    # 32.4 us each straight after start-up and 29.9 us after 15 ms of any load), and falls back to it whenever the host
    # spends some tens of ms building the next pool.  The driver's run is 25 steps = 0.8 ms, so the clocks are ramped
    # before every timed region -- on a SCRATCH pool holding the same scene: the measured pools' states, step counts
    # and actions are untouched, the timed regions are unchanged.  Stated in config.untimed_prewarm.
    pool_w = None
    if args.clock_warm:
        # (on the null stream: one more torch stream would share a hardware queue with an env group's, see pipeline.py)
        from tactics2d_amd.pool import ParticipantPool
        pool_w = ParticipantPool(scene.n_env, scene.A, local_rank)
        scene.load(pool_w)
        setup(pool_w)

    def clock_warm():
        if pool_w is None:
            return
        for k in range(args.clock_warm):
            a0, a1 = ring[k & 3]
            pool_w.bind_actions(a0.data_ptr(), a1.data_ptr())
            pool_w.step(scene.interval_ms)
        torch.cuda.synchronize()
    clock_warm()
    for k in range(args.warmup):
        one_step(k)
    drain()
    barrier()
    # ---- timed region: EXACTLY --steps steps, nothing but the step launches in it; the HIP events on the
    # launch streams bracket the same region for the roofline's aggregate figure -------------------------
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    eg.fork()
    t0 = time.perf_counter()
    for k in range(args.steps):
        one_step(k)
    host_enqueue_us = 1e6 * (time.perf_counter() - t0) / args.steps
    drain()
    ev_end = []
    for s in eg.streams:
        e = torch.cuda.Event(enable_timing=True)
        e.record(s)
        ev_end.append(e)
    barrier()
    elapsed = time.perf_counter() - t0
    span_ms = max(ev0.elapsed_time(e) for e in ev_end)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel pass (outside `value`): the same steps again with HIP events recorded on the launch streams around
    # every kernel (the event packets cost ~2 us per kernel).  At least PROF_MIN launches per kernel whatever --steps
    # is, and every kernel form is run PREWARM times untimed first: the two-kernel form's kernels have never run in
    # this process before that, and a first launch (code object load, cold instruction cache) is not a launch duration.
    kern = {}
    PREWARM, PROF_MIN = 10, 100

    def read_kernels(ids):
        for kid, name in ids:
            tot, cnt = 0.0, 0
            for p in eg.pools:
                ms, launches = p.profile_read(kid)
                tot += ms; cnt += launches
            if cnt:
                kern[name] = dict(avg_us=1e3 * tot / cnt, launches=cnt)

    def profiled_pass(n):
        eg.configure(lambda p: p.profile_enable(False))
        for k in range(PREWARM):
            one_step(k)
        drain()
        barrier()
        eg.configure(lambda p: p.profile_enable(True))
        for k in range(n):
            one_step(k)
        drain()
        barrier()

    n_prof = 0
    if not args.no_profile:
        n_prof = min(max(args.steps, PROF_MIN), 2000 // G)
        profiled_pass(n_prof)
        read_kernels(((0, "integrate_kernel"), (1, "collide_kernel"), (2, "step_kernel"), (4, "idm_kernel")))
        if not args.split:   # also time the two stand-alone kernels (the integrator is north_star's roofline kernel)
            eg.configure(lambda p: p.set_fused_step(False))
            profiled_pass(min(n_prof, 200))
            read_kernels(((0, "integrate_kernel"), (1, "collide_kernel")))
            eg.configure(lambda p: p.set_fused_step(True))
        eg.configure(lambda p: p.profile_enable(False))

    # state sanity after the run (not timed): flags/status distribution
    flags = eg.download(L.F_FLAGS)
    status = eg.download(L.F_STATUS)
    x_end = eg.download(L.F_X)
    finite = bool(np.isfinite(x_end).all())

    pipelined = None
    if pipelined_G and not gathers:
        # 2 and 4 groups are both timed: 4 overlap more, but cost 4 host launches per step, and a short run (the driver's
        # 20 steps) ends before the host has the queues full; the better one is reported, both are listed
        tried = {}
        for Gp in (2, pipelined_G):
            eg.close()
            eg = EnvGroups(scene, Gp, device_id=local_rank)
            eg.configure(setup)
            torch.cuda.synchronize()
            clock_warm()
            for k in range(max(args.warmup, 20)):
                one_step(k)
            barrier()
            evp = torch.cuda.Event(enable_timing=True)
            evp.record()
            eg.fork()
            tp = time.perf_counter()
            for k in range(args.steps):
                one_step(k)
            ends = []
            for s_ in eg.streams:
                e = torch.cuda.Event(enable_timing=True)
                e.record(s_)
                ends.append(e)
            barrier()
            el_p = time.perf_counter() - tp
            tried[Gp] = (el_p, max(evp.elapsed_time(e) for e in ends))
        best = min(tried, key=lambda g_: tried[g_][0])
        el_p, span_p = tried[best]
        pipelined = dict(env_groups=best, value=N * args.steps / el_p, unit="participant-steps/s",
                         ms_per_step=1e3 * el_p / args.steps, timed_region_event_span_ms=span_p,
                         ms_per_step_by_groups={str(g_): 1e3 * v[0] / args.steps for g_, v in tried.items()},
                         note=f"same workload and steps, cut into {best} env groups of {n_env // best} envs on "
                              f"{best} HIP streams (tactics2d_amd/pipeline.py, t2d_step_groups): one group's start-up "
                              f"latency and tail overlap the others' busy middle and the next step of the next group; "
                              f"results identical to the single launch (tests/test_gpu_pipeline.py)")
    eg.close()

    # ---- BASELINE.json's other configurations, timed by the same process ("for each config") ----------
    configs = None
    if world == 1 and rank == 0 and not args.no_configs and args.config == "metric":
        configs = {}
        for name in ("cfg2", "cfg3", "cfg4", "cfg5"):
            configs[name] = time_config(name, max(args.steps, 100), max(args.warmup, 20), dev, args.variant, clock_warm)
        configs["note"] = ("per-GPU sizes (cfg4 = 2048 x 32 over 4 GPUs, cfg5 = 8192 x 64 over 8 GPUs), one launch per step, "
                           "device-resident actions, auto-reset on, >= 100 timed steps after >= 20 warm-up steps each; "
                           "wall time of the step loop incl. the final synchronise")

    if pool_w is not None:
        pool_w.close()
    if rank == 0:
        value = world * N * args.steps / elapsed
        # geometry the step reads per launch: the packed per-workgroup records (fp32 vertices and boxes, the fp64
        # boundary pieces of the lane unions, index ranges) as the library lays them out + the 16-B map boundary per env
        geo_bytes = geo_record_bytes + 16 * n_env
        roof = None
        if kern:
            # ALGORITHMIC bytes of ONE launch = one env group of N / G participants
            Ng, geo_g = N // G, geo_bytes / G
            per_launch = {"integrate_kernel": INTEGRATOR_BYTES * Ng, "collide_kernel": COLLIDE_BYTES * Ng + geo_g,
                          # fused: the integrator's 44 B + the 4-B flag word (poses never leave registers)
                          "step_kernel": (INTEGRATOR_BYTES + 4) * Ng + geo_g,
                          # idm: x, y, heading, speed, ids, ctrl id in; 2 actions + leader out
                          "idm_kernel": 33 * Ng}
            in_step = {"step_kernel"} if not args.split else {"integrate_kernel", "collide_kernel"}
            dom = max(in_step & set(kern), key=lambda k_: kern[k_]["avg_us"])
            per_launch_gbs = per_launch[dom] / (kern[dom]["avg_us"] * 1e-6) / 1e9
            # G launches of the kernel are in flight at any time (one per env group / stream): the bandwidth the
            # kernel achieves is the bytes of ALL its launches in the timed region over the HIP-event span of
            # that region (for G = 1: bytes per launch / launch duration, launches being back to back)
            ach = per_launch[dom] * G * args.steps / (span_ms * 1e-3) / 1e9
            traffic, traffic_src = None, None
            tf = os.path.join(ROOT, "profiles", "traffic_latest.json")
            tj = json.load(open(tf)) if os.path.exists(tf) else {}
            same = (tj.get("config"), tj.get("envs_per_gpu"), tj.get("participants_per_env"), tj.get("groups", 1)) == \
                (args.config, n_env, agents, G)
            if same:   # PMC counters cannot be read from inside the process: taken from the committed rocprofv3
                traffic = tj["hbm_bytes_per_launch"].get(dom)       # pass of the same command (scripts/profile_round.sh)
                traffic_src = f"profiles/traffic_latest.json ({tj.get('tag')}): " + tj.get("source", "")
            sq = tj.get("sq_counters_per_dispatch", {}).get(dom) if same else None
            hbm = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                       algorithmic_bytes_per_launch=per_launch[dom], per_launch_GBs=per_launch_gbs,
                       note="what north_star names; the kernel is not under this roof (20 fp64 Euler sub-steps per "
                            "participant-step: ~2000 VALU instructions per wave for 48 B per lane)")
            if sq:
                # The roof the step kernel IS under: fp64 VALU issue.  Instructions per launch from the
                # committed SQ pass of this very command and kernel (a property of kernel + data, like the algorithmic
                # bytes), launch duration measured here.  SQ_ACTIVE_INST_VALU (quad-cycles) of the same pass gives the
                # VALU-busy fraction of that (serialised, slower-clocked) profiled launch for comparison.
                insts = float(sq["SQ_INSTS_VALU"])
                # launch duration: with ONE launch per step and the launches back to back on one stream (fused step, one env
                # group) the HIP-event span of the timed region divided by its launches IS the average launch duration
                # -- the figure rocprofv3 reports for the same command (profiles/*_kernel_stats.csv); events recorded
                # around every single launch (the second pass) put ~2.4 us of event packets between the kernels
                back_to_back = G == 1 and in_step == {"step_kernel"}
                launch_us = span_ms * 1e3 / args.steps if back_to_back else kern[dom]["avg_us"]
                ach_i = insts / (launch_us * 1e-6) / 1e9
                kcyc = sq["SQ_BUSY_CYCLES"] / 32.0          # summed over 8 XCDs x 4 SEs
                roof = dict(bound="valu_fp64_issue", kernel=dom, achieved=ach_i, peak=VALU_ISSUE_PEAK_GINST,
                            unit="G wave-instructions/s", frac=ach_i / VALU_ISSUE_PEAK_GINST,
                            valu_insts_per_launch=insts, valu_insts_per_wave=insts / sq["SQ_WAVES"],
                            valu_busy_frac_in_profiled_launch=4.0 * sq["SQ_ACTIVE_INST_VALU"] / (256 * 4 * kcyc),
                            peak_is="256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 VALU instruction (fp64 is full rate)",
                            counters_source=traffic_src)
            else:
                roof = dict(hbm, kernel=dom)
            roof.update(traffic=traffic, traffic_source=traffic_src, hbm=hbm, concurrent_launches=G,
                        launches_in_timed_region=G * args.steps, timed_region_event_span_ms=span_ms,
                        avg_kernel_us=kern[dom]["avg_us"],
                        launch_us=(span_ms * 1e3 / args.steps if (G == 1 and in_step == {"step_kernel"}) else kern[dom]["avg_us"]),
                        how=(f"launch_us (what `achieved` divides by): HIP-event span of the timed region / its launches when there "
                             f"is one back-to-back launch per step, else avg_kernel_us; "
                             f"avg_kernel_us: HIP events around each launch of the kernel on its launch stream, {kern[dom]['launches']} "
                             f"launches in a second pass of the same steps after {PREWARM} untimed launches of the same form (outside "
                             f"`value`); hbm.achieved = algorithmic bytes of all launches in the timed region / HIP-event span of "
                             f"that region"),
                        kernels={k_: dict(avg_us=v["avg_us"], launches=v["launches"],
                                          algorithmic_bytes=per_launch[k_],
                                          achieved_GBs=per_launch[k_] / (v["avg_us"] * 1e-6) / 1e9)
                                 for k_, v in kern.items()})
        if pipelined is not None:
            step_bytes = (INTEGRATOR_BYTES + 4) * N + geo_bytes
            pipelined["aggregate_GBs"] = step_bytes * args.steps / (pipelined["timed_region_event_span_ms"] * 1e-3) / 1e9
            pipelined["aggregate_frac_of_hbm_peak"] = pipelined["aggregate_GBs"] / HBM_PEAK_GBS
        gather_how = ("t2d_gather: RCCL all-gather issued by the library from the record ring, on a stream of the pool's own"
                      if native_gather else "torch.distributed all_gather_into_tensor" +
                      (f" ({gather_note})" if gather_note else " (no RCCL on this backend)"))
        out = dict(metric="participant-steps/sec (physics+collision)", value=value,
                   unit="participant-steps/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=1e3 * elapsed / args.steps, higher_is_better=True, scaling="weak",
                   vs_baseline=None, dtype="f64", data="synthetic",
                   config=dict(workload=f"{scene.name}: {n_env} envs x {agents} participants per GPU, "
                                        f"interval 100 ms / delta_t 5 ms (20 Euler sub-steps), "
                                        f"integrator variant {args.variant}, {'two-kernel' if args.split else 'fused single-launch'} step, auto-reset "
                                        f"{'off' if args.no_reset else 'on'}" + (", IDM agents on" if args.idm else "") +
                                        (f", {G} env groups of {n_env // G} envs pipelined on {G} HIP streams" if G > 1 else ""),
                               config=args.config, envs_per_gpu=n_env, participants_per_env=agents, env_groups=G,
                               host_enqueue_us_per_step=host_enqueue_us,
                               untimed_prewarm=f"{args.clock_warm} steps of a scratch pool with the same scene before every timed region (GPU clock ramp, the "
                                               f"measured pools untouched), then "
                                               f"{args.warmup} warm-up steps before the timed region; per-kernel pass: {PREWARM} untimed "
                                               f"launches of each kernel form, then {n_prof} timed ones",
                               parallelism=f"env-sharded x{world}, per env group one async all-gather of the 8 B/env result records per "
                                           f"{GATHER_EVERY} steps ({gather_how})"
                               if world > 1 else "single GPU"),
                   roofline=roof, pipelined=pipelined, configs=configs,
                   check=dict(state_finite=finite,
                              flag_rates=[float((flags & b).astype(bool).mean()) for b in (1, 2, 4, 8)],
                              truncated_frac=float(status[:, 3].mean())))
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only (the other ranks would sit in the barrier)
            out["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
