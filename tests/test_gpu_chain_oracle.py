"""t2d_step_n held against the ORACLE directly -- not against t2d_step launches (tests/test_gpu_chain.py does that) -- and its
failure path.  Every form of the multi-step launch (chain, chain_split, loop, loop_pipe with and without lane waves,
ego_loop_pipe) steps its BASELINE.json shard for 8 steps with the exact integrator; a sample of 40 envs is replayed by the
oracle (oracle/t2d_oracle.c: integrate -> fp32 store -> collide -> status -> auto-reset, the loop of envs/parking.py:240-256
per env) and every per-step record, the final state, the flags and the counters must agree bit for bit (rewards to one fp32
ulp of the time-penalty table).  The golden roll-outs of tests/golden/rollouts.npz (made by the imported reference) go
through t2d_step_n teacher-forced.  Then the safety net: a hand-off broken on purpose (t2d_debug_chain_fault) must surface
T2D_ERR_STATE once, leave the pool at the start of the failed fragment, and the pool must reach the same end state with
plain launches from there."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

N_STEPS = 8
N_SAMPLE = 40


def _ring(sc, n_steps, seed=9):
    rng = np.random.default_rng(seed)
    sets = [sc.sample_actions(rng) for _ in range(n_steps)]
    return np.stack([s[0] for s in sets]), np.stack([s[1] for s in sets])


def _gpu_fragment(sc, r0, r1, chaining, split=True, want_form=None, variant="exact", calls=None):
    """n steps of the scene through t2d_step_n on a device-resident action ring; returns the start velocity columns, the
    per-step records and the final fields"""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    n_steps = r0.shape[0]
    a0 = torch.from_numpy(r0).to(dev).contiguous()
    a1 = torch.from_numpy(r1).to(dev).contiguous()
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_integrator_variant(variant)
    pool.set_auto_reset(True)
    pool.set_step_chaining(chaining)
    pool.set_split_step(split)
    if want_form is not None:
        assert pool.step_form(n_steps) == want_form, (pool.step_form(n_steps), want_form)
    start = dict(vx=pool.download(L.F_VX), vy=pool.download(L.F_VY))
    done = 0
    for c in (calls or (n_steps,)):
        pool.bind_actions(a0.data_ptr() + 4 * sc.n * done, a1.data_ptr() + 4 * sc.n * done)
        pool.step_n(c, sc.interval_ms, sc.n)
        done += c
    assert done == n_steps
    out = {f: pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_FLAGS, L.F_ENV_FLAGS,
                                         L.F_CNT_STEP, L.F_FRAME_MS, L.F_STATUS, L.F_REWARD, L.F_IOU, L.F_CNT_NO_ACTION)}
    out["record"] = pool.download(L.F_RECORD).reshape(L.RECORD_RING, sc.n_env, 2)
    assert pool.step_count() == n_steps
    pool.close()
    return start, out


def _oracle_env_chain(O, sc, e, r0, r1, vx0, vy0):
    """env e of the scene stepped by the oracle alone, auto-reset to its start state: the per-step (status bytes, reward),
    and what the pool's fields hold after the last step"""
    from tactics2d_amd import layout as L
    A = sc.A
    one = sc.shard(int(e), int(e) + 1)
    sl = slice(e * A, (e + 1) * A)
    iou = bool(sc.status.get("check_arrival") or sc.status.get("check_no_action"))
    cfg = O.make_config(**sc.status)
    ep = O.EpisodeState(1, one.target, None, np.stack([one.x[:1], one.y[:1]], 1)) if iou else None
    x0, y0, h0, v0 = one.x.copy(), one.y.copy(), one.heading.copy(), one.speed.copy()
    x, y, h, v, vx, vy = x0.copy(), y0.copy(), h0.copy(), v0.copy(), vx0[sl].copy(), vy0[sl].copy()
    is_dyn = sc.rows[one.type_id, L.P_MODEL] == L.MODEL_DYNAMICS
    cnt = np.zeros(1, np.int32); frame = np.zeros(1, np.int32)
    steps = []
    flags = None
    for k in range(r0.shape[0]):
        O.set_trig(1)
        o = O.integrate(sc.rows, x, y, h, v, vx, vy, r0[k][sl], r1[k][sl], one.type_id, one.active, sc.interval_ms)
        O.set_trig(0)
        act = one.active.astype(bool)
        x = np.where(act, np.float32(o[:, 0]), x); y = np.where(act, np.float32(o[:, 1]), y)
        h = np.where(act, np.float32(o[:, 2]), h); v = np.where(act, np.float32(o[:, 3]), v)
        vx = np.where(act & ~is_dyn, np.float32(o[:, 4]), vx); vy = np.where(act & ~is_dyn, np.float32(o[:, 5]), vy)
        flags, envf = O.collide(sc.rows, 1, A, x, y, h, one.type_id, one.active, one.static, one.boundary, one.boundary_valid,
                                one.lanes, 1)
        if iou:
            st, rw, _ = O.status_ex(cfg, A, flags, sc.interval_ms, cnt, frame, sc.rows, x, y, h, one.type_id, ep)
        else:
            st, rw = O.status(cfg, 1, A, flags, sc.interval_ms, cnt, frame)
        steps.append((st[0].copy(), np.float32(rw[0])))
        if st[0, 2] or st[0, 3]:   # terminated | truncated: ParkingEnv.reset -- the fused auto-reset of the step launch
            x, y, h, v, vx, vy = x0.copy(), y0.copy(), h0.copy(), v0.copy(), vx0[sl].copy(), vy0[sl].copy()
            cnt[:] = 0; frame[:] = 0
            if ep is not None:
                ep.reset_envs(np.ones(1, bool))
    return steps, dict(x=x, y=y, h=h, v=v, vx=vx, vy=vy, flags=flags, env_flags=envf[0], cnt=cnt[0], frame=frame[0], is_dyn=is_dyn)


FORMS = [
    # name, scene, t2d_set_step_chaining, split, the form t2d_step_form must name
    ("chain", lambda S: S.mixed(4096, 64, seed=3), 1, True, "chain"),
    ("chain_split", lambda S: S.mixed(1024, 64, seed=3), 2, True, "chain_split"),
    ("chain_small", lambda S: S.highway(1024, 64, seed=1), 2, False, "chain"),
    ("loop", lambda S: S.highway(1024, 64, seed=1), 3, True, "loop"),
    ("loop_pipe", lambda S: S.highway(1024, 64, seed=1), 1, True, "loop_pipe"),
    ("loop_pipe_lane_waves_mixed", lambda S: S.mixed(1024, 64, seed=3), 1, True, "loop_pipe"),
    ("loop_pipe_lane_waves_intersection", lambda S: S.intersection(512, 32, seed=2), 1, True, "loop_pipe"),
    ("ego_loop_pipe", lambda S: S.parking(4096), 1, True, "ego_loop_pipe"),
]


@pytest.mark.parametrize("name,make,chaining,split,form", FORMS, ids=[f[0] for f in FORMS])
def test_every_form_of_step_n_against_the_oracle_chain(oracle, name, make, chaining, split, form):
    from tactics2d_amd import layout as L, scenarios as S
    sc = make(S)
    rng = np.random.default_rng(31)
    if sc.A > 1:   # (stress jitter, test only: poses scattered so that collisions, off-lane and out-of-bound fire within 8 steps)
        sc.x = (sc.x + rng.normal(0, 1.5, sc.n)).astype(np.float32)
        sc.y = (sc.y + rng.normal(0, 1.0, sc.n)).astype(np.float32)
        sc.status.update(max_step=5)   # ... and the time limit ends every episode inside the fragment
    else:          # parking: egos on the bay (Arrival), egos that never move (NoAction), a short time limit
        tc = sc.target.mean(1)
        on = np.arange(sc.n_env) % 4 == 0
        sc.x[on] = tc[on, 0] + rng.normal(0, 0.05, on.sum()).astype(np.float32)
        sc.y[on] = tc[on, 1] + rng.normal(0, 0.05, on.sum()).astype(np.float32)
        sc.heading[on] = sc.target_heading[on]
        sc.status.update(max_step=6, no_action_max_step=3)
    r0, r1 = _ring(sc, N_STEPS)
    if sc.A == 1:
        still = np.arange(sc.n_env) % 4 <= 1
        r0[:, still] = 0.0
    start, got = _gpu_fragment(sc, r0, r1, chaining, split, form)
    envs = np.sort(rng.choice(sc.n_env, size=min(sc.n_env, N_SAMPLE), replace=False))
    ends = 0
    for e in envs:
        steps, fin = _oracle_env_chain(oracle, sc, int(e), r0, r1, start["vx"], start["vy"])
        sl = slice(e * sc.A, (e + 1) * sc.A)
        for k, (st, rw) in enumerate(steps):
            word = int(st[0]) | int(st[1]) << 8 | int(st[2]) << 16 | int(st[3]) << 24
            assert int(got["record"][k, e, 1]) == word, (name, int(e), k, hex(int(got["record"][k, e, 1])), hex(word))
            grw = got["record"][k, e, 0:1].view(np.float32)[0]
            assert abs(float(grw) - float(rw)) <= 2e-6, (name, int(e), k, float(grw), float(rw))
            ends += int(st[2] or st[3])
        for f, key in ((L.F_X, "x"), (L.F_Y, "y"), (L.F_HEADING, "h"), (L.F_SPEED, "v")):
            assert np.array_equal(got[f][sl], fin[key]), (name, int(e), key)
        nd = ~fin["is_dyn"]
        assert np.array_equal(got[L.F_VX][sl][nd], fin["vx"][nd]) and np.array_equal(got[L.F_VY][sl][nd], fin["vy"][nd])
        assert np.array_equal(got[L.F_FLAGS][sl], fin["flags"]), (name, int(e))
        assert got[L.F_ENV_FLAGS][e] == fin["env_flags"]
        assert got[L.F_CNT_STEP][e] == fin["cnt"] and got[L.F_FRAME_MS][e] == fin["frame"]
        st_last = steps[-1][0]
        assert np.array_equal(got[L.F_STATUS][e], st_last)
    assert ends > 0, "no sampled episode ended: the auto-reset inside the fragment was not exercised"


@pytest.mark.parametrize("tag", ["kin_100_5", "dyn_100_5", "kin_50_3", "dyn_50_3"])
def test_golden_rollouts_through_step_n(tag):
    """tests/golden/rollouts.npz -- the VEHICLE_ACTION_LIST roll-outs of the imported reference (tests/test_physics.py:52-73)
    -- through every form of t2d_step_n: each recorded reference state k is a participant of its own (teacher forcing: 64
    per env, far apart), a fragment of TWO steps on a resident action ring takes it through actions k and k + 1, and the result
    is held against the reference's own state k + 2: 1e-5 per step (north_star) on top of the fp32 rounding of the start
    state.  Exact and fast integrator; chained, looping, looping with integrator waves, and plain launches agree bit for bit."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    g = H.load_npz("rollouts.npz")
    traj, acts, row = g[f"{tag}_traj"], g[f"{tag}_act"], g[f"{tag}_row"]
    interval = int(tag.split("_")[1])
    n_case = len(acts) - 1
    A = 64
    n_env = (n_case + A - 1) // A
    n = n_env * A
    st = np.zeros((n, 4), np.float32); st[:n_case] = np.float32(traj[:n_case])
    st[:, 0] += 0.0   # (no geometry, no boundary: events cannot end an episode, poses may coincide)
    active = np.zeros(n, np.uint8); active[:n_case] = 1
    ring0 = np.zeros((2, n), np.float32); ring1 = np.zeros((2, n), np.float32)
    for j in range(2):
        ring0[j, :n_case] = np.float32(acts[j:j + n_case, 0]); ring1[j, :n_case] = np.float32(acts[j:j + n_case, 1])
    a0 = torch.from_numpy(ring0).to(dev).contiguous(); a1 = torch.from_numpy(ring1).to(dev).contiguous()
    want = traj[2:2 + n_case]
    start_err = np.abs(np.float32(traj[:n_case]).astype(np.float64) - traj[:n_case]).max()
    for variant in ("exact", "fast"):
        outs = {}
        for chaining, form in ((0, "step"), (2, "chain"), (3, "loop"), (1, "loop_pipe")):
            pool = ParticipantPool(n_env, A)
            pool.set_param_table(row[None])
            pool.set_integrator_variant(variant)
            pool.set_step_chaining(chaining)
            pool.reset(st[:, 0], st[:, 1], st[:, 2], st[:, 3], np.zeros(n, np.uint8), active)
            pool.bind_actions(a0.data_ptr(), a1.data_ptr())
            assert pool.step_form(2) == form, (pool.step_form(2), form)
            pool.step_n(2, interval, n)
            outs[form] = np.stack([pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED)], 1)
            pool.close()
        for form, o in outs.items():
            assert np.array_equal(o, outs["step"]), (tag, variant, form)
        err = H.state_err(outs["chain"][:n_case], want)
        tol = 2e-5 + 4 * start_err
        if tag.startswith("dyn"):   # (every step of the dynamics roll-outs is well conditioned: the fixture says so)
            assert H.dyn_tolerance(g[f"{tag}_sens"], 1e-5)[1].all()
        assert err.max() <= tol, (tag, variant, err.max(0), tol)


def _fault_scene():
    from tactics2d_amd import scenarios as S
    return S.mixed(96, 64, seed=13)   # 24 step workgroups, every env kind


@pytest.mark.parametrize("kind,code", [(1, "different XCDs"), (2, "ran out")])
def test_a_broken_hand_off_is_reported_once_rolled_back_and_survived(kind, code):
    """kind 1: workgroup 1 posts its step 1 with a foreign XCC id (what a consumer on another XCD would see); kind 2: it never
    posts (its consumer's bounded wait runs out).  Three fragments of 6 steps are enqueued, the SECOND carries the fault, no
    host synchronisation in between.  The first sync must fail with T2D_ERR_STATE, the pool must then sit exactly where the
    second fragment began (state, counters, step count = 6), a second sync must succeed, and 12 further steps -- plain
    launches now -- must end in exactly the state 18 ordinary steps reach."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import _ffi, layout as L
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    sc = _fault_scene()
    r0, r1 = _ring(sc, 18, seed=2)
    a0 = torch.from_numpy(r0).to(dev).contiguous(); a1 = torch.from_numpy(r1).to(dev).contiguous()
    fields = (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_IDS, L.F_CNT_STEP, L.F_FRAME_MS)

    from tactics2d_amd import debug as D      # (fault injection is a hook of libt2d_hip_debug.so: include/t2d_debug.h)

    def fresh():
        p = D.pool(sc.n_env, sc.A)
        sc.load(p)
        p.set_integrator_variant("exact")
        p.set_auto_reset(True)
        return p

    ref = fresh()
    ref.set_step_chaining(0)
    after6 = None
    for k in range(18):
        ref.bind_actions(a0.data_ptr() + 4 * sc.n * k, a1.data_ptr() + 4 * sc.n * k)
        ref.step(sc.interval_ms)
        if k == 5:
            after6 = [ref.download(f) for f in fields]
    want = [ref.download(f) for f in fields + (L.F_FLAGS, L.F_STATUS, L.F_REWARD)]
    want_rec = ref.download(L.F_RECORD)
    ref.close()

    pool = fresh()
    pool.set_step_chaining(2)           # CHAIN form whatever the pool's size
    pool.set_split_step(False)
    assert pool.step_form(6) == "chain"

    def frag(k0, n):
        pool.bind_actions(a0.data_ptr() + 4 * sc.n * k0, a1.data_ptr() + 4 * sc.n * k0)
        pool.step_n(n, sc.interval_ms, sc.n)

    frag(0, 6)
    D.chain_fault(pool, kind)
    frag(6, 6)
    D.chain_fault(pool, 0)
    frag(12, 6)                         # enqueued behind the failed fragment: must not disturb its checkpoint
    assert pool.step_count() == 18
    with pytest.raises(_ffi.T2DError) as ei:
        pool.sync()
    assert ei.value.code == _ffi.ERR_STATE and code in str(ei.value) and "rolled back to step 6" in str(ei.value), str(ei.value)
    pool.sync()                         # reported once
    assert pool.step_count() == 6
    pm = sc.rows[sc.type_id, L.P_MODEL] == L.MODEL_POINTMASS
    for f, w in zip(fields, after6):
        g = pool.download(f)
        if f in (L.F_VX, L.F_VY):   # (state only for a point mass; the single-track models' vx / vy are outputs of the next step)
            g, w = g[pm], w[pm]
        assert np.array_equal(g, w, equal_nan=True), f
    assert pool.step_form(6) == "step"  # chaining is off for this pool now
    frag(6, 12)                         # ... so this is twelve plain launches
    got = [pool.download(f) for f in fields + (L.F_FLAGS, L.F_STATUS, L.F_REWARD)]
    for f, g, w in zip(fields + (L.F_FLAGS, L.F_STATUS, L.F_REWARD), got, want):
        assert np.array_equal(g, w, equal_nan=True), f
    assert np.array_equal(pool.download(L.F_RECORD), want_rec)
    # chaining can be switched on again: the counters start afresh and the pool chains as before
    pool.set_step_chaining(2)
    frag(0, 6)
    pool.sync()
    pool.close()


def test_a_pool_that_changes_its_chained_shape_restarts_the_counters():
    """LOOP -> CHAIN -> CHAIN with one workgroup per env -> fewer envs per workgroup (new lane geometry), all on one pool:
    the per-workgroup step counters of the chained forms only continue between launches of one shape (round-3 advisor:
    `chain_count` moved for every form, and a form switch left every workgroup waiting for a count that never came)."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    sc = S.mixed(96, 64, seed=4)
    r0, r1 = _ring(sc, 24, seed=6)
    a0 = torch.from_numpy(r0).to(dev).contiguous(); a1 = torch.from_numpy(r1).to(dev).contiguous()
    fields = (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_STATUS, L.F_REWARD, L.F_CNT_STEP)

    def fresh(chaining):
        p = ParticipantPool(sc.n_env, sc.A)
        sc.load(p)
        p.set_integrator_variant("exact")
        p.set_auto_reset(True)
        p.set_step_chaining(chaining)
        return p

    ref = fresh(0)
    for k in range(24):
        ref.bind_actions(a0.data_ptr() + 4 * sc.n * k, a1.data_ptr() + 4 * sc.n * k)
        ref.step(sc.interval_ms)
    want = [ref.download(f) for f in fields]
    ref.close()
    pool = fresh(3)
    forms = []
    for k0, chaining, split in ((0, 3, False), (6, 2, False), (12, 2, True), (18, 2, False)):
        pool.set_step_chaining(chaining)
        pool.set_split_step(split)
        forms.append(pool.step_form(6))
        pool.bind_actions(a0.data_ptr() + 4 * sc.n * k0, a1.data_ptr() + 4 * sc.n * k0)
        pool.step_n(6, sc.interval_ms, sc.n)
    assert forms == ["loop", "chain", "chain_split", "chain"], forms
    pool.sync()
    for f, w in zip(fields, want):
        assert np.array_equal(pool.download(f), w, equal_nan=True), f
    pool.close()


def test_an_action_ring_needs_bound_memory():
    """t2d_step_n(act_step_stride > 0) on the pool's own action fields would read past them: refused (round-3 advisor)"""
    from tactics2d_amd import _ffi, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    sc = S.highway(8, 64, seed=2)
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    a0, a1 = sc.sample_actions(np.random.default_rng(0))
    pool.set_actions(a0, a1)
    with pytest.raises(_ffi.T2DError) as ei:
        pool.step_n(4, 100, sc.n)
    assert ei.value.code == _ffi.ERR_INVALID
    pool.step_n(4, 100, 0)   # one action set repeated: fine
    pool.sync()
    pool.close()


def test_a_failure_during_the_first_step_leaves_a_complete_checkpoint():
    """Fault kind 3: workgroup 1 posts its step 0 with a foreign XCC id, on a grid larger than the device holds (8192 x 64:
    2048 step workgroups per step, 1024 resident) -- its consumer fails while the last step-0 workgroups have not started.
    The fragment's own failure must not keep them from storing their checkpoint: after the rollback EVERY env is exactly
    where the fragment began (round 4 skipped the checkpoint whenever any failure was on record: stale or zero state for
    the late workgroups' envs)."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import _ffi, layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    sc = S.mixed(8192, 64, seed=4)
    r0, r1 = _ring(sc, 12, seed=3)
    a0 = torch.from_numpy(r0).to(dev).contiguous(); a1 = torch.from_numpy(r1).to(dev).contiguous()
    fields = (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_IDS, L.F_CNT_STEP, L.F_FRAME_MS)
    from tactics2d_amd import debug as D
    pool = D.pool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_auto_reset(True)
    pool.set_step_chaining(2)
    pool.set_split_step(False)
    assert pool.step_form(6) == "chain"

    def frag(k0, n):
        pool.bind_actions(a0.data_ptr() + 4 * sc.n * k0, a1.data_ptr() + 4 * sc.n * k0)
        pool.step_n(n, sc.interval_ms, sc.n)

    frag(0, 6)
    pool.sync()
    after6 = [pool.download(f) for f in fields]
    for rep in range(3):
        pool.set_step_chaining(2)
        D.chain_fault(pool, 3)
        frag(6, 6)
        D.chain_fault(pool, 0)
        with pytest.raises(_ffi.T2DError) as ei:
            pool.sync()
        assert ei.value.code == _ffi.ERR_STATE and "rolled back to step 6" in str(ei.value), str(ei.value)
        assert pool.step_count() == 6
        for f, w in zip(fields, after6):
            g = pool.download(f)
            assert np.array_equal(g, w, equal_nan=True), (rep, f, int((g != w).sum()))
    pool.close()
