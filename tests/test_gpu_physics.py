"""GPU parity of t2d_integrate (through the C ABI) -- run on the MI355X with -m gpu.

Bars (BASELINE.json north_star): fp32 pose within 1e-5 abs of the reference's fp64 result.
On top of that, the "exact" kernel variant must equal the deterministic-trig oracle bit for bit
after the fp32 store -- including the stiff SingleTrackDynamics cases no tolerance can cover.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

TOL = 1e-5  # north_star tolerance, abs, on fp32 pose state


def _groups(d):
    for iv in np.unique(d["timing"][:, 0]):
        yield int(iv), np.nonzero(d["timing"][:, 0] == iv)[0]


@pytest.mark.parametrize("name,model", [("kin_random.npz", "kin"), ("dyn_random.npz", "dyn"),
                                        ("pm_random.npz", "pm")])
def test_exact_variant_is_bit_identical_to_oracle(oracle, name, model):
    d = H.load_npz(name)
    mism = 0
    for iv, m in _groups(d):
        want = oracle_f32 = np.float32(H.oracle_physics(oracle, d["rows"], d["type_id"][m], d["state"][m],
                                                        d["action"][m], iv, model, trig=1))
        got = H.gpu_physics(d["rows"], d["type_id"][m], d["state"][m], d["action"][m], iv, "exact", model)
        cols = [0, 1, 2, 3, 6, 7] + ([4, 5] if model != "dyn" else [])
        for c in cols:
            bad = got[:, c].view(np.uint32) != want[:, c].view(np.uint32)
            # +0.0 / -0.0 are the same value
            bad &= ~((got[:, c] == 0) & (want[:, c] == 0))
            mism += int(bad.sum())
            assert not bad.any(), (f"{name} interval {iv} col {c}: {bad.sum()} of {len(bad)} differ, "
                                   f"first idx {np.nonzero(bad)[0][:5]} got {got[bad, c][:3]} want {want[bad, c][:3]}")
    assert mism == 0


@pytest.mark.parametrize("variant", ["fast", "exact", "fast_resummed", "fast_iterated"])
def test_kinematics_within_1e5_of_reference(variant):
    d = H.load_npz("kin_random.npz")
    worst = np.zeros(6)
    for iv, m in _groups(d):
        got = H.gpu_physics(d["rows"], d["type_id"][m], d["state"][m], d["action"][m], iv, variant, "kin")
        e = H.state_err(got, d["out"][m], cols=6)
        worst = np.maximum(worst, e.max(0))
        assert np.allclose(got[:, 6:8], np.float32(d["applied"][m]), rtol=0, atol=0)
    print("kinematics", variant, "max abs err x,y,h,v,vx,vy:", worst)
    assert worst.max() <= TOL, worst


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_pointmass_within_1e5_of_reference(variant):
    d = H.load_npz("pm_random.npz")
    worst = np.zeros(6)
    for iv, m in _groups(d):
        got = H.gpu_physics(d["rows"], d["type_id"][m], d["state"][m], d["action"][m], iv, variant, "pm")
        e = H.state_err(got, d["out"][m], cols=6)
        worst = np.maximum(worst, e.max(0))
    print("pointmass", variant, "max abs err:", worst)
    assert worst.max() <= TOL, worst


def _pm_euler_gpu(d, m, iv, variant, through_step=False):
    """every case of the mask as its own one-agent env: t2d_integrate, or the whole t2d_step (the side kernel runs ahead of it)"""
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    st = np.float32(d["state"][m])
    pool = ParticipantPool(int(m.sum()), 1)
    try:
        pool.set_param_table(d["rows"])
        pool.set_integrator_variant(variant)
        pool.reset(st[:, 0], st[:, 1], st[:, 2], np.zeros(len(st), np.float32), np.uint8(d["type_id"][m]), vx=st[:, 3], vy=st[:, 4])
        pool.set_actions(np.float32(d["action"][m, 0]), np.float32(d["action"][m, 1]))
        (pool.step if through_step else pool.integrate)(int(iv))
        return np.stack([pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_APPLIED0, L.F_APPLIED1)], 1)
    finally:
        pool.close()


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_pointmass_euler_backend_within_1e5_of_reference_and_bit_equal_to_the_oracle(variant, oracle):
    """PointMass(backend="euler") (point_mass.py:177-207; T2D_MODEL_POINTMASS_EULER): fixtures by import within 1e-5, the fp32
    state bit-equal to the oracle in deterministic-trig mode, the same through t2d_step (the side kernel ahead of the fused one)."""
    d = H.load_npz("pm_euler.npz")
    st = np.float32(d["state"])
    worst = np.zeros(6)
    for iv in np.unique(d["timing"][:, 0]):
        m = d["timing"][:, 0] == iv
        got = _pm_euler_gpu(d, m, iv, variant)
        worst = np.maximum(worst, H.state_err(got, d["out"][m], cols=6).max(0))
        oracle.set_trig(1)
        o = oracle.integrate(d["rows"], st[m, 0], st[m, 1], st[m, 2], None, st[m, 3], st[m, 4], np.float32(d["action"][m, 0]),
                             np.float32(d["action"][m, 1]), d["type_id"][m], None, int(iv))
        oracle.set_trig(0)
        assert np.array_equal(np.float32(o[:, :6]), got[:, :6]), np.abs(np.float32(o[:, :6]) - got[:, :6]).max(0)
        assert np.array_equal(got[:, 6:8], np.float32(d["action"][m]))
        assert np.array_equal(_pm_euler_gpu(d, m, iv, variant, through_step=True)[:, :6], got[:, :6])
    assert worst.max() <= TOL, worst


def test_pointmass_euler_mirror_and_pools_that_hold_both_backends(oracle):
    """physics.PointMass(backend="euler").step (the mirror of point_mass.py:209-232) and a pool whose table holds newton AND euler
    rows: each participant takes its own back-end; t2d_step_n on such a pool falls back to plain launches (form 'unfused')."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.physics import BatchedState, PointMass
    from tactics2d_amd.pool import ParticipantPool
    d = H.load_npz("pm_euler.npz")
    m = d["type_id"] == 1                                   # speed_range (0.5, 1.2), interval 100, delta_t 5
    st = np.float32(d["state"][m])
    model = PointMass(speed_range=(0.5, 1.2), interval=100, delta_t=5, backend="euler")
    assert model.backend == "euler" and model.model_id == L.MODEL_POINTMASS_EULER
    assert PointMass(backend="rk4").backend == "newton"     # point_mass.py:77-81
    nxt = model.step(BatchedState(0, st[:, 0], st[:, 1], st[:, 2], vx=st[:, 3], vy=st[:, 4]),
                     (np.float32(d["action"][m, 0]), np.float32(d["action"][m, 1])), 100)
    got = np.stack([nxt.x, nxt.y, nxt.heading, nxt.vx, nxt.vy], 1)
    assert np.abs(got - d["out"][m][:, [0, 1, 2, 4, 5]]).max() <= TOL
    assert nxt.frame == 100
    # newton and euler rows side by side
    newton = PointMass(speed_range=(0.5, 1.2), interval=100, delta_t=5)
    rows = np.stack([newton.param_row(), model.param_row()])
    n = int(m.sum())
    pool = ParticipantPool(n, 2)
    pool.set_param_table(rows)
    rep = lambda a: np.repeat(np.float32(a), 2)
    tid = np.tile(np.uint8([0, 1]), n)
    pool.reset(rep(st[:, 0]), rep(st[:, 1]), rep(st[:, 2]), np.zeros(2 * n, np.float32), tid, vx=rep(st[:, 3]), vy=rep(st[:, 4]))
    pool.set_actions(rep(d["action"][m, 0]), rep(d["action"][m, 1]))
    assert pool.step_form(4) == "unfused"
    pool.snapshot()
    pool.step_n(3, 100)
    three = [pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_VX, L.F_VY)]
    pool.restore(False)
    for _ in range(3):
        pool.step(100)
    assert all(np.array_equal(a, pool.download(f)) for a, f in zip(three, (L.F_X, L.F_Y, L.F_HEADING, L.F_VX, L.F_VY)))
    pool.restore(False)
    pool.step(100)
    gx, gy, gh, gvx, gvy = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_VX, L.F_VY))
    pool.close()
    oracle.set_trig(1)
    o = oracle.integrate(rows, rep(st[:, 0]), rep(st[:, 1]), rep(st[:, 2]), None, rep(st[:, 3]), rep(st[:, 4]),
                         rep(d["action"][m, 0]), rep(d["action"][m, 1]), tid, None, 100)
    oracle.set_trig(0)
    for g, c in ((gx, 0), (gy, 1), (gh, 2), (gvx, 4), (gvy, 5)):
        assert np.array_equal(np.float32(o[:, c]), g), c
    assert (gx[0::2] != gx[1::2]).any()                      # the two back-ends do differ where the speed is clipped


@pytest.mark.parametrize("variant,only", [("fast", None), ("exact", None), ("fast", "pm"), ("exact", "pm"), ("fast", "kin"), ("exact", "kin")])
def test_four_per_lane_integrator_of_large_pools_equals_the_one_per_lane_step(variant, only):
    """Pools of >= 2 M participants without a dynamics row take t2d_integrate with four consecutive participants per lane
    (16-byte loads and stores: integrate_wide_kernel).  Same arithmetic per participant: every state column equals, bit for
    bit, what the fused step's one-per-lane integrator leaves behind -- kinematic bicycles and point masses in one pool,
    inactive slots untouched."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    rows, _ = S.full_type_table()
    # (only = "pm" / "kin": every active participant has that one model -- the instantiation that carries it alone)
    rows = rows[((rows[:, L.P_MODEL] == L.MODEL_KINEMATICS) & (only != "pm")) | ((rows[:, L.P_MODEL] == L.MODEL_POINTMASS) & (only != "kin"))]
    models = rows[:, L.P_MODEL].astype(int)
    usable = np.arange(len(rows))
    n_env, A = 32768, 64
    n = n_env * A
    rng = np.random.default_rng(12)
    tid = usable[rng.integers(0, usable.size, n)].astype(np.uint8)
    active = (rng.random(n) > 0.03).astype(np.uint8)
    pm = models[tid] == L.MODEL_POINTMASS
    x, y = np.float32(rng.uniform(-100, 100, n)), np.float32(rng.uniform(-100, 100, n))
    h = np.float32(rng.uniform(0, 6.28, n))
    v = np.float32(np.where(pm, rng.uniform(0.5, 1.4, n), rng.uniform(0.0, 9.0, n)))
    vx, vy = np.float32(v * np.cos(h)), np.float32(v * np.sin(h))
    a0, a1 = np.float32(rng.uniform(-2.0, 2.0, n)), np.float32(rng.uniform(-0.3, 0.3, n))
    fields = (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_APPLIED0, L.F_APPLIED1)

    def run(fused):
        pool = ParticipantPool(n_env, A)
        pool.set_param_table(rows)
        pool.set_integrator_variant(variant)
        pool.reset(x, y, h, v, tid, active, vx=vx, vy=vy)
        pool.set_actions(a0, a1)
        (pool.step if fused else pool.integrate)(100)
        out = [pool.download(f) for f in fields]
        pool.close()
        return out
    wide, narrow = run(False), run(True)
    for f, w, nr in zip(fields, wide, narrow):
        assert np.array_equal(w.view(np.uint32), nr.view(np.uint32)), (f, int((w.view(np.uint32) != nr.view(np.uint32)).sum()))
    off = active == 0
    assert np.array_equal(wide[0][off], x[off]) and np.array_equal(wide[3][off], v[off])
    assert (wide[0][~off] != x[~off]).mean() > 0.9


@pytest.mark.parametrize("variant", ["fast", "exact"])
def test_dynamics_within_1e5_of_reference_wherever_it_is_conditioned(variant):
    """Every dyn_random case against the reference's own result, with the tolerance its conditioning allows
    (helpers.dyn_tolerance): 1e-5 where 100 ulp-sensitivities stay under 1e-6 (5731 of 6000 cases), 1e-5 + 1000 x
    sensitivity up to a sensitivity of 1 mm per ulp (221 more), and only the 48 cases beyond that reported."""
    d = H.load_npz("dyn_random.npz")
    tol, strict = H.dyn_tolerance(d["sens"], TOL)
    err4 = np.zeros((len(tol), 4))
    for iv, m in _groups(d):
        got = H.gpu_physics(d["rows"], d["type_id"][m], d["state"][m], d["action"][m], iv, variant, "dyn")
        err4[m] = H.state_err(got, d["out"][m], cols=4)
    err = err4.max(1)
    chaotic = ~np.isfinite(tol)
    print(f"dynamics {variant}: {int(strict.sum())} cases at 1e-5 (max err {err[strict].max():.3g}), "
          f"{int((~strict & ~chaotic).sum())} at the conditioning-scaled bound (max err / bound "
          f"{(err / tol)[~strict & ~chaotic].max():.3g}), {int(chaotic.sum())} chaotic in the reference itself "
          f"({int((err[chaotic] > TOL).sum())} of them beyond 1e-5)")
    assert strict.sum() > 5700 and chaotic.sum() < 60
    assert (err[strict] <= TOL).all(), err[strict].max()
    assert (err[~chaotic] <= tol[~chaotic]).all(), (err / tol)[~chaotic].max()
    # the chaotic cases (one fp64 ulp of an input moves the reference's own result by >= 1 mm) are BOUNDED too, by the same
    # K x sens: heading and speed of every one of them, and the position of every one whose trig arguments stay inside the
    # domain the deterministic sincos is specified for (|x| < 1e9 rad).  In a handful the reference drives its own slip angle
    # to 1e14 .. 1e18 rad inside the step (helpers.dyn_max_trig_argument replays its recurrence): there cos / sin of kernel
    # and oracle (bit-equal to each other: test_exact_variant_is_bit_identical_to_oracle) are not those of numpy, and the
    # position built from them is reported, not bounded
    cb = H.dyn_chaotic_bound(d["sens"], TOL)
    assert (err4[chaotic, 2:] <= cb[chaotic, None]).all(), (err4[chaotic, 2:] / cb[chaotic, None]).max()
    beyond = np.nonzero(chaotic & (err4[:, :2].max(1) > cb))[0]
    for k in beyond:
        assert H.dyn_max_trig_argument(d, int(k)) > 1e9, (int(k), err4[k], cb[k])
    assert len(beyond) <= 4, beyond
    print(f"   chaotic cases: all {int(chaotic.sum())} headings / speeds within 1e-5 + {H.SENS_SCALE:g} x sens; positions beyond it "
          f"(trig argument > 1e9 rad in the reference itself): {beyond.tolist()}")


def test_known_answers():
    kats = H.load_json("physics_kats.json")
    for k in kats:
        model = {"kinematics": "kin", "dynamics": "dyn", "pointmass": "pm"}[k["model"]]
        st = np.array([k["state"]], np.float32)
        act = np.array([k["action"]], np.float32)
        if not (np.float64(st) == np.array([k["state"]])).all():
            continue  # KAT input is not fp32-representable (e.g. speed -1e-15 is) -> skip
        row = np.array([k["row"]])
        # dynamics KATs carry the reference's conditioning at their input (oracle/gen_golden.py: conditioning())
        tol = H.dyn_tolerance([k["sens"]], TOL)[0][0] if model == "dyn" else TOL
        for variant in ("fast", "exact"):
            got = H.gpu_physics(row, np.array([0]), st, act, k["interval"], variant, model)
            want = np.array([k["out"]])
            cols = 4 if model == "dyn" else 6
            e = H.state_err(got, want, cols=cols)
            if not np.isfinite(tol):  # one ulp of an input moves the reference's own result by > 1 mm here
                print("chaotic KAT", k["state"], k["action"], variant, "sens", k["sens"], "err vs reference", e.max(0))
                continue
            assert e.max() <= tol, (k["model"], k["ctor"], variant, got, want, tol)
            if k["applied"] is not None:
                assert np.allclose(got[0, 6:8], np.float32(k["applied"]), atol=0)


def test_mod_two_pi_quirk_matches_numpy():
    # np.mod(-tiny, 2*pi) == 2*pi: the stored heading may equal fp32(2*pi)
    k = [q for q in H.load_json("physics_kats.json") if q["ctor"] == "unconstrained"][0]
    st = np.array([k["state"]], np.float32); act = np.array([k["action"]], np.float32)
    got = H.gpu_physics(np.array([k["row"]]), np.array([0]), st, act, 100, "exact", "kin")
    assert got[0, 2] == np.float32(k["out"][2]) == np.float32(2 * np.pi)


@pytest.mark.parametrize("tag", ["kin_100_5", "kin_50_3", "kin_9_5", "dyn_100_5", "dyn_50_3"])
def test_rollout_teacher_forced_and_free_running(oracle, tag):
    """Per-step parity from shared fp32 inputs (teacher forcing on the reference trajectory) and
    the drift of a free-running fp32 pool over the reference's VEHICLE_ACTION_LIST roll-out."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    r = H.load_npz("rollouts.npz")
    traj, acts, row = r[f"{tag}_traj"], r[f"{tag}_act"], r[f"{tag}_row"]
    interval = int(tag.split("_")[1])
    model = tag[:3]
    n = len(acts)
    # teacher forced: every step of the reference trajectory is one independent case
    st = np.float32(traj[:-1]); act = np.float32(acts)
    ref = H.oracle_physics(oracle, row[None], np.zeros(n, np.uint8), st, act, interval, model, trig=0)
    got = H.gpu_physics(row[None], np.zeros(n, np.uint8), st, act, interval, "fast", model)
    e = H.state_err(got, ref, cols=4)
    # the dynamics roll-outs carry the reference's conditioning per step: every step of them is well conditioned
    if model == "dyn":
        assert H.dyn_tolerance(r[f"{tag}_sens"], TOL)[1].all()
    assert e.max() <= TOL, e.max(0)
    # free running on the device, state re-rounded to fp32 every step
    pool = ParticipantPool(1, 1)
    pool.set_param_table(row[None])
    pool.reset([traj[0, 0]], [traj[0, 1]], [traj[0, 2]], [traj[0, 3]], [0])
    drift = 0.0
    for k in range(n):
        pool.set_actions([acts[k, 0]], [acts[k, 1]])
        pool.integrate(interval)
    fin = np.array([pool.download(f)[0] for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED)], np.float64)
    pool.close()
    drift = np.abs(fin - traj[-1]); drift[2] = H.ang_err(fin[2], traj[-1, 2])
    print(f"{tag}: {n} steps, teacher-forced max err {e.max(0)}, free-running fp32 drift {drift}")
    if model == "kin":
        assert drift.max() < 5e-3  # fp32 re-rounding over hundreds of steps; reported in DESIGN.md


def test_inactive_participants_are_untouched():
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    d = H.load_npz("kin_random.npz")
    pool = ParticipantPool(2, 4)
    pool.set_param_table(d["rows"][:4])
    x = np.arange(8, dtype=np.float32); act = np.array([1, 0, 1, 0, 0, 1, 1, 1], np.uint8)
    pool.reset(x, x + 1, x * 0.1, x * 0 + 3, np.zeros(8, np.uint8), act)
    pool.set_actions(np.ones(8), np.ones(8) * 0.1)
    pool.integrate(100)
    gx = pool.download(L.F_X)
    pool.close()
    assert (gx[act == 0] == x[act == 0]).all() and (gx[act == 1] != x[act == 1]).all()


@pytest.mark.parametrize("interval,delta_t", [(100, 5), (50, 3), (9, 5), (100, 7)])
def test_fast_kinematics_speed_clipping_paths(oracle, interval, delta_t):
    """The fast variant treats the clamped speed as piecewise linear in the sub-step index (linear
    until the bound is crossed, pinned afterwards).  Cases sitting on a bound, crossing it at every
    possible sub-step, starting outside the range (plain loop) and never clipping must all stay
    within 2e-6 m / 1e-6 rad of the plain 20-sub-step loop (the exact variant == oracle)."""
    from tactics2d_amd import layout as L
    rng = np.random.default_rng(42 + interval)
    k = [q for q in H.load_json("physics_kats.json") if q["ctor"] == "parking" and q["model"] == "kinematics"][0]
    rows = []
    ranges = [(-0.5, 0.5), (0.0, 1.2), (0.5, 7.0), (-7.0, 7.0), (0.0, 20.0), (2.0, 30.0)]
    for lo, hi in ranges:
        r = np.array(k["row"], np.float64)
        r[L.P_SPEED_LO], r[L.P_SPEED_HI] = lo, hi
        r[L.P_RANGE_FLAGS] = int(r[L.P_RANGE_FLAGS]) | 2
        r[L.P_ACCEL_LO], r[L.P_ACCEL_HI] = -6.0, 6.0
        r[L.P_DELTA_T_MS] = delta_t
        rows.append(r)
    rows = np.array(rows)
    n = 6000
    tid = rng.integers(0, len(ranges), n).astype(np.uint8)
    lo = rows[tid, L.P_SPEED_LO]; hi = rows[tid, L.P_SPEED_HI]
    mode = rng.integers(0, 5, n)
    u = rng.random(n)
    v = np.where(mode == 0, lo, np.where(mode == 1, hi, lo + u * (hi - lo)))          # on a bound / inside
    v = np.where(mode == 3, np.where(u < 0.5, hi - 0.02 * u, lo + 0.02 * u), v)       # about to cross
    v = np.where(mode == 4, np.where(u < 0.5, hi + 0.3 * u, lo - 0.3 * u), v)         # outside (plain loop)
    st = np.stack([rng.uniform(-200, 200, n), rng.uniform(-200, 200, n), rng.uniform(-7, 7, n), v], 1).astype(np.float32)
    act = np.stack([rng.uniform(-6, 6, n), rng.uniform(-0.6, 0.6, n)], 1).astype(np.float32)
    act[rng.random(n) < 0.1, 0] = 0.0
    ref = H.oracle_physics(oracle, rows, tid, st, act, interval, "kin", trig=1)
    exact = H.gpu_physics(rows, tid, st, act, interval, "exact", "kin")
    fast = H.gpu_physics(rows, tid, st, act, interval, "fast", "kin")
    assert np.array_equal(exact[:, :4], np.float32(ref[:, :4]))
    e = H.state_err(fast, exact, cols=4)
    # one fp32 ulp of the stored coordinate (|x| <= 256 m: 1.5e-5) can flip on a 1e-12 difference
    ulp = np.spacing(np.abs(exact[:, :2]).astype(np.float32)).max(1)
    assert (e[:, :2].max(1) <= ulp + 2e-6).all(), e.max(0)
    assert e[:, 2].max() <= 1e-6, e.max(0)
    # the closed-form speed v0 + n*ah may round to the neighbouring fp32 of the iterated sum
    assert (e[:, 3] <= np.spacing(np.abs(exact[:, 3]).astype(np.float32)) + 1e-7).all(), e.max(0)
    print("clip paths", interval, delta_t, "max err", e.max(0), "flipped-ulp cases", int((e[:, :2].max(1) > 2e-6).sum()))


@pytest.mark.parametrize("interval,delta_t", [(100, 5), (50, 3), (100, 1), (5, 5), (9, 5)])
def test_resummed_kinematics_against_the_iterated_and_the_exact_step(oracle, interval, delta_t):
    """The fast variant evaluates a kinematic step whose speed stays inside its bounds as a SERIES (the Euler sum resummed,
    t2d_integrate_dev.h resum_sums) instead of iterating it.  Tiny vehicles at walking speed make the test sharp: wheel base
    0.1 m turns 1 m/s into the step's full heading range (|a| up to and beyond the series' limit 0.5, |b| up to and beyond
    5e-3) while the fp32 store of a coordinate < 0.3 m resolves 3e-8 m -- against the exact variant (== the oracle, bit for
    bit) and against variant 2 (the same step iterated, rounds 1-4's form).  Waves are homogeneous (sorted by |a|), so that
    the wave-uniform path choice really takes the series for the small angles and the loop beyond."""
    from tactics2d_amd import layout as L
    rng = np.random.default_rng(7 + interval)
    k = [q for q in H.load_json("physics_kats.json") if q["ctor"] == "parking" and q["model"] == "kinematics"][0]
    r = np.array(k["row"], np.float64)
    r[L.P_LF], r[L.P_LR], r[L.P_WB] = 0.04, 0.06, 0.1
    r[L.P_RANGE_FLAGS] = int(r[L.P_RANGE_FLAGS]) & ~2          # speed unbounded: every lane linear
    r[L.P_ACCEL_LO], r[L.P_ACCEL_HI] = -6.0, 6.0
    r[L.P_DELTA_T_MS] = delta_t
    rows = r[None]
    n = 64 * 300
    tid = np.zeros(n, np.uint8)
    v = rng.uniform(-2.5, 2.5, n)
    steer = rng.uniform(-0.524, 0.524, n)
    order = np.argsort(np.abs(v * np.tan(steer)))              # |a| grows along the pool: homogeneous waves
    v, steer = v[order], steer[order]
    acc = rng.uniform(-6, 6, n) * rng.choice([0.0, 0.05, 1.0], n)
    st = np.stack([rng.uniform(-0.01, 0.01, n), rng.uniform(-0.01, 0.01, n), rng.uniform(0, 6.28, n), v], 1).astype(np.float32)
    act = np.stack([acc, steer], 1).astype(np.float32)
    ref = H.oracle_physics(oracle, rows, tid, st, act, interval, "kin", trig=1)
    exact = H.gpu_physics(rows, tid, st, act, interval, "exact", "kin")
    fast = H.gpu_physics(rows, tid, st, act, interval, "fast_resummed", "kin")   # (the series whatever the pool size)
    loop = H.gpu_physics(rows, tid, st, act, interval, "fast_iterated", "kin")
    assert np.array_equal(exact[:, :4], np.float32(ref[:, :4]))
    for name, got in (("resummed", fast), ("iterated", loop)):
        e = np.abs(got[:, :6].astype(np.float64) - ref[:, :6])
        ulp = np.spacing(np.abs(np.float32(ref[:, :6])))
        # half an ulp of the fp32 store + 2e-9 of arithmetic (the series' truncation bound is 2e-10 m)
        assert (e[:, :2] <= 0.5 * ulp[:, :2] + 2e-9).all(), (name, e.max(0))
        assert (e[:, 2:] <= 0.5 * ulp[:, 2:] + 1e-7).all(), (name, e.max(0))
    d = np.abs(fast[:, :2].astype(np.float64) - loop[:, :2]).max(1)
    print(f"resummed vs iterated ({interval}, {delta_t}): max |dx| {d.max():.2e}, {int((d > 0).sum())} of {n} stores differ by an ulp")
    assert d.max() <= 6e-8
