"""Pin the CPU oracle's physics against golden vectors produced by IMPORTING the reference
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

import helpers as H


def _run(O, d, model, trig=0):
    out = np.empty((len(d["type_id"]), 8))
    for iv in np.unique(d["timing"][:, 0]):
        m = d["timing"][:, 0] == iv
        out[m] = H.oracle_physics(O, d["rows"], d["type_id"][m], d["state"][m], d["action"][m], int(iv), model, trig)
    return out


def test_kinematics_matches_reference(oracle):
    d = H.load_npz("kin_random.npz")
    out = _run(oracle, d, "kin")
    e = H.state_err(out, d["out"], cols=6)
    assert e.max() < 1e-12, e.max(0)
    assert np.array_equal(out[:, 6:8], d["applied"])  # np.clip of the actions, exactly


def test_pointmass_matches_reference(oracle):
    d = H.load_npz("pm_random.npz")
    out = _run(oracle, d, "pm")
    e = H.state_err(out, d["out"], cols=6)
    assert e.max() < 1e-12, e.max(0)


def test_pointmass_euler_backend_matches_reference(oracle):
    """PointMass(backend="euler") -- point_mass.py:177-207, a selectable back-end of the reference -- against fixtures made by
    importing it (oracle/gen_golden_pm_euler.py): sub-steps + remainder, both speed bounds, the re-projection onto the previous
    sub-step's heading, delta_t clamped to the interval."""
    d = H.load_npz("pm_euler.npz")
    assert (d["rows"][:, 0] == 4).all()
    st = np.float32(d["state"])
    worst = 0.0
    for iv in np.unique(d["timing"][:, 0]):
        m = d["timing"][:, 0] == iv
        oracle.set_trig(0)
        o = oracle.integrate(d["rows"], st[m, 0], st[m, 1], st[m, 2], None, st[m, 3], st[m, 4], np.float32(d["action"][m, 0]),
                             np.float32(d["action"][m, 1]), d["type_id"][m], None, int(iv))
        worst = max(worst, H.state_err(o, d["out"][m], cols=6).max())
        assert np.array_equal(o[:, 6:8], np.float64(np.float32(d["action"][m])))
    assert worst < 1e-12, worst
    # the cases exercise what they were drawn for: clipped and unclipped steps, moving and resting starts
    sp_in = np.hypot(st[:, 3], st[:, 4])
    assert (sp_in == 0).sum() > 10 and (np.abs(d["out"][:, 3] - np.hypot(st[:, 3] + d["action"][:, 0] * d["timing"][:, 0] / 1000,
                                                                        st[:, 4] + d["action"][:, 1] * d["timing"][:, 0] / 1000)) > 1e-3).sum() > 50


def test_dynamics_matches_reference_wherever_it_is_conditioned(oracle):
    """libm-mode oracle vs the reference's own fp64 results, tolerance set by the fixture's conditioning column
    (helpers.dyn_tolerance): rounding noise where the reference is conditioned, nothing asserted only where one ulp
    of an input moves the reference's own output by a millimetre or more."""
    d = H.load_npz("dyn_random.npz")
    out = _run(oracle, d, "dyn")
    e = H.state_err(out, d["out"], cols=4).max(1)
    sens = d["sens"]
    chaotic = sens >= H.SENS_CHAOTIC
    assert chaotic.sum() < 60 and (sens < 1e-12).sum() > 5000
    # libm's and numpy's trig may differ in the last bit here and there: a few ulp-sensitivities at most
    assert (e[~chaotic] <= 1e-12 + 10.0 * sens[~chaotic]).all(), (e / (1e-12 + 10.0 * sens))[~chaotic].max()
    assert np.isnan(out[:, 4]).all()  # vx, vy are None in the reference's dynamics State
    assert np.array_equal(out[:, 6:8], d["applied"])


def test_known_answers(oracle):
    for k in H.load_json("physics_kats.json"):
        model = {"kinematics": "kin", "dynamics": "dyn", "pointmass": "pm"}[k["model"]]
        row = np.array([k["row"]])
        fn = {"kin": oracle.lib().t2do_kinematics, "dyn": oracle.lib().t2do_dynamics,
              "pm": oracle.lib().t2do_pointmass}[model]
        import ctypes as C
        fn.argtypes = [np.ctypeslib.ndpointer(np.float64)] + [C.c_double] * 6 + [C.c_int, np.ctypeslib.ndpointer(np.float64)]
        out = np.empty(8)
        fn(row[0].copy(), *[float(v) for v in k["state"]], *[float(v) for v in k["action"]], int(k["interval"]), out)
        want = np.array(k["out"])
        cols = 4 if model == "dyn" else 6
        e = H.state_err(out[None], want[None], cols=cols)
        assert e.max() < 1e-9, (k["model"], k["ctor"], k["state"], out, want)
        if k["applied"] is not None:
            assert np.array_equal(out[6:8], np.array(k["applied"]))


def test_np_mod_quirk(oracle):
    """np.mod(-tiny, 2*pi) == 2*pi (SURVEY finding 10): heading may equal 2*pi exactly."""
    k = [q for q in H.load_json("physics_kats.json") if q["ctor"] == "unconstrained"][0]
    assert k["out"][2] == 2 * np.pi


@pytest.mark.parametrize("tag", ["kin_100_5", "kin_50_3", "kin_9_5", "dyn_100_5", "dyn_50_3", "dyn_9_5"])
def test_free_running_rollout_reproduces_reference(oracle, tag):
    """VEHICLE_ACTION_LIST roll-outs (reference tests/test_physics.py:65-73) in fp64, free running:
    final states of SURVEY.md 8c.  The dynamic model spends the first seconds at v = 0 (low-speed
    branch) and crosses the stiff band while accelerating; it still reproduces because libm and
    numpy agree bit-for-bit on this trajectory's tan/atan inputs."""
    import ctypes as C
    r = H.load_npz("rollouts.npz")
    traj, acts, row = r[f"{tag}_traj"], r[f"{tag}_act"], r[f"{tag}_row"].copy()
    interval = int(tag.split("_")[1])
    fn = oracle.lib().t2do_kinematics if tag.startswith("kin") else oracle.lib().t2do_dynamics
    fn.argtypes = [np.ctypeslib.ndpointer(np.float64)] + [C.c_double] * 6 + [C.c_int, np.ctypeslib.ndpointer(np.float64)]
    s = traj[0].copy(); out = np.empty(8)
    worst = 0.0
    for k in range(len(acts)):
        fn(row, s[0], s[1], s[2], s[3], acts[k, 0], acts[k, 1], interval, out)
        s = out[:4].copy()
        worst = max(worst, H.state_err(s[None], traj[k + 1][None]).max())
    assert len(acts) == {100: 285, 50: 570, 9: 3175}[interval]
    assert int(r[f"{tag}_frame"][0]) == len(acts) * interval
    assert worst < 1e-9, worst


def test_pointmass_newton_vs_euler_hausdorff_invariant(oracle):
    """Reference invariant (tests/test_physics.py:248-249): the Newton and Euler back-ends stay within
    a Hausdorff distance of 0.01 m over PEDESTRIAN_ACTION_LIST.  Checked on the oracle's restatement
    of both back-ends, and the Newton roll-out is compared with the reference's own."""
    import ctypes as C
    r = H.load_npz("rollouts.npz")
    fn = oracle.lib().t2do_pointmass
    fn.argtypes = [np.ctypeslib.ndpointer(np.float64)] + [C.c_double] * 6 + [C.c_int, np.ctypeslib.ndpointer(np.float64)]
    for k in range(6):
        traj, eul, acts, row = r[f"pm_{k}_traj"], r[f"pm_{k}_euler"], r[f"pm_{k}_act"], r[f"pm_{k}_row"].copy()
        interval = int(r[f"pm_{k}_timing"][0])
        s = np.array([10.0, 10.0, 0.0, 0.0]); out = np.empty(8)   # x, y, vx, vy (speed 0, heading 0)
        e = [10.0, 10.0, 0.0, 0.0, 0.0]                           # x, y, heading, vx, vy
        newton, euler = [s[:2].copy()], [np.array(e[:2])]
        for a in acts:
            fn(row, s[0], s[1], s[2], s[3], a[0], a[1], interval, out)
            s = np.array([out[0], out[1], out[4], out[5]])
            e = oracle.pointmass_euler(row, *e, a[0], a[1], interval)
            newton.append(s[:2].copy()); euler.append(np.array(e[:2]))
        newton, euler = np.array(newton), np.array(euler)
        assert np.abs(newton - traj[:, :2]).max() < 1e-9
        assert np.abs(euler - eul).max() < 1e-9
        # discrete Hausdorff distance between the two polylines' vertices (upper bound of the
        # continuous one the reference asserts on)
        D = np.linalg.norm(newton[:, None] - euler[None], axis=2)
        assert max(D.min(1).max(), D.min(0).max()) < 0.01


def test_deterministic_trig_mode_agrees_with_libm_mode(oracle):
    """The bit-reproducible trig the GPU 'exact' variant shares with the oracle stays within 1e-9 of
    the libm (reference-faithful) mode wherever the model is well conditioned."""
    for name, model, cols in (("kin_random.npz", "kin", 6), ("pm_random.npz", "pm", 6), ("dyn_random.npz", "dyn", 4)):
        d = H.load_npz(name)
        a = _run(oracle, d, model, 0); b = _run(oracle, d, model, 1)
        e = H.state_err(a, b, cols=cols).max(1)
        if model == "dyn":   # a few hundred ulp-sensitivities (about 80 trig calls per step, each <= 1 ulp apart)
            ok = d["sens"] < H.SENS_CHAOTIC
            assert (e[ok] <= 1e-9 + 1000.0 * d["sens"][ok]).all(), (name, (e[ok] / (1e-9 + 1000.0 * d["sens"][ok])).max())
        else:
            assert e.max() < 1e-9, (name, e.max())


def test_dynamics_ignores_the_remainder_substep(oracle):
    """Reference quirk: SingleTrackDynamics never integrates interval % delta_t (:143)."""
    ks = {k["ctor"]: k for k in H.load_json("physics_kats.json") if k["ctor"].endswith("_9_5")}
    kin = [k for k in H.load_json("physics_kats.json") if k["ctor"] == "medium_car_9_5" and k["model"] == "kinematics"][0]
    dyn = [k for k in H.load_json("physics_kats.json") if k["ctor"] == "medium_car_9_5" and k["model"] == "dynamics"][0]
    assert abs(kin["out"][3] - (2.0 + 3.0 * 0.009)) < 1e-12      # 5 ms + 4 ms remainder
    assert abs(dyn["out"][3] - (2.0 + 3.0 * 0.005)) < 1e-12      # remainder dropped
    assert ks


def test_python_loop_baseline_is_the_same_algorithm():
    """oracle/py_loop.py (the reference-call-pattern CPU baseline of bench.py) against the golden vectors."""
    from oracle import py_loop
    d = H.load_npz("kin_random.npz")
    worst = 0.0
    for k in range(0, len(d["state"]), 37):
        row = d["rows"][d["type_id"][k]]
        st, act = np.float64(np.float32(d["state"][k])), np.float64(np.float32(d["action"][k]))
        iv = int(d["timing"][k][0]) if "timing" in d else 100
        o = py_loop.kinematics_step(row, st[0], st[1], st[2], st[3], act[0], act[1], iv)
        e = np.abs(np.array(o[:4]) - d["out"][k][:4]); e[2] = min(e[2], 2 * np.pi - e[2])
        worst = max(worst, e.max())
    assert worst <= 1e-12, worst


@pytest.mark.parametrize("n,dt", [(20, 0.005), (16, 0.003), (100, 0.001), (1, 0.005), (2, 0.005), (33, 0.003)])
def test_resummed_kinematic_step_is_the_euler_sum(n, dt):
    """The series the fast kernel variant evaluates instead of iterating SingleTrackKinematics._step's sub-steps
    (single_track_kinematics.py:149-160 with the speed inside its bounds), restated in numpy with the table
    t2d_api.hip's kinematics_resum_table builds: x_n = x_0 + dt sum_k v_k cos(theta_k) to < 1e-9 m inside the range the
    kernel accepts (|a| <= 0.5, |b| <= 5e-3), against the iterated sum in extended precision."""
    from math import factorial
    D = 4
    M, m = n / 2, (n - 1) / 2
    kk = np.arange(n, dtype=np.longdouble)
    w = (kk - np.longdouble(m)) / np.longdouble(M)
    mu = [float(np.mean(w ** q)) for q in range(2 * D + 9)]
    qe = lambda i: (-1) ** i / factorial(2 * i)
    ro = lambda i: (-1) ** i / factorial(2 * i + 1)
    T = {"Q0": [qe(i) * mu[2 * i] for i in range(D + 1)], "Q4": [-0.5 * qe(i) * mu[2 * i + 4] for i in range(D + 1)],
         "Q2": [qe(i) * mu[2 * i + 2] for i in range(D + 1)], "Q6": [-qe(i) * mu[2 * i + 6] / 6 for i in range(D + 1)],
         "R4": [-ro(i) * mu[2 * i + 4] for i in range(D + 1)], "R8": [ro(i) * mu[2 * i + 8] / 6 for i in range(D + 1)],
         "R2": [ro(i) * mu[2 * i + 2] for i in range(D + 1)], "R6": [-0.5 * ro(i) * mu[2 * i + 6] for i in range(D + 1)]}

    def horner(c, x):
        acc = np.full_like(x, c[-1])
        for ci in c[-2::-1]:
            acc = acc * x + ci
        return acc

    rng = np.random.default_rng(n)
    N = 20000
    v0 = rng.uniform(-16.67, 69.44, N); acc = rng.uniform(-11, 3.2, N); delta = rng.uniform(-0.524, 0.524, N)
    lr, wb = 1.375, 2.637
    t = lr / wb * np.tan(delta); cb = 1 / np.sqrt(1 + t * t)
    ah, kh = acc * dt, np.tan(delta) / wb * cb * dt
    x0, y0, th0 = rng.uniform(-200, 200, N), rng.uniform(-200, 200, N), rng.uniform(0, 2 * np.pi, N)
    eps0, dlt = v0 * kh, ah * kh
    a = (eps0 + dlt * (m - 0.5)) * M
    b = dlt * (M * M / 2)
    a2, b2 = a * a, b * b
    ec = horner(T["Q4"], a2) * b2 + horner(T["Q0"], a2)
    es = b * (horner(T["Q6"], a2) * b2 + horner(T["Q2"], a2))
    ewc = a * b * (horner(T["R8"], a2) * b2 + horner(T["R4"], a2))
    ews = a * (horner(T["R6"], a2) * b2 + horner(T["R2"], a2))
    V = v0 + m * ah
    P, Q = V * ec + ah * M * ewc, V * es + ah * M * ews
    Th = th0 + m * eps0 + dlt * (m * (m - 1) / 2)
    xr = x0 + n * dt * (np.cos(Th) * P - np.sin(Th) * Q)
    yr = y0 + n * dt * (np.sin(Th) * P + np.cos(Th) * Q)
    th_end = Th + (eps0 + dlt * (m - 0.5)) * (m + 1) + dlt * ((m + 1) ** 2 / 2)
    x = x0.astype(np.longdouble); y = y0.astype(np.longdouble); th = th0.astype(np.longdouble); v = v0.astype(np.longdouble)
    for _ in range(n):
        x, y, th, v = x + v * dt * np.cos(th), y + v * dt * np.sin(th), th + v * kh, v + ah
    ok = (np.abs(a) <= 0.5) & (np.abs(b) <= 5e-3)
    assert ok.sum() > N // 3
    err = np.maximum(np.abs(xr - x), np.abs(yr - y)).astype(np.float64)
    assert err[ok].max() < 1e-9, err[ok].max()
    assert np.abs(th_end - th).astype(np.float64).max() < 1e-12   # the second rotation lands on theta_n


def _same_nonfinite(got, want, tol):
    """NaN exactly where `want` has NaN, +-inf equal, finite values within tol"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        return False
    fin = np.isfinite(want)
    if not np.array_equal(got[~fin & ~np.isnan(want)], want[~fin & ~np.isnan(want)]):
        return False
    return bool(np.all(np.abs(got[fin] - want[fin]) <= tol))


def test_nonfinite_inputs_propagate_like_the_reference(oracle):
    """tests/golden/nonfinite.npz (oracle/gen_golden_nonfinite.py, by importing the reference): one input of a step -- a
    state field or an action component -- is nan / +inf / -inf.  np.clip(nan) is nan (single_track_kinematics.py:192-193),
    np.clip(+-inf) the bound, np.mod(+-inf, 2 pi) nan: the oracle must put NaN exactly where the reference does and agree
    everywhere else, in both trig modes."""
    d = H.load_npz("nonfinite.npz")
    names = {0: "kin", 1: "dyn", 2: "pm"}
    n_nan = 0
    for t in np.unique(d["type_id"]):
        m = np.nonzero(d["type_id"] == t)[0]
        model = names[int(d["model"][m[0]])]
        cols = 4 if model == "dyn" else 6
        for trig in (0, 1):
            got = H.oracle_physics(oracle, d["rows"], d["type_id"][m], d["state"][m], d["action"][m], int(d["interval"][m[0]]), model, trig=trig)
            for i, k in enumerate(m):
                # (the crawling-speed dynamics case is ill-conditioned in the reference itself: 1e-9 between the trig modes)
                assert _same_nonfinite(got[i, :cols], d["out"][k, :cols], 1e-6 if model == "dyn" else 1e-9), \
                    (model, trig, d["state"][k], d["action"][k], got[i, :cols], d["out"][k, :cols])
                n_nan += int(np.isnan(d["out"][k, :cols]).any())
    assert n_nan > 100   # the fixture does exercise the propagation
