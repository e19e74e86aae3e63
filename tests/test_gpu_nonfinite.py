"""Non-finite inputs (-m gpu): RL policies do emit NaN, and the reference lets it through -- np.clip(nan, lo, hi) is nan
(physics/single_track_kinematics.py:192-193), np.clip(+-inf) the bound, np.mod(+-inf, 2 pi) nan.  What this library promises:

  * the integrators put NaN exactly where the reference does (tests/golden/nonfinite.npz, made by importing the reference;
    the exact variant equals the oracle bit for bit there too, the fast variants agree to the 1e-5 contract elsewhere);
  * BUILD-DEFINED (the reference hands such a pose to GEOS, whose answer is undefined): a participant whose pose (x, y or
    heading) is not finite takes no part in event detection -- it raises no flag and nobody collides with it -- its IoU
    events are not evaluated, as a lidar ego it sees nothing (+inf on every beam) and as an obstacle it is skipped;
  * nothing hangs, and every OTHER participant's results are bit-identical to a run without the poison -- through
    t2d_integrate, t2d_step, every form of t2d_step_n, t2d_lidar_scan and the IDM controllers.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

NAMES = {0: "kin", 1: "dyn", 2: "pm"}


def _same(got, want, tol):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if not np.array_equal(np.isnan(got), np.isnan(want)):
        return False
    inf = np.isinf(want)
    fin = np.isfinite(want)
    return np.array_equal(got[inf], want[inf]) and bool(np.all(np.abs(got[fin] - want[fin]) <= tol))


@pytest.mark.parametrize("variant", ["exact", "fast", "fast_resummed", "fast_iterated"])
def test_integrators_propagate_nonfinite_inputs_like_the_reference(oracle, variant):
    d = H.load_npz("nonfinite.npz")
    n_bad = 0
    for t in np.unique(d["type_id"]):
        m = np.nonzero(d["type_id"] == t)[0]
        model = NAMES[int(d["model"][m[0]])]
        iv = int(d["interval"][m[0]])
        cols = 4 if model == "dyn" else 6
        st, act = np.float32(d["state"][m]), np.float32(d["action"][m])
        got = H.gpu_physics(d["rows"], d["type_id"][m], st, act, iv, variant, model)
        if variant == "exact":   # == the oracle (deterministic trig), bit for bit where it is a number, NaN where it is NaN
            want = np.float32(H.oracle_physics(oracle, d["rows"], d["type_id"][m], st, act, iv, model, trig=1))
            for c in range(cols):
                assert np.array_equal(np.isnan(got[:, c]), np.isnan(want[:, c])), (model, t, c)
                ok = np.isnan(want[:, c]) | (got[:, c] == want[:, c])
                assert ok.all(), (model, t, c, got[~ok, c], want[~ok, c])
        for i, k in enumerate(m):
            ref = d["out"][k, :cols]
            # (fp32 store of the result: an ulp of the coordinate on top of the 1e-5 contract; the crawling dynamics case is
            # ill-conditioned in the reference itself)
            tol = 1e-5 + 4e-7 * np.abs(np.nan_to_num(ref, nan=0.0, posinf=0.0, neginf=0.0)).max() + (1e-4 if model == "dyn" else 0.0)
            want32 = np.float32(ref).astype(np.float64)   # (a finite fp64 result beyond fp32's range stores as inf)
            if not _same(got[i, :cols], want32, tol):
                n_bad += 1
                print(variant, model, "state", d["state"][k], "action", d["action"][k], "\n   got", got[i, :cols], "\n   ref", ref)
    assert n_bad == 0


def _poison(sc, r0, r1, envs, rng):
    """NaN / +-inf into the action ring (steps 1 and 4) and the start state of a few participants of `envs`; returns the
    participant indices touched"""
    bad = [np.nan, np.inf, -np.inf]
    touched = []
    for n, e in enumerate(envs):
        agents = rng.choice(sc.A, size=min(sc.A, 4), replace=False)
        for q, a in enumerate(agents):
            i = int(e) * sc.A + int(a)
            v = bad[(n + q) % 3]
            kind = (n + q) % 5
            if kind == 0: r0[1, i] = v
            elif kind == 1: r1[1, i] = v
            elif kind == 2: r0[4, i] = v; r1[4, i] = -v
            elif kind == 3: sc.x[i] = v
            else: sc.heading[i] = v if sc.A > 1 else np.nan
            touched.append(i)
    return np.array(touched)


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True)


import test_gpu_chain_oracle as CO  # noqa: E402


@pytest.mark.parametrize("name,make,chaining,split,form", CO.FORMS + [("step", lambda S: S.mixed(600, 64, seed=3), 0, True, None)],
                         ids=[f[0] for f in CO.FORMS] + ["plain_launches"])
def test_poisoned_participants_through_every_step_form(oracle, name, make, chaining, split, form):
    """8 steps of every form of the step launch with NaN / +-inf in the actions and the start state of a few participants of
    a few envs (exact integrator): every other env is bit-identical to the clean run, and the poisoned envs follow the
    oracle chain (integrate -> collide -> status -> auto-reset) bit for bit, NaN where it has NaN."""
    from tactics2d_amd import layout as L, scenarios as S
    sc = make(S)
    rng = np.random.default_rng(77)
    if sc.A > 1:
        sc.x = (sc.x + rng.normal(0, 1.5, sc.n)).astype(np.float32)
        sc.y = (sc.y + rng.normal(0, 1.0, sc.n)).astype(np.float32)
        sc.status.update(max_step=5)
    else:
        sc.status.update(max_step=6, no_action_max_step=3)
    r0, r1 = CO._ring(sc, CO.N_STEPS)
    _, clean = CO._gpu_fragment(sc, r0, r1, chaining, split, form)
    envs = np.sort(rng.choice(sc.n_env, size=12, replace=False))
    touched = _poison(sc, r0, r1, envs, rng)
    start, got = CO._gpu_fragment(sc, r0, r1, chaining, split, form)
    other = np.ones(sc.n_env, bool); other[envs] = False
    om = np.repeat(other, sc.A)
    for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_FLAGS):
        assert np.array_equal(got[f][om], clean[f][om]), (name, f)
    for f in (L.F_ENV_FLAGS, L.F_CNT_STEP, L.F_FRAME_MS, L.F_STATUS, L.F_REWARD):
        assert np.array_equal(got[f][other], clean[f][other]), (name, f)
    assert np.array_equal(got["record"][:, other], clean["record"][:, other])
    n_nan = 0
    for e in envs:
        steps, fin = CO._oracle_env_chain(oracle, sc, int(e), r0, r1, start["vx"], start["vy"])
        sl = slice(e * sc.A, (e + 1) * sc.A)
        for k, (st, rw) in enumerate(steps):
            word = int(st[0]) | int(st[1]) << 8 | int(st[2]) << 16 | int(st[3]) << 24
            assert int(got["record"][k, e, 1]) == word, (name, int(e), k, hex(int(got["record"][k, e, 1])), hex(word))
            grw = got["record"][k, e, 0:1].view(np.float32)[0]
            assert (np.isnan(grw) and np.isnan(rw)) or abs(float(grw) - float(rw)) <= 2e-6, (name, int(e), k, float(grw), float(rw))
        for f, key in ((L.F_X, "x"), (L.F_Y, "y"), (L.F_HEADING, "h"), (L.F_SPEED, "v")):
            assert _eq(got[f][sl], fin[key]), (name, int(e), key, got[f][sl][~(got[f][sl] == fin[key])], fin[key][~(got[f][sl] == fin[key])])
            n_nan += int(np.isnan(got[f][sl]).sum())
        nd = ~fin["is_dyn"]
        assert _eq(got[L.F_VX][sl][nd], fin["vx"][nd]) and _eq(got[L.F_VY][sl][nd], fin["vy"][nd])
        assert np.array_equal(got[L.F_FLAGS][sl], fin["flags"]), (name, int(e), got[L.F_FLAGS][sl], fin["flags"])
        assert got[L.F_ENV_FLAGS][e] == fin["env_flags"]
        assert got[L.F_CNT_STEP][e] == fin["cnt"] and got[L.F_FRAME_MS][e] == fin["frame"]
        assert np.array_equal(got[L.F_STATUS][e], steps[-1][0])
    print(name, "poisoned participants", len(touched), "non-finite state values left after 8 steps", n_nan)


@pytest.mark.parametrize("scene,part", [("parking", False), ("mixed", True)])
def test_lidar_with_poisoned_poses_equals_the_oracle(oracle, scene, part):
    """An ego whose pose is not finite sees nothing, a participant whose pose is not finite is no obstacle (build-defined,
    kernel == oracle), every other env's scan is bit-identical to the clean one."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    sc = {"parking": lambda: S.parking(300, seed0=9), "mixed": lambda: S.mixed(96, 64, seed=6)}[scene]()
    rng = np.random.default_rng(3)

    def scan(x, y, h):
        pool = ParticipantPool(sc.n_env, sc.A)
        sc.load(pool)
        pool.reset(x, y, h, sc.speed, sc.type_id, sc.active)
        pool.lidar_config(360, 20.0, part)
        pool.lidar_scan()
        out = pool.download(L.F_LIDAR)
        pool.close()
        return out

    clean = scan(sc.x, sc.y, sc.heading)
    x, y, h = sc.x.copy(), sc.y.copy(), sc.heading.copy()
    envs = rng.choice(sc.n_env, 20, replace=False)
    for n, e in enumerate(envs):
        a = 0 if n % 2 == 0 else int(rng.integers(1, sc.A)) if sc.A > 1 else 0     # the ego itself, or an obstacle participant
        [x, y, h][n % 3][e * sc.A + a] = [np.nan, np.inf, -np.inf][(n // 3) % 3]
    got = scan(x, y, h)
    want = oracle.lidar(sc.rows, sc.n_env, sc.A, 0, x, y, h, sc.type_id, sc.active, sc.static, int(part), 360, 20.0, trig=0)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    other = np.ones(sc.n_env, bool); other[envs] = False
    assert np.array_equal(got[other].view(np.uint32), clean[other].view(np.uint32))
    blind = [e for n, e in enumerate(envs) if n % 2 == 0 or sc.A == 1]
    assert np.isinf(got[blind]).all() and (got[blind] > 0).all()


def test_idm_controllers_with_poisoned_participants_equal_the_oracle(oracle):
    """IDM car following (t2d_idm_actions) on traffic with NaN / inf positions, headings and speeds: leaders and
    accelerations bit-equal to the oracle (a comparison with NaN is false: such a participant leads nobody), other envs
    identical to the clean run."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    rng = np.random.default_rng(8)
    n_env, A = 40, 32
    n = n_env * A
    lane = rng.integers(0, 3, n)
    x = np.float32(rng.uniform(-200, 200, n)); y = np.float32((lane - 1) * 3.75 + rng.normal(0, 0.2, n))
    h = np.float32(rng.normal(0, 0.02, n)); v = np.float32(rng.uniform(5, 30, n))
    act = np.ones(n, np.uint8)
    rows = np.array([[30.0, 1.5, 2.0, 1.0, 3.0, 4.0, 1.875, 120.0], [25.0, 1.2, 3.0, 1.5, 2.0, 2.5, 1.5, 80.0]])
    cid = rng.choice([0, 1, L.IDM_NONE], n, p=[0.5, 0.3, 0.2]).astype(np.uint8)
    a0 = np.float32(rng.uniform(-3, 2, n)); a1 = np.float32(rng.normal(0, 0.02, n))
    row = np.zeros((1, L.PARAM_COLS)); row[0, [L.P_LF, L.P_LR, L.P_WB, L.P_DELTA_T_MS, L.P_LENGTH, L.P_WIDTH]] = 1.2, 1.3, 2.5, 5, 4.5, 1.8

    def run(x, y, h, v):
        pool = ParticipantPool(n_env, A)
        pool.set_param_table(row)
        pool.reset(x, y, h, v, np.zeros(n, np.uint8), active=act)
        pool.set_actions(a0, a1)
        pool.set_idm(rows, cid)
        pool.idm_actions()
        out = pool.download(L.F_ACT0), pool.download(L.F_ACT1), pool.download(L.F_LEADER)
        pool.close()
        return out

    c0, c1, cl = run(x, y, h, v)
    px, py, ph, pv = x.copy(), y.copy(), h.copy(), v.copy()
    envs = rng.choice(n_env, 12, replace=False)
    for k, e in enumerate(envs):
        for q in range(3):
            i = e * A + int(rng.integers(0, A))
            [px, py, ph, pv][(k + q) % 4][i] = [np.nan, np.inf, -np.inf][(k + 2 * q) % 3]
    g0, g1, gl = run(px, py, ph, pv)
    w0, w1, wl = oracle.idm(rows, cid, n_env, A, px, py, ph, pv, act, a0, a1)
    assert np.array_equal(gl, wl)
    assert np.array_equal(g0, w0, equal_nan=True) and np.array_equal(g1, w1, equal_nan=True)
    om = np.repeat(~np.isin(np.arange(n_env), envs), A)
    assert np.array_equal(g0[om], c0[om]) and np.array_equal(gl[om], cl[om])


def test_fast_variant_steps_keep_poison_inside_its_env():
    """The default (fast) integrator through t2d_step and t2d_step_n on the metric scene's shard: with poisoned participants
    every other env is bit-identical to the clean run, the flags of a participant whose pose went non-finite are 0, and
    NaN sits exactly where the exact variant (== the oracle, test above) has it."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    out = {}
    for variant in ("fast", "exact"):
        for poisoned in (False, True):
            sc = S.mixed(4096, 64, seed=3)
            rng = np.random.default_rng(5)
            r0, r1 = CO._ring(sc, CO.N_STEPS)
            envs = np.sort(rng.choice(sc.n_env, size=40, replace=False))
            if poisoned:
                _poison(sc, r0, r1, envs, rng)
            out[variant, poisoned] = CO._gpu_fragment(sc, r0, r1, 1, True, "chain", variant=variant)[1]
    other = np.ones(4096, bool); other[envs] = False
    om = np.repeat(other, 64)
    a, b = out["fast", True], out["fast", False]
    for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS):
        assert np.array_equal(a[f][om], b[f][om]), f
    assert np.array_equal(a[L.F_STATUS][other], b[L.F_STATUS][other]) and np.array_equal(a["record"][:, other], b["record"][:, other])
    e = out["exact", True]
    n_nan = 0
    for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED):
        assert np.array_equal(np.isfinite(a[f]), np.isfinite(e[f])), f
        assert np.array_equal(np.isnan(a[f]), np.isnan(e[f])), f
        n_nan += int(np.isnan(a[f]).sum())
    bad_pose = ~(np.isfinite(a[L.F_X]) & np.isfinite(a[L.F_Y]) & np.isfinite(a[L.F_HEADING]))
    assert n_nan > 0 and (a[L.F_FLAGS][bad_pose] == 0).all()
