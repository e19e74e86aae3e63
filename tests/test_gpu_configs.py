"""BASELINE.json's configs at FULL size on one MI355X, checked through size-independent properties
(the oracle cannot step half a million participants in seconds):
  * oracle spot check: a random sample of envs of the full run, bit-exact (state via the exact variant,
    flags, status) against the oracle;
  * shard consistency: a pool holding only envs [lo, hi) reproduces the big pool's slice bit for bit
    (what env-sharding across GPUs relies on);
  * permutation: re-ordering the agents inside every env permutes states and flags and changes nothing else;
  * idempotence: a second t2d_collide on the same poses returns the same flags;
  * fast vs exact integrator variants agree to 1e-6 on the stored fp32 state.
"""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

STATE = None


def _fields():
    from tactics2d_amd import layout as L
    return (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY)


def _run(sc, acts, variant="exact", collect_pre=False):
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_integrator_variant(variant)
    pre = None
    for k, (a0, a1) in enumerate(acts):
        if collect_pre and k == len(acts) - 1:
            pre = [pool.download(f) for f in _fields()] + [pool.download(L.F_CNT_STEP), pool.download(L.F_FRAME_MS)]
        pool.set_actions(a0, a1)
        pool.step(100)
    out = dict(state=[pool.download(f) for f in _fields()], flags=pool.download(L.F_FLAGS),
               env_flags=pool.download(L.F_ENV_FLAGS), status=pool.download(L.F_STATUS),
               reward=pool.download(L.F_REWARD), cnt=pool.download(L.F_CNT_STEP), pre=pre)
    pool.collide()
    out["flags_again"] = pool.download(L.F_FLAGS)
    pool.close()
    return out


def _scene(name):
    from tactics2d_amd import scenarios as S
    return {"cfg2": lambda: S.parking(4096), "cfg3": lambda: S.highway(1024, 64),
            "cfg4": lambda: S.intersection(2048, 32), "cfg5": lambda: S.mixed(8192, 64),
            "metric": lambda: S.mixed(4096, 64)}[name]()


@pytest.mark.parametrize("name", ["cfg2", "cfg3", "cfg4", "cfg5", "metric"])
def test_full_size_config(oracle, name):
    sc = _scene(name)
    # the IoU events keep per-env history and are covered step by step in tests/test_iou_events.py
    sc.status.update(check_arrival=0, check_no_action=0, shaped_reward=0)
    rng = np.random.default_rng(17)
    acts = [sc.sample_actions(rng) for _ in range(3)]
    # stress jitter (test only): scatter the start poses so that every predicate fires within 3 steps
    jx, jy = {"cfg2": (4.0, 5.0), "cfg3": (3.0, 2.5), "cfg4": (3.0, 3.0)}.get(name, (3.0, 3.0))
    sc.x = (sc.x + rng.normal(0, jx, sc.n)).astype(np.float32)
    sc.y = (sc.y + rng.normal(0, jy, sc.n)).astype(np.float32)
    sc.heading = np.mod(sc.heading + rng.normal(0, 0.3, sc.n), 2 * np.pi).astype(np.float32)
    if name == "cfg2":
        sc.speed[:] = np.where(np.arange(sc.n) % 2 == 0, 0.5, -0.5)
        sc.boundary[::3, 1] = sc.boundary[::3, 0] + 20.0      # shrink every third map: out-of-bound fires
    big = _run(sc, acts, "exact", collect_pre=True)

    # idempotence of the event kernel
    assert np.array_equal(big["flags"], big["flags_again"])
    assert np.isfinite(np.stack(big["state"][:4])).all()

    # ---- oracle spot check on a sample of envs (last step, teacher-forced on the pool's fp32 state)
    E, A = sc.n_env, sc.A
    envs = np.sort(rng.choice(E, size=min(E, 40), replace=False))
    idx = (envs[:, None] * A + np.arange(A)[None]).reshape(-1)
    pre = big["pre"]
    a0, a1 = acts[-1]
    oracle.set_trig(1)
    o = oracle.integrate(sc.rows, pre[0][idx], pre[1][idx], pre[2][idx], pre[3][idx], pre[4][idx], pre[5][idx],
                         a0[idx], a1[idx], sc.type_id[idx], sc.active[idx], 100)
    oracle.set_trig(0)
    for c in range(4):
        assert np.array_equal(np.float32(o[:, c]), big["state"][c][idx]), f"{name}: state column {c}"
    sub = sc.shard(0, E)  # full CSR; pick the sampled envs one by one for the oracle
    wf_all = []
    for e in envs:
        one = sc.shard(int(e), int(e) + 1)
        sl = slice(e * A, (e + 1) * A)
        wf, we = oracle.collide(sc.rows, 1, A, big["state"][0][sl], big["state"][1][sl], big["state"][2][sl],
                                one.type_id, one.active, one.static, one.boundary, one.boundary_valid, one.lanes, 0)
        assert np.array_equal(wf, big["flags"][sl]), f"{name}: flags of env {e}"
        assert we[0] == big["env_flags"][e]
        wf_all.append(wf)
    cfg = oracle.make_config(**sc.status)
    cnt = pre[6][envs].copy(); frame = pre[7][envs].copy()
    wst, wrw = oracle.status(cfg, len(envs), A, np.concatenate(wf_all), 100, cnt, frame)
    assert np.array_equal(wst, big["status"][envs]) and np.array_equal(cnt, big["cnt"][envs])
    assert np.allclose(wrw, big["reward"][envs], rtol=0, atol=1e-9)
    del sub

    # ---- shard consistency: envs [lo, hi) alone == the slice of the big pool
    lo, hi = E // 3, E // 3 + min(E // 4, 160)
    part = sc.shard(lo, hi)
    pacts = [(a0[lo * A:hi * A], a1[lo * A:hi * A]) for a0, a1 in acts]
    small = _run(part, pacts, "exact")
    for c in range(6):
        assert np.array_equal(small["state"][c], big["state"][c][lo * A:hi * A], equal_nan=True)
    assert np.array_equal(small["flags"], big["flags"][lo * A:hi * A])
    assert np.array_equal(small["status"], big["status"][lo:hi]) and np.array_equal(small["reward"], big["reward"][lo:hi])

    # ---- fast vs exact variant
    fast = _run(sc, acts, "fast")
    from tactics2d_amd import layout as L
    # Every participant is compared; the bound is 2e-5 (the fp32 store of both results: an ulp at |x| >= 128 m is 1.5e-5)
    # plus what the roll-out's own conditioning allows.  SingleTrackDynamics is the reference's explicit Euler of a stiff
    # tyre model: at crawling speed an fp32 ulp of the start speed can move the end state by metres.  `sens` = the oracle's
    # end-state change per fp32 ulp of an input (helpers.rollout_sensitivity); two correct integrators differ by ~1e-7 of
    # such an ulp per operation, a few hundred operations per step: 1e-4 x sens bounds it with two orders to spare.
    # Participants whose 3-step roll-out amplifies one ulp to more than 10 m (chaotic by any standard) are reported,
    # and must be rare; they are covered -- like everything -- by the exact variant's bit-equality with the oracle.
    dyn = np.nonzero((sc.rows[sc.type_id, L.P_MODEL] == L.MODEL_DYNAMICS) & sc.active.astype(bool))[0]
    sens = np.zeros(sc.n)
    if len(dyn):
        sens[dyn] = H.rollout_sensitivity(oracle, sc, acts, dyn)
    chaotic = sens > 10.0
    ok = sc.active.astype(bool) & ~chaotic
    tol = 2e-5 + 1e-4 * sens
    for c in (0, 1, 3):
        err = np.abs(fast["state"][c].astype(np.float64) - big["state"][c])
        assert (err[ok] <= tol[ok]).all(), (c, float((err[ok] / tol[ok]).max()))
    assert (H.ang_err(fast["state"][2][ok], big["state"][2][ok]) <= 1e-6 + 1e-4 * sens[ok]).all()
    if len(dyn):
        errx = np.abs(fast["state"][0].astype(np.float64) - big["state"][0])[dyn]
        print(f"{name}: dynamics participants {len(dyn)}, sens quantiles (50/90/99/max) "
              f"{np.quantile(sens[dyn], [0.5, 0.9, 0.99, 1.0])}, chaotic {int(chaotic.sum())}, worst |dx| / bound "
              f"{float((errx / tol[dyn])[~chaotic[dyn]].max()):.3f}, over 2e-5: {int((errx[~chaotic[dyn]] > 2e-5).sum())}")
        assert chaotic.sum() <= 0.01 * len(dyn), (int(chaotic.sum()), len(dyn))
    # flags may only differ where a pose moved by an fp32 ulp across a touching configuration: rare
    assert (fast["flags"] != big["flags"]).mean() < 1e-4

    # event rates: the scenes must exercise the predicates at full size
    rates = [(big["flags"] & b).astype(bool).mean() for b in (1, 2, 4, 8)]
    if A > 1:
        assert rates[0] > 0.005
    assert rates[2] > 0.002 or name not in ("cfg2",)
    if sc.static is not None and sc.static[0][-1] > 0:
        assert rates[1] > 0.001, rates
    if sc.lanes is not None:
        assert rates[3] > 0.005
        # off-lane is `not union(lanes).contains(pose)`: among the sampled envs there must be bodies with all four
        # vertices in lanes that are off-lane all the same (corner of the crossing cut, gap under the body)
        scd = dict(rows=sc.rows, n_env=E, A=A, x=big["state"][0], y=big["state"][1], heading=big["state"][2],
                   type_id=sc.type_id, active=sc.active, lanes=sc.lanes)
        n_edge = H.count_vertices_in_but_not_contained(oracle, scd, big["flags"], envs=[int(e) for e in envs])
        print(f"{name}: vertices-in-but-not-contained among {len(envs)} sampled envs: {n_edge}")
        if name in ("cfg4", "cfg5", "metric"):
            assert n_edge > 0
    print(f"{name}: E={E} A={A} flag rates dyn/static/out/lane = {np.round(rates, 4)}; "
          f"status counts = {dict(zip(*np.unique(big['status'][:, 0], return_counts=True)))}")


def test_agent_permutation_property():
    """Re-ordering the agents of every env permutes per-participant results and nothing else."""
    from tactics2d_amd import scenarios as S
    sc = S.mixed(96, 64, seed=21)
    rng = np.random.default_rng(4)
    acts = [sc.sample_actions(rng) for _ in range(4)]
    base = _run(sc, acts, "exact")
    A = sc.A
    perm = rng.permutation(A)
    perm = np.concatenate([[0], perm[perm != 0]])      # the ego (agent 0) drives the env status: keep it
    gidx = (np.arange(sc.n_env)[:, None] * A + perm[None]).reshape(-1)
    import copy
    sc2 = copy.copy(sc)
    for f in ("x", "y", "heading", "speed", "type_id", "active"):
        setattr(sc2, f, getattr(sc, f)[gidx].copy())
    acts2 = [(a0[gidx], a1[gidx]) for a0, a1 in acts]
    out = _run(sc2, acts2, "exact")
    for c in range(6):
        assert np.array_equal(out["state"][c], base["state"][c][gidx], equal_nan=True)
    assert np.array_equal(out["flags"], base["flags"][gidx])
    assert np.array_equal(out["env_flags"], base["env_flags"]) and np.array_equal(out["status"], base["status"])


def test_cfg2_full_size_with_the_iou_events_on(oracle):
    """cfg2 at its full size (4096 parking envs) with Arrival / NoAction / the shaped reward ON -- the configuration
    ParkingEnv runs -- every env against the oracle's status chain on the pool's own fp32 states: status bytes, IoU,
    reward, NoAction counter, step counter, for 8 steps (egos on the bay, egos that never move, egos that drive)."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    sc = S.parking(4096)
    n_env = sc.n_env
    rng = np.random.default_rng(3)
    tc = sc.target.mean(1)
    on = np.arange(n_env) % 4 == 0
    sc.x[on] = tc[on, 0] + rng.normal(0, 0.05, on.sum()).astype(np.float32)
    sc.y[on] = tc[on, 1] + rng.normal(0, 0.05, on.sum()).astype(np.float32)
    sc.heading[on] = sc.target_heading[on]
    still = np.arange(n_env) % 4 == 1
    sc.status.update(max_step=6, no_action_max_step=3)
    assert sc.status["check_arrival"] and sc.status["check_no_action"] and sc.status["shaped_reward"]
    pool = ParticipantPool(n_env, 1)
    sc.load(pool)
    pool.set_integrator_variant("exact")
    cfg = oracle.make_config(**sc.status)
    ep = oracle.EpisodeState(n_env, sc.target, None, np.stack([sc.x, sc.y], 1))
    cnt = np.zeros(n_env, np.int32); frame = np.zeros(n_env, np.int32)
    x, y, h, v = sc.x.copy(), sc.y.copy(), sc.heading.copy(), sc.speed.copy()
    seen = set()
    for t in range(8):
        a0, a1 = sc.sample_actions(rng)
        a0[still | on] = 0.0
        pool.set_actions(a0, a1)
        pool.step(100)
        oracle.set_trig(1)
        o = oracle.integrate(sc.rows, x, y, h, v, None, None, a0, a1, sc.type_id, sc.active, 100)
        oracle.set_trig(0)
        x, y, h, v = (np.float32(o[:, k]) for k in range(4))
        assert np.array_equal(pool.download(L.F_X), x) and np.array_equal(pool.download(L.F_HEADING), h)
        wf, _ = oracle.collide(sc.rows, n_env, 1, x, y, h, sc.type_id, sc.active, sc.static, sc.boundary, sc.boundary_valid, sc.lanes, 0)
        wst, wrw, wiou = oracle.status_ex(cfg, 1, wf, 100, cnt, frame, sc.rows, x, y, h, sc.type_id, ep)
        gst, grw, giou = pool.download(L.F_STATUS), pool.download(L.F_REWARD), pool.download(L.F_IOU)
        assert np.array_equal(pool.download(L.F_FLAGS), wf)
        assert np.array_equal(gst, wst), (t, np.nonzero((gst != wst).any(1))[0][:5])
        assert np.array_equal(np.isnan(giou), np.isnan(wiou)) and np.allclose(giou, wiou, rtol=0, atol=1e-7, equal_nan=True)
        assert np.allclose(grw, wrw, rtol=0, atol=2e-6), float(np.abs(grw - wrw).max())
        assert np.array_equal(pool.download(L.F_CNT_NO_ACTION), ep.cnt_na) and np.array_equal(pool.download(L.F_CNT_STEP), cnt)
        seen |= set(map(tuple, gst[:, :2].tolist()))
    pool.close()
    assert {(1, 1), (2, 1), (1, 5), (3, 1)} <= seen, seen     # normal, completed, no-action quirk, time exceeded


def test_fast_fused_step_at_metric_size_against_the_exact_one():
    """The default (fast) integrator inside the FUSED step at the metric size -- where the resummed kinematic step is on
    (>= two waves per SIMD) -- one step from the same state as the exact variant (== the oracle, test above): kinematic and
    point-mass participants within half an fp32 ulp + 1e-8 of it in x, y (the series' truncation bound is 2e-10 m), 1e-6 in
    heading and speed; dynamics participants within the 1e-5 contract wherever they are well conditioned (|v| >= 1 m/s);
    and the event flags of all but a handful (an ulp of a pose can flip a touching predicate) identical."""
    from tactics2d_amd import layout as L
    sc = _scene("metric")
    rng = np.random.default_rng(23)
    acts = [sc.sample_actions(rng)]
    ex = _run(sc, acts, "exact")
    fa = _run(sc, acts, "fast")
    it = _run(sc, acts, "fast_iterated")
    model = sc.rows[sc.type_id, L.P_MODEL].astype(int)
    act = sc.active.astype(bool)
    kin = act & (model != L.MODEL_DYNAMICS)
    dyn = act & (model == L.MODEL_DYNAMICS) & (np.abs(sc.speed) >= 1.0)
    assert kin.sum() > 100000 and dyn.sum() > 50000
    for name, got in (("fast", fa), ("fast_iterated", it)):
        for c in (0, 1):
            d = np.abs(got["state"][c].astype(np.float64) - ex["state"][c])
            ulp = np.spacing(np.abs(ex["state"][c]))
            assert (d[kin] <= 0.5 * ulp[kin] + 1e-8).all(), (name, c, d[kin].max())
            assert (d[dyn] <= 1e-5).all(), (name, c, d[dyn].max())
        for c in (2, 3):
            d = np.abs(got["state"][c].astype(np.float64) - ex["state"][c])
            if c == 2:
                d = np.minimum(d, 2 * np.pi - d)
            assert (d[kin] <= 1e-6).all() and (d[dyn] <= 1e-5).all(), (name, c, d[kin].max(), d[dyn].max())
        assert (got["flags"] != ex["flags"]).sum() <= 20, (name, int((got["flags"] != ex["flags"]).sum()))
    # the resummed and the iterated fast step differ by the series' truncation only: almost every fp32 store identical
    same = np.mean([(fa["state"][c][kin] == it["state"][c][kin]).mean() for c in (0, 1)])
    print(f"fast (resummed) vs fast_iterated: {100 * same:.3f} % of the kinematic x / y stores identical")
    assert same > 0.995
