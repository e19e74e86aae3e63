"""GPU parity of t2d_collide / t2d_step (through the C ABI): event flags bit-exact against the
oracle's brute-force fp64 evaluation on identical fp32 inputs.  Run with -m gpu."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_env,A,extent,kw", [
    (64, 64, (60.0, 16.0), dict(n_static=6, n_lanes=3)),        # wave == env, grid + LDS staging
    (64, 64, (400.0, 15.0), dict(n_static=0, n_lanes=4)),       # highway-like, sparse
    (96, 32, (40.0, 40.0), dict(n_static=8, n_lanes=2)),        # two envs per wave
    (33, 48, (40.0, 30.0), dict(n_static=4, n_lanes=0)),        # A not a power of two, ragged grid
    (200, 8, (20.0, 12.0), dict(n_static=5, n_lanes=2)),        # grid, geometry read from global
    (300, 4, (12.0, 8.0), dict(n_static=5, n_lanes=1)),         # brute-force pairs
    (1000, 1, (30.0, 20.0), dict(n_static=12, n_lanes=0)),      # parking-like: ego vs static only
    (5, 200, (120.0, 60.0), dict(n_static=16, n_lanes=0)),      # A_pad = 256, one env per workgroup
    (40, 64, (30.0, 10.0), dict(n_static=3, n_lanes=2, with_peds=False, inactive_frac=0.5)),
])
def test_flags_bit_exact(oracle, n_env, A, extent, kw):
    rng = np.random.default_rng(n_env * 1000 + A)
    sc = H.random_scene(rng, n_env, A, extent, **kw)
    want_f, want_e = H.oracle_collide(oracle, sc)
    got_f, got_e = H.gpu_collide(sc)
    bad = np.nonzero(got_f != want_f)[0]
    assert bad.size == 0, f"{bad.size} participants differ, first {bad[:8]}: got {got_f[bad[:8]]} want {want_f[bad[:8]]}"
    assert (got_e == want_e).all()
    # the scene must actually exercise the predicates
    frac = [(want_f & b).astype(bool).mean() for b in (1, 2, 4, 8)]
    print(f"E={n_env} A={A}: flag rates dyn/static/out/lane = {np.round(frac, 3)}")
    if A > 1:
        assert 0.01 < frac[0] < 0.99
    if kw.get("n_static"):
        assert 0.005 < frac[1] < 0.99


@pytest.mark.parametrize("n_env,A", [(64, 64), (96, 32), (40, 8), (6, 150)])
def test_off_lane_is_contains_on_structured_lane_unions(oracle, n_env, A):
    """Off-lane = not union(lanes).contains(pose): abutting lanes (shared vertices), rings, crossing roads with fillets,
    frames with a hole -- bit-exact against the oracle, with bodies that have all four vertices in lanes and still
    leave the union (the half a vertex-only rule misses) present in numbers."""
    rng = np.random.default_rng(5 * n_env + A)
    sc = H.structured_lane_scene(rng, n_env, A)
    want_f, want_e = H.oracle_collide(oracle, sc)
    got_f, got_e = H.gpu_collide(sc)
    bad = np.nonzero(got_f != want_f)[0]
    assert bad.size == 0, f"{bad.size} participants differ, first {bad[:8]}: got {got_f[bad[:8]]} want {want_f[bad[:8]]}"
    assert (got_e == want_e).all()
    rate = (want_f & 8).astype(bool).mean()
    n_edge = H.count_vertices_in_but_not_contained(oracle, sc, want_f)
    print(f"E={n_env} A={A}: off-lane rate {rate:.3f}, vertices-in-but-not-contained {n_edge}")
    assert 0.1 < rate < 0.9 and n_edge >= (10 if n_env * A > 1000 else 1)


@pytest.mark.parametrize("n_env,A", [(64, 64), (120, 16), (300, 1)])
def test_polygons_of_up_to_eight_vertices(oracle, n_env, A):
    """Static obstacles and lanes with 3..8 vertices: the library cuts 5..8-gons into fans of quads (t2d_set_*_geometry),
    the oracle does the same (fan_parts); flags bit-exact, either winding."""
    rng = np.random.default_rng(31 * n_env + A)
    sc = H.polygon_scene(rng, n_env, A)
    want_f, want_e = H.oracle_collide(oracle, sc)
    got_f, got_e = H.gpu_collide(sc)
    assert np.array_equal(got_f, want_f) and np.array_equal(got_e, want_e), int((got_f != want_f).sum())
    rates = [(want_f & b).astype(bool).mean() for b in (2, 8)]
    print(f"E={n_env} A={A}: static / off-lane rates {np.round(rates, 3)}")
    assert 0.02 < rates[0] < 0.9 and 0.05 < rates[1] < 0.99


def test_geometry_kats(oracle):
    """Hand-built touching / nesting / near-miss cases (tests/golden/geometry_kats.json)."""
    kats = H.load_json("geometry_kats.json")
    for k in kats["scenes"]:
        sc = dict(rows=np.array(k["rows"]), n_env=1, A=len(k["x"]), x=np.float32(k["x"]),
                  y=np.float32(k["y"]), heading=np.float32(k["heading"]),
                  type_id=np.array(k["type_id"], np.uint8), active=np.ones(len(k["x"]), np.uint8),
                  static=H.to_csr([[np.float32(q) for q in k["static"]]]) if k["static"] else None,
                  lanes=H.to_csr([[np.float32(q) for q in k["lanes"]]]) if k["lanes"] else None,
                  boundary=np.float32([k["boundary"]]) if k["boundary"] else None, boundary_valid=None)
        got_f, _ = H.gpu_collide(sc)
        assert got_f.tolist() == k["flags"], (k["name"], got_f.tolist(), k["flags"])


def test_non_convex_polygon_is_rejected():
    from tactics2d_amd import _ffi
    from tactics2d_amd.pool import ParticipantPool
    pool = ParticipantPool(1, 1)
    dart = np.float32([[0, 0], [4, 0], [1, 1], [0, 4]])
    with pytest.raises(_ffi.GeometryError):
        pool.set_static_geometry(H.to_csr([[dart]]))
    pool.close()


def test_step_status_reward_matches_oracle(oracle):
    """t2d_step = integrate + collide + status epilogue, against oracle integrate->collide->status
    chained on the same fp32 pool state (teacher-forced on the GPU's own fp32 state)."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    rng = np.random.default_rng(7)
    n_env, A = 128, 8
    sc = H.random_scene(rng, n_env, A, (30.0, 20.0), n_static=4, n_lanes=2, with_peds=False, inactive_frac=0.0)
    kin = H.load_npz("kin_random.npz")["rows"][0].copy()
    rows = sc["rows"].copy()
    rows[:, :18] = kin[:18]  # every type drives like the medium_car rig, keeps its own shape
    sc["rows"] = rows
    N = n_env * A
    speed = rng.uniform(-2, 10, N).astype(np.float32)
    pool = ParticipantPool(n_env, A)
    pool.set_param_table(rows)
    pool.set_static_geometry(sc["static"], sc["boundary"], sc["boundary_valid"])
    pool.set_lane_geometry(sc["lanes"])
    pool.set_status_config(max_step=6, check_dynamic=1, check_off_lane=1)
    pool.set_integrator_variant("exact")
    pool.reset(sc["x"], sc["y"], sc["heading"], speed, sc["type_id"], sc["active"])
    cfg = oracle.make_config(max_step=6, check_dynamic=1, check_off_lane=1)
    cnt = np.zeros(n_env, np.int32); frame = np.zeros(n_env, np.int32)
    seen = set()
    for step in range(9):
        a0 = rng.uniform(-3, 3, N).astype(np.float32); a1 = rng.uniform(-0.5, 0.5, N).astype(np.float32)
        st = [pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED)]
        pool.set_actions(a0, a1)
        pool.step(100)
        gx, gy, gh = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        oracle.set_trig(1)
        o = oracle.integrate(rows, st[0], st[1], st[2], st[3], None, None, a0, a1, sc["type_id"], sc["active"], 100)
        oracle.set_trig(0)
        assert (np.float32(o[:, 0]) == gx).all() and (np.float32(o[:, 2]) == gh).all()
        wf, we = oracle.collide(rows, n_env, A, gx, gy, gh, sc["type_id"], sc["active"], sc["static"],
                                sc["boundary"], sc["boundary_valid"], sc["lanes"], 0)
        wst, wrw = oracle.status(cfg, n_env, A, wf, 100, cnt, frame)
        assert (pool.download(L.F_FLAGS) == wf).all()
        assert (pool.download(L.F_ENV_FLAGS) == we).all()
        assert (pool.download(L.F_CNT_STEP) == cnt).all() and (pool.download(L.F_FRAME_MS) == frame).all()
        gst = pool.download(L.F_STATUS)
        assert (gst == wst).all(), np.nonzero((gst != wst).any(1))[0][:5]
        grw = pool.download(L.F_REWARD)
        assert np.allclose(grw, wrw, rtol=0, atol=1e-9), np.abs(grw - wrw).max()
        rec = pool.download(L.F_RECORD)[step % L.RECORD_RING]   # packed 8-byte records, ring slot = step number
        assert np.array_equal(rec[:, 0].view(np.float32), grw)
        assert np.array_equal(rec[:, 1].copy().view(np.uint8).reshape(-1, 4), gst)
        seen |= set(map(tuple, gst[:, :2].tolist()))
    pool.close()
    # NORMAL, TIME_EXCEEDED, OUT_BOUND, FAILED/static, FAILED/dynamic must all have occurred
    assert {(1, 1), (3, 1), (4, 1), (6, 3)} <= seen, seen


def test_fused_auto_reset_matches_explicit_restore():
    """t2d_set_auto_reset: finished envs return to the snapshot inside the step launch; the result must
    equal step + t2d_restore(mode 1), except that status / reward keep the terminal values."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    sc = S.mixed(48, 64, seed=5)
    sc.status = dict(max_step=7, check_dynamic=1, check_off_lane=1)
    rng = np.random.default_rng(1)
    acts = [sc.sample_actions(rng) for _ in range(12)]
    pools = []
    for fused in (False, True):
        pool = ParticipantPool(sc.n_env, sc.A)
        sc.load(pool)
        if fused:
            pool.set_auto_reset(True)
        pools.append(pool)
    n_done = 0
    for a0, a1 in acts:
        for fused, pool in zip((False, True), pools):
            pool.set_actions(a0, a1)
            pool.step(100)
        st_terminal = pools[0].download(L.F_STATUS)
        rw_terminal = pools[0].download(L.F_REWARD)
        pools[0].restore(done_only=True)
        for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_IDS, L.F_CNT_STEP, L.F_FRAME_MS):
            assert np.array_equal(pools[0].download(f), pools[1].download(f), equal_nan=True), f
        assert np.array_equal(pools[1].download(L.F_STATUS), st_terminal)      # terminal status stays visible
        assert np.array_equal(pools[1].download(L.F_REWARD), rw_terminal)
        n_done += int((st_terminal[:, 2] | st_terminal[:, 3]).sum())
    assert n_done > 20
    for p in pools:
        p.close()


@pytest.mark.parametrize("variant", ["exact", "fast"])
@pytest.mark.parametrize("n_env,A", [(96, 64), (50, 32), (300, 3), (3, 200)])
def test_fused_step_equals_two_kernel_step(variant, n_env, A):
    """t2d_step as ONE launch (integrate in registers -> events -> status) must be bit-identical to
    t2d_integrate followed by t2d_check_status."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    sc = S.mixed(n_env, A, seed=31)
    sc.status = dict(max_step=4, check_dynamic=1, check_off_lane=1)
    rng = np.random.default_rng(3)
    acts = [sc.sample_actions(rng) for _ in range(6)]
    outs = []
    for fused in (False, True):
        pool = ParticipantPool(sc.n_env, sc.A)
        sc.load(pool)
        pool.set_integrator_variant(variant)
        pool.set_fused_step(fused)
        pool.set_auto_reset(True)
        for a0, a1 in acts:
            pool.set_actions(a0, a1)
            pool.step(100)
        outs.append([pool.download(f) for f in range(L.F_COUNT) if f not in (L.F_ACT0, L.F_ACT1)])
        pool.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


def test_reference_lane_rings_through_the_map_adapter(oracle):
    """A curved two-lane road given the way the reference holds it (two side polylines per lane, 40 points each:
    map/element/lane.py:125-130) cut by tactics2d_amd.mapgeom into abutting convex quads, installed, and stepped over: the
    kernel's off-lane flags equal the oracle's on the pieces AND `ring polygon contains pose` on the UNDIVIDED outline in exact
    rational arithmetic; the budget query agrees with what t2d_set_lane_geometry accepts."""
    import test_mapgeom as TM
    from tactics2d_amd import _ffi, layout as L, mapgeom as MG
    from tactics2d_amd.pool import ParticipantPool
    road = TM._curved_road(wiggle=1.5, seed=4)
    pieces = [q for left, right in road for q in MG.lanes_from_sides(left, right)]
    outline = np.concatenate([road[0][0], road[1][1][::-1]])
    n_env, A = 24, 16
    rng = np.random.default_rng(12)
    rows = H.shape_rows(with_peds=False)
    n = n_env * A
    ang = rng.uniform(-0.02, 1.32, n); rr = 60.0 + rng.uniform(-5.5, 5.5, n)
    x, y = np.float32(rr * np.cos(ang)), np.float32(rr * np.sin(ang))
    h = np.float32(ang + np.pi / 2 + rng.normal(0, 0.25, n))
    tid = rng.integers(0, len(rows), n).astype(np.uint8)
    act = np.ones(n, np.uint8)
    lanes = [pieces] * n_env
    budget = MG.geometry_budget(n_env, A, lanes=lanes)
    assert budget["fits"], budget
    pool = ParticipantPool(n_env, A)
    pool.set_param_table(rows)
    from tactics2d_amd.traffic import polygons_to_csr
    csr = polygons_to_csr(lanes)
    pool.set_lane_geometry(csr)
    pool.reset(x, y, h, np.zeros(n, np.float32), tid, act)
    pool.collide()
    got = pool.download(L.F_FLAGS)
    want, _ = oracle.collide(rows, n_env, A, x, y, h, tid, act, None, None, None, csr, 0)
    assert np.array_equal(got & L.FLAG_OFF_LANE, want & L.FLAG_OFF_LANE)
    n_in = 0
    for i in range(0, n, 3):
        pose = oracle.pose_obb(float(x[i]), float(y[i]), float(h[i]), rows[tid[i], L.P_LENGTH], rows[tid[i], L.P_WIDTH], trig=0)
        inside = TM._exact_box_in_ring(pose, outline)
        assert bool(got[i] & L.FLAG_OFF_LANE) == (not inside), i
        n_in += inside
    assert 20 < n_in < n // 3 - 20
    # a scene that does not fit the LDS record: the budget query says so, and the library keeps it in the HBM grid tier
    # (tests/test_gpu_mapgrid.py) instead of refusing it
    too_many = [pieces * 8] * n_env
    bud = MG.geometry_budget(n_env, A, lanes=too_many)
    assert not bud["fits"] and bud["tier"] == "hbm_grid"
    pool.set_lane_geometry(polygons_to_csr(too_many))
    assert pool.step_form(1) == "unfused"
    pool.collide()
    assert np.array_equal(pool.download(L.F_FLAGS), got)          # the same lanes eight times over: the same verdicts
    pool.close()
