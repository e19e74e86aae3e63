"""Maps larger than the step kernel's 32 KiB LDS record: the HBM grid tier (tactics2d_amd/csrc/t2d_mapgrid.hip).

The reference answers "which polygons are near this pose" with an STRtree over the whole map (map/element/map.py:242-329); a
lanelet of 100-point sides is ~100 convex pieces (tactics2d_amd/mapgeom.py).  A 1-km four-lane curved road with 220-point rails -- ~880 lane
pieces per env -- overflows the record (with 100-point rails, 396 pieces, it still fits at one env per workgroup); t2d_set_lane_geometry then keeps the parts in global memory behind one uniform grid per env and
the step runs as integrate -> map events -> events + status.  Checked here: it loads and steps; every participant's flags and
every env's flags equal the oracle's (which knows no tiers) bit for bit, step after step; the off-lane verdicts equal exact
`ring.contains(pose)` on the UNDIVIDED carriageway outline; and a scene small enough for the LDS record gives the same flags,
statuses and rewards through either tier."""
from fractions import Fraction as Fr

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _road(n_pts=100, radius=500.0, arc=2.0, lanes=4, width=3.75):
    """rails of a curved road centred on the origin's neighbourhood (|x|, |y| < 520 m): rail j at radius + (j - lanes / 2) width"""
    t = np.linspace(-arc / 2, arc / 2, n_pts)
    rail = lambda r: np.stack([r * np.sin(t), r * (1.0 - np.cos(t)) - (r - radius)], 1)    # passes near (0, 0) at t = 0
    return [rail(radius + (j - lanes / 2) * width) for j in range(lanes + 1)]              # rail 0 = outermost right ... (CCW turn: left = inner)


def _exact_box_in_ring(box, ring):
    F = lambda P: [(Fr(float(x)), Fr(float(y))) for x, y in P]
    a2 = lambda P: sum(P[i][0] * P[(i + 1) % len(P)][1] - P[(i + 1) % len(P)][0] * P[i][1] for i in range(len(P)))
    B, R = F(box), F(np.float32(ring))
    if a2(B) < 0:
        B = B[::-1]
    if a2(R) < 0:
        R = R[::-1]
    out = R
    for k in range(4):
        a, b = B[k], B[(k + 1) % 4]
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        nxt = []
        for i in range(len(out)):
            p, q = out[i], out[(i + 1) % len(out)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                nxt.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                nxt.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        out = nxt
        if len(out) < 3:
            return False
    return a2(out) == a2(B)


def _scene(n_env, A, seed, rails, n_static):
    """participants scattered along the road (on lanes, across lane borders, half off the carriageway, beside it), boxes beside it"""
    from tactics2d_amd import layout as L, mapgeom as MG
    from tactics2d_amd.participant import full_type_table
    rows, names = full_type_table()
    rng = np.random.default_rng(seed)
    models = rows[:, L.P_MODEL].astype(int)
    veh = np.nonzero(models == L.MODEL_KINEMATICS)[0]
    ped = np.nonzero(models == L.MODEL_POINTMASS)[0]
    lanes = []
    for j in range(len(rails) - 1):     # travel along +t: for this left-turning arc the LEFT side is the rail of smaller radius
        lanes += MG.lanes_from_sides(rails[j], rails[j + 1])
    n = n_env * A
    radius, arc = 500.0, 2.0
    t = rng.uniform(-arc / 2 * 0.98, arc / 2 * 0.98, n)
    off = np.where(rng.random(n) < 0.75, rng.uniform(-7.4, 7.4, n), rng.uniform(-12, 12, n))   # lateral offset from the centre line
    r = radius + off
    x, y = np.float32(r * np.sin(t)), np.float32(r * (1.0 - np.cos(t)) - (r - radius))
    h = np.float32(t + rng.normal(0, 0.15, n))
    tid = np.where(rng.random(n) < 0.12, ped[rng.integers(0, ped.size, n)], veh[rng.integers(0, veh.size, n)]).astype(np.uint8)
    v = np.float32(np.where(models[tid] == L.MODEL_POINTMASS, rng.uniform(0.3, 1.2, n), rng.uniform(0.0, 12.0, n)))
    active = (rng.random(n) > 0.05).astype(np.uint8)
    statics = []
    for e in range(n_env):
        polys = []
        for _ in range(n_static):
            tt = rng.uniform(-arc / 2, arc / 2); rr = radius + rng.choice([-1, 1]) * rng.uniform(6.0, 11.0)
            cx, cy = rr * np.sin(tt), rr * (1.0 - np.cos(tt)) - (rr - radius)
            a = rng.uniform(0, np.pi); hl, hw = rng.uniform(0.5, 3.0), rng.uniform(0.5, 1.5)
            c, s = np.cos(a), np.sin(a)
            polys.append(np.float32([(cx + c * lx - s * ly, cy + s * lx + c * ly) for lx, ly in ((hl, -hw), (hl, hw), (-hl, hw), (-hl, -hw))]))
        statics.append(polys)
    return dict(rows=rows, lanes=lanes, statics=statics, x=x, y=y, h=h, v=v, tid=tid, active=active)


def _load(pool, sc, n_env, lanes_per_env):
    from tactics2d_amd.traffic import polygons_to_csr
    pool.set_param_table(sc["rows"])
    static = polygons_to_csr(sc["statics"])
    lanes = polygons_to_csr(lanes_per_env)
    pool.set_static_geometry(static)
    pool.set_lane_geometry(lanes)
    pool.set_status_config(max_step=50, check_dynamic=1, check_off_lane=1)
    pool.reset(sc["x"], sc["y"], sc["h"], sc["v"], sc["tid"], sc["active"])
    return static, lanes


def test_a_one_kilometre_four_lane_road_loads_steps_and_agrees_with_the_oracle_and_with_exact_containment(oracle):
    from tactics2d_amd import layout as L, mapgeom as MG
    from tactics2d_amd.pool import ParticipantPool
    n_env, A = 6, 64
    # (the same road with 100-point rails -- 396 pieces -- still fits the record at one env per workgroup: 26 of the 32 KiB)
    assert MG.geometry_budget(n_env, A, lanes=[_scene(1, 1, 1, _road(), 0)["lanes"]] * n_env)["fits"]
    rails = _road(n_pts=220)
    sc = _scene(n_env, A, 17, rails, n_static=24)
    assert len(sc["lanes"]) >= 860                                            # 4 x 219 quads
    bud = MG.geometry_budget(n_env, A, lanes=[sc["lanes"]] * n_env)
    assert not bud["fits"] and bud["tier"] == "hbm_grid", bud
    pool = ParticipantPool(n_env, A)
    static, lanes = _load(pool, sc, n_env, [sc["lanes"]] * n_env)
    assert pool.step_form(1) == "unfused" and pool.step_form(8) == "unfused"
    rng = np.random.default_rng(3)
    outline = np.concatenate([rails[0], rails[-1][::-1]])                     # the carriageway, undivided
    is_box = sc["rows"][sc["tid"], L.P_SHAPE] == L.SHAPE_OBB
    seen_off = seen_static = seen_exact = 0
    for step in range(5):
        a0 = np.float32(rng.uniform(-2.0, 2.0, n_env * A)); a1 = np.float32(rng.uniform(-0.3, 0.3, n_env * A))
        pool.set_actions(a0, a1)
        (pool.step if step % 2 == 0 else (lambda ms: pool.step_n(1, ms)))(100)
        x, y, h = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        got, got_env = pool.download(L.F_FLAGS), pool.download(L.F_ENV_FLAGS)
        want, want_env = oracle.collide(sc["rows"], n_env, A, x, y, h, sc["tid"], sc["active"], static, None, None, lanes, 0)
        assert np.array_equal(got, want), (step, int((got != want).sum()), np.nonzero(got != want)[0][:8])
        assert np.array_equal(got_env, want_env), step
        seen_off += int((got & L.FLAG_OFF_LANE).astype(bool).sum()); seen_static += int((got & L.FLAG_COLLISION_STATIC).astype(bool).sum())
        if step == 0:   # off-lane on ~400 pieces == not outline.contains(pose), exactly, for every box of env 0
            for i in np.nonzero(is_box[:A] & (sc["active"][:A] != 0))[0]:
                Lg, W = sc["rows"][sc["tid"][i], L.P_LENGTH], sc["rows"][sc["tid"][i], L.P_WIDTH]
                pose = oracle.pose_obb(float(x[i]), float(y[i]), float(h[i]), Lg, W, trig=0)
                assert bool(got[i] & L.FLAG_OFF_LANE) == (not _exact_box_in_ring(pose, outline)), (i, float(x[i]), float(y[i]))
                seen_exact += 1
    assert seen_off > 200 and seen_static > 5 and seen_exact > 40, (seen_off, seen_static, seen_exact)
    st = pool.download(L.F_STATUS)
    assert st[:, 0].any()                                                      # the status epilogue saw the map's verdicts
    pool.close()


@pytest.mark.parametrize("idm", [False, True])
def test_grid_tier_and_lds_record_give_the_same_flags_statuses_and_rewards(idm):
    """the SAME short road through both tiers: a pool that fits the LDS record, and one whose lanes are padded with far-away
    pieces until the record overflows (the extra pieces lie 300 m from every participant: they change no verdict).  idm: every
    vehicle but the first of each env driven by an on-device IDM controller (a launch of its own ahead of the step on both tiers'
    unfused / fused paths) -- the controllers' actions, and everything behind them, agree too"""
    from tactics2d_amd import layout as L, mapgeom as MG
    from tactics2d_amd.pool import ParticipantPool
    from tactics2d_amd.controller import IDMController, install
    n_env, A = 4, 64
    rails = _road(n_pts=12, arc=0.2)                                           # 100 m of road: 4 x 11 pieces
    sc = _scene(n_env, A, 5, rails, n_static=6)
    keep = np.abs(np.arctan2(sc["x"], 500.0)) < 0.09                          # participants on the short road only
    sc["active"] = (sc["active"].astype(bool) & keep).astype(np.uint8)
    far = [np.float32(q + np.float32([0.0, 300.0])) for q in _scene(1, 1, 1, _road(n_pts=220), 0)["lanes"]]   # a road 300 m to the side
    outs = []
    for extra in ([], far):
        pool = ParticipantPool(n_env, A)
        _load(pool, sc, n_env, [sc["lanes"] + extra] * n_env)
        if idm:
            veh = (sc["rows"][sc["tid"], L.P_MODEL] != L.MODEL_POINTMASS).reshape(n_env, A)
            cid = np.full((n_env, A), L.IDM_NONE, np.uint8)
            cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
            install(pool, [IDMController(desired_speed=12.0, horizon=60.0)], cid.reshape(-1))
        pool.snapshot()
        pool.set_auto_reset(True)
        if extra:
            assert pool.step_form(1) == "unfused"
        rng = np.random.default_rng(9)
        rec = []
        for _ in range(12):
            pool.set_actions(np.float32(rng.uniform(-2, 2, n_env * A)), np.float32(rng.uniform(-0.3, 0.3, n_env * A)))
            pool.step(100)
            rec.append([pool.download(f).copy() for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_FLAGS, L.F_ENV_FLAGS, L.F_STATUS, L.F_REWARD, L.F_CNT_STEP, L.F_APPLIED0)])
        outs.append(rec)
        pool.close()
    assert not MG.geometry_budget(n_env, A, lanes=[sc["lanes"] + far] * n_env)["fits"]
    for a, b in zip(*outs):
        for u, w in zip(a, b):
            assert np.array_equal(u, w, equal_nan=True)
    flags = np.concatenate([r[3] for r in outs[0]])
    env_flags = np.concatenate([r[4] for r in outs[0]])
    assert (flags & L.FLAG_OFF_LANE).any() and (env_flags & (L.FLAG_OFF_LANE | L.FLAG_COLLISION_STATIC | L.FLAG_COLLISION_DYNAMIC)).any()


@pytest.mark.parametrize("case", ["crowded_statics", "long_vehicles_small_cells"])
def test_grid_tier_edge_cases_agree_with_the_oracle(oracle, case):
    """crowded_statics: thousands of overlapping boxes around eight poses of each env -- more (participant, part) pairs than a
    workgroup's queue holds: the decisions' launch then decides that workgroup's poses from scratch; long_vehicles_small_cells: a dense map (cells at
    the 4-m floor) under 18-m vehicles -- a pose's box spans dozens of cells and meets every part through several of them
    (taken once, in the first cell both ranges share).  Flags == oracle, bit for bit, either way."""
    from tactics2d_amd import layout as L, mapgeom as MG
    from tactics2d_amd.pool import ParticipantPool
    n_env, A = 3, 64
    rails = _road(n_pts=220)
    sc = _scene(n_env, A, 23, rails, n_static=0)
    rng = np.random.default_rng(11)
    if case == "crowded_statics":
        # 2600 boxes per env, 2400 of them piled on eight participants of the env: more pairs than any workgroup's queue holds
        for e in range(n_env):
            polys = []
            for j in range(2600):
                i = e * A + (rng.integers(0, 8) if j < 2400 else rng.integers(0, A))
                cx, cy = sc["x"][i] + rng.uniform(-1.5, 1.5), sc["y"][i] + rng.uniform(-1.5, 1.5)
                a = rng.uniform(0, np.pi); hl, hw = rng.uniform(0.3, 2.5), rng.uniform(0.3, 1.2)
                c, s = np.cos(a), np.sin(a)
                polys.append(np.float32([(cx + c * lx - s * ly, cy + s * lx + c * ly) for lx, ly in ((hl, -hw), (hl, hw), (-hl, hw), (-hl, -hw))]))
            sc["statics"][e] = polys
    else:
        rows = sc["rows"].copy()
        box = rows[:, L.P_SHAPE] == L.SHAPE_OBB
        rows[box, L.P_LENGTH] = 18.0                         # every vehicle a road train
        rows[box, L.P_WIDTH] = 2.6
        sc["rows"] = rows
        # a denser map: every lane quad of the road cut in two along its length (1752 pieces): the cell size falls to its floor
        lanes = []
        for q in sc["lanes"]:
            q = np.float32(q)
            if len(q) == 4:
                m01, m23 = (q[0] + q[1]) / 2, (q[2] + q[3]) / 2
                lanes += [np.float32([q[0], m01, m23, q[3]]), np.float32([m01, q[1], q[2], m23])]
            else:
                lanes.append(q)
        sc["lanes"] = lanes
    assert MG.geometry_budget(n_env, A, static=sc["statics"], lanes=[sc["lanes"]] * n_env)["tier"] == "hbm_grid"
    pool = ParticipantPool(n_env, A)
    static, lanes = _load(pool, sc, n_env, [sc["lanes"]] * n_env)
    assert pool.step_form(1) == "unfused"
    seen = 0
    for step in range(3):
        pool.set_actions(np.float32(rng.uniform(-2.0, 2.0, n_env * A)), np.float32(rng.uniform(-0.3, 0.3, n_env * A)))
        pool.step(100)
        x, y, h = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        got, got_env = pool.download(L.F_FLAGS), pool.download(L.F_ENV_FLAGS)
        want, want_env = oracle.collide(sc["rows"], n_env, A, x, y, h, sc["tid"], sc["active"], static, None, None, lanes, 0)
        assert np.array_equal(got, want), (case, step, int((got != want).sum()), np.nonzero(got != want)[0][:8])
        assert np.array_equal(got_env, want_env), (case, step)
        seen += int((got & (L.FLAG_COLLISION_STATIC if case == "crowded_statics" else L.FLAG_OFF_LANE)).astype(bool).sum())
    assert seen > 60, seen
    pool.close()


def test_envs_on_the_same_map_share_one_grid_and_envs_on_another_do_not_mix_with_them(oracle):
    """every env of a pool on one reference map holds the SAME polygons: the library keeps one grid and one set of registrations
    for them (build_map_grid).  Five envs on one map and one env on a map of its own (other obstacles): flags == oracle for all"""
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    n_env, A = 6, 64
    sc = _scene(n_env, A, 31, _road(n_pts=220), n_static=40)
    sc["statics"] = [sc["statics"][0]] * 3 + [sc["statics"][4]] + [sc["statics"][0]] * 2      # env 3 has obstacles of its own
    pool = ParticipantPool(n_env, A)
    static, lanes = _load(pool, sc, n_env, [sc["lanes"]] * n_env)
    assert pool.step_form(1) == "unfused"
    rng = np.random.default_rng(2)
    seen = 0
    for step in range(3):
        pool.set_actions(np.float32(rng.uniform(-2.0, 2.0, n_env * A)), np.float32(rng.uniform(-0.3, 0.3, n_env * A)))
        pool.step(100)
        x, y, h = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        got, got_env = pool.download(L.F_FLAGS), pool.download(L.F_ENV_FLAGS)
        want, want_env = oracle.collide(sc["rows"], n_env, A, x, y, h, sc["tid"], sc["active"], static, None, None, lanes, 0)
        assert np.array_equal(got, want), (step, int((got != want).sum()), np.nonzero(got != want)[0][:8])
        assert np.array_equal(got_env, want_env), step
        seen += int((got & L.FLAG_COLLISION_STATIC).astype(bool).sum())
    assert seen > 5, seen
    pool.close()
