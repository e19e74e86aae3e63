"""Host-side logic that mirrors the reference's Python API (no GPU needed)."""
import numpy as np
import pytest

import helpers as H


def test_constructor_normalisation_matches_reference():
    """Ranges / delta_t are normalised exactly like the reference constructors
    (single_track_kinematics.py:87-124, single_track_dynamics.py:101-138, point_mass.py:50-75):
    rows built from reference objects by oracle/gen_golden.py vs rows built by our classes."""
    from tactics2d_amd.physics import PointMass, SingleTrackDynamics, SingleTrackKinematics
    cases = H.load_json("ctor_rows.json")
    assert len(cases) >= 60
    for c in cases:
        ra = tuple(c["range"]) if isinstance(c["range"], list) else c["range"]
        k = SingleTrackKinematics(1.2, 1.3, ra, ra, ra, c["interval"], c["delta_t"]).param_row()
        d = SingleTrackDynamics(1.2, 1.3, 1500.0, 0.7, steer_range=ra, speed_range=ra, accel_range=ra,
                                interval=c["interval"], delta_t=c["delta_t"]).param_row()
        p = PointMass(ra, ra, c["interval"], c["delta_t"]).param_row(0, 0.0, 0.0)
        assert np.array_equal(k, np.array(c["kin"])), (c["range"], c["delta_t"])
        assert np.array_equal(d, np.array(c["dyn"])), (c["range"], c["delta_t"])
        assert np.array_equal(p, np.array(c["pm"])), (c["range"], c["delta_t"])


def test_int_range_means_unbounded_for_vehicle_models():
    """Reference quirk: isinstance(5, float) is False -> speed_range=5 silently means 'no limit'."""
    from tactics2d_amd.physics import PointMass, SingleTrackKinematics
    assert SingleTrackKinematics(1.0, 1.0, speed_range=5).speed_range is None
    assert SingleTrackKinematics(1.0, 1.0, speed_range=5.0).speed_range == [-5.0, 5.0]
    assert PointMass(speed_range=(-7, 7)).speed_range == [0, 7]


def test_template_rows_reproduce_survey_appendix_b():
    from tactics2d_amd import layout as L
    from tactics2d_amd.participant import full_type_table, vehicle_row
    r = vehicle_row("medium_car", "kinematics")
    assert r[L.P_LF] == 4.284 / 2 - 0.880 and r[L.P_LR] == 4.284 / 2 - 0.767
    assert abs(r[L.P_WB] - 2.637) < 1e-12
    assert (r[L.P_STEER_LO], r[L.P_STEER_HI]) == (-0.524, 0.524)
    assert (r[L.P_SPEED_LO], r[L.P_SPEED_HI]) == (-16.67, 69.44)
    assert (r[L.P_ACCEL_LO], r[L.P_ACCEL_HI]) == (-11.0, 3.121)
    rows, names = full_type_table()
    assert rows.shape == (25, 24) and len(set(names)) == 25
    want_amax = dict(mini_car=1.929, small_car=2.480, medium_car=3.121, large_car=3.307, executive_car=3.429,
                     luxury_car=4.146, sports_coupe=5.241, multi_purpose_car=2.955, sports_utility_car=7.310)
    for n, a in want_amax.items():
        assert rows[names.index(n + ":kin"), L.P_ACCEL_HI] == a
        d = rows[names.index(n + ":dyn")]
        assert d[L.P_MODEL] == L.MODEL_DYNAMICS and d[L.P_MU] == 0.7 and d[L.P_IZ] == 1500 and d[L.P_CF] == 20.89
    ped = rows[names.index("adult_male")]
    assert ped[L.P_MODEL] == L.MODEL_POINTMASS and ped[L.P_SHAPE] == L.SHAPE_CIRCLE
    assert (ped[L.P_SPEED_LO], ped[L.P_SPEED_HI]) == (0, 7.0) and ped[L.P_WIDTH] == 0.40
    cyc = rows[names.index("cyclist")]
    assert cyc[L.P_LF] == cyc[L.P_LR] == 0.9 and (cyc[L.P_SPEED_LO], cyc[L.P_SPEED_HI]) == (0, 22.78)


def test_batched_state_derived_fields_follow_state_semantics():
    """State.velocity / State.speed laziness (participant/trajectory/state.py:135-169)."""
    from tactics2d_amd.physics import BatchedState
    s = BatchedState(0, [1.0, 2.0], [0.0, 0.0], heading=[0.0, np.pi / 2], speed=[2.0, 3.0])
    vx, vy = s.velocity
    assert np.allclose(vx, [2.0, 0.0], atol=1e-6) and np.allclose(vy, [0.0, 3.0], atol=1e-6)
    s2 = BatchedState(0, [0.0], [0.0], vx=[3.0], vy=[4.0])
    assert s2.speed[0] == 5.0 and s2.velocity[0][0] == 3.0
    assert BatchedState(0, [0.0], [0.0]).velocity is None


@pytest.mark.parametrize("builder,kw", [("parking", dict(n_env=32)), ("highway", dict(n_env=6, A=64)),
                                        ("intersection", dict(n_env=6, A=32)), ("mixed", dict(n_env=9, A=64))])
def test_scenes_are_deterministic_valid_and_start_clean(oracle, builder, kw):
    from tactics2d_amd import layout as L, scenarios as S
    a = getattr(S, builder)(**kw); b = getattr(S, builder)(**kw)
    for f in ("x", "y", "heading", "speed", "type_id"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    assert np.abs(a.x).max() < 256 and np.abs(a.y).max() < 256           # env-local coordinate contract
    assert a.rows.shape[0] <= L.MAX_TYPES
    for csr in (a.static, a.lanes):
        if csr is None:
            continue
        eo, vo, xy = csr
        for p in range(len(vo) - 1):
            assert 3 <= vo[p + 1] - vo[p] <= L.MAX_POLY_VERTS
            assert oracle.polygon_is_convex(xy[vo[p]:vo[p + 1]])
    f, _ = oracle.collide(a.rows, a.n_env, a.A, a.x, a.y, a.heading, a.type_id, a.active, a.static,
                          a.boundary, a.boundary_valid, a.lanes, 0)
    veh = a.rows[a.type_id, L.P_SHAPE] == L.SHAPE_OBB
    assert (f[veh] & (L.FLAG_COLLISION_DYNAMIC | L.FLAG_COLLISION_STATIC | L.FLAG_OUT_BOUND)).sum() == 0
    assert (f[veh] & L.FLAG_OFF_LANE).mean() < 0.02


def test_parking_scene_follows_the_reference_layout():
    """cfg1/cfg2: 1 ego + 8 static quads, ParkingEnv ranges (envs/parking.py:318-327), boundary = start /
    target -+ 13 m rounded outwards (generate_parking_lot.py:434-438)."""
    from tactics2d_amd import layout as L, scenarios as S
    sc = S.parking(16)
    assert sc.A == 1 and (np.diff(sc.static[0]) == 8).all()
    r = sc.rows[0]
    assert (r[L.P_SPEED_LO], r[L.P_SPEED_HI], r[L.P_ACCEL_LO], r[L.P_ACCEL_HI]) == (-0.5, 0.5, -2.0, 2.0)
    b = sc.boundary
    assert (b == np.round(b)).all() and ((b[:, 1] - b[:, 0]) >= 26).all()
    a0, a1 = sc.sample_actions(np.random.default_rng(0))
    assert np.abs(a0).max() <= 2.0 and np.abs(a1).max() <= 0.524      # ParkingEnv action box


def test_shard_ranges_cover_every_env_once():
    from tactics2d_amd.dist import shard_range
    for total, world in ((8192, 8), (4096, 3), (10, 4), (5, 8)):
        if total % world:   # the result gather needs equal shards: uneven splits must be asked for explicitly
            with pytest.raises(ValueError):
                shard_range(total, 0, world)
        got = [shard_range(total, r, world, allow_uneven=True) for r in range(world)]
        assert got[0][0] == 0 and got[-1][1] == total
        assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in got]
        assert max(sizes) - min(sizes) <= 1


def test_status_enums_match_the_reference_values():
    """traffic/status.py:10-61 and the C ABI's T2D_SCENARIO_* / T2D_TRAFFIC_* constants."""
    from tactics2d_amd.traffic import ScenarioStatus, TrafficStatus
    assert [s.value for s in ScenarioStatus] == [1, 2, 3, 4, 5, 6]
    assert (ScenarioStatus.TIME_EXCEEDED, ScenarioStatus.OUT_BOUND, ScenarioStatus.FAILED) == (3, 4, 6)
    assert (TrafficStatus.COLLISION_STATIC, TrafficStatus.COLLISION_DYNAMIC, TrafficStatus.OFF_LANE) == (3, 4, 6)
    import re, os
    h = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "t2d.h")).read()
    for name, val in re.findall(r"#define\s+T2D_SCENARIO_(\w+)\s+(\d+)", h):
        assert ScenarioStatus[name] == int(val)
    for name, val in re.findall(r"#define\s+T2D_TRAFFIC_(\w+)\s+(\d+)", h):
        if name.endswith("_QUIRK"):   # parking.py:373 stores ScenarioStatus.NO_ACTION in traffic_status
            assert int(val) == ScenarioStatus.NO_ACTION
            continue
        assert TrafficStatus[name] == int(val)


def test_action_space_and_discrete_table():
    """ParkingEnv action box (envs/parking.py:130-139) and discrete table (:95)."""
    from tactics2d_amd.envs import MAX_ACCEL, MAX_STEER, Box, VecParkingEnv
    b = Box([-MAX_STEER, -MAX_ACCEL], [MAX_STEER, MAX_ACCEL])
    assert b.contains(np.float32([0.524, -2.0])) and not b.contains(np.float32([0.53, 0.0]))
    assert VecParkingEnv._discrete_actions == {1: (0, 0), 2: (-0.5, 0), 3: (0.5, 0), 4: (0, 1), 5: (0, -1)}


def test_batched_trajectory_follows_the_reference_rules(caplog):
    """participant/trajectory/trajectory.py:97-188 restated for a batch (scope row a7)."""
    import logging
    from tactics2d_amd.physics import BatchedState
    from tactics2d_amd.history import BatchedTrajectory
    t = BatchedTrajectory(id_=7)
    assert len(t) == 0 and t.initial_state is None and t.last_state is None and t.first_frame is None and t.last_frame is None
    with pytest.raises(ValueError):
        t.add_state("not a state")
    s = lambda f, v: BatchedState(frame=f, x=[f * 0.1, 1.0], y=[0.0, 2.0], heading=[0.0, 0.0], speed=[v, 2 * v])
    t.add_state(s(0, 1.0)); t.add_state(s(100, 2.0)); t.add_state(s(200, 3.0))
    assert t.frames == [0, 100, 200] and t.stable_freq and t.has_state(100) and not t.has_state(50)
    assert t.get_state().frame == 200 and t.get_state(100).speed[0] == 2.0
    with pytest.raises(KeyError):
        t.get_state(50)
    with pytest.raises(KeyError):
        t.add_state(s(150, 9.0))                      # earlier than the last stamp
    with caplog.at_level(logging.WARNING):
        t.add_state(s(350, 4.0))                      # uneven interval
    assert not t.stable_freq and "uneven" in caplog.text
    assert np.allclose(t.average_speed, [2.5, 5.0])
    tr = t.get_trace((100, 200))
    assert len(tr) == 2 and np.allclose(tr[0][0], [10.0, 1.0])
    t.reset(keep_history=True)
    assert t.get_state().frame == 0 and len(t) == 4
    t.reset()
    assert t.frames == [0] and t.get_state().frame == 0
    t.reset(s(500, 1.0))
    assert t.frames == [500]


def test_batched_state_keeps_the_reference_state_semantics():
    """Row a6: derived speed / velocity / accel / acceleration and the typed __setattr__ of `State`
    (participant/trajectory/state.py:108-204) against vectors produced by the reference itself
    (oracle/gen_golden_state.py -> tests/golden/state_kats.json)."""
    from tactics2d_amd.physics import BatchedState
    k = H.load_json("state_kats.json")
    for case in k["derived"]:
        kw = {key: [val] for key, val in case["kw"].items()}
        s = BatchedState(0, x=[1.0], y=[2.0], **kw)
        for name in ("speed", "velocity", "accel", "acceleration"):
            got, want = getattr(s, name), case[name]
            if want is None:
                assert got is None, (case["kw"], name, got)
                continue
            got = np.asarray(got, np.float64).reshape(-1)
            assert np.allclose(got, want, rtol=4e-16, atol=0), (case["kw"], name, got, want)
    # the accel quirk: a state that only carries the scalar reports ||accel (cos h, sin h)||, not the scalar
    q = k["quirk"]
    s = BatchedState(0, x=[0.0], y=[0.0], heading=[np.float32(q["heading"])], accel=[q["accel_in"]])
    assert q["accel_out"] == 0.9999999999999999 and abs(s.accel[0] - 1.0) < 3e-16 and s._accel[0] == 1.0
    s = BatchedState(0, x=[0.0, 1.0], y=[0.0, 0.0], heading=[0.3, 2.0], accel=[-2.5, 4.0])
    assert np.allclose(s.accel, [2.5, 4.0], rtol=3e-16)                  # |accel|: the sign is gone, as in the reference
    # typed __setattr__: coercion, None, and the reference's ValueError text
    s = BatchedState("12", x=[0.0, 1.0], y=3)                            # frame "12" -> 12, y scalar fills the batch
    assert s.frame == 12 and isinstance(s.frame, int) and s.y.tolist() == [3.0, 3.0] and s.x.dtype == np.float32
    s.frame = 7.9
    assert s.frame == 7
    for c in k["coerced"]:
        if c["name"] == "frame":
            s.frame = eval(c["value"])
            assert s.frame == c["stored"]
    for e in k["errors"]:
        if not e["raised"]:
            setattr(s, e["name"], eval(e["value"]))
            assert getattr(s, e["name"]) is None
            continue
        if e["name"] == "heading":      # a list is one scalar too many for the reference, and a column here
            continue
        with pytest.raises(ValueError) as ei:
            setattr(s, e["name"], eval(e["value"]))
        assert str(ei.value) == e["message"]
    with pytest.raises(ValueError):
        s.vx = [1.0, 2.0, 3.0]                                              # not a column of this batch
    # setters (state.py:206-223)
    s2 = BatchedState(5, x=[0.0], y=[0.0], vx=[3.0], vy=[4.0])
    s2.set_accel([1.0], [-2.0])
    assert abs(s2.accel[0] - k["setters"]["accel_after_set_accel"]) < 1e-15 and abs(s2._accel[0] - k["setters"]["_accel"]) < 1e-6
    s3 = BatchedState(5, x=[1.0], y=[2.0], heading=[0.5], speed=[3.0])
    s3.set_velocity([1.0], [1.0])
    assert [float(v[0]) for v in s3.velocity] == k["setters"]["velocity_after_set_velocity"]
    assert float(s3.speed[0]) == k["setters"]["speed_after_set_velocity"] == 3.0


def test_host_frame_views_and_in_use_tracking():
    """pool.HostFrame (the Gym-API host path): section views of a frame laid out by t2d_frame_config, and the reference-count
    test that lets a pinned frame be handed out again without a copy only when nobody holds a view of it."""
    from tactics2d_amd._ffi import FrameLayout
    from tactics2d_amd.pool import HostFrame
    n, beams = 5, 7
    lay = FrameLayout()
    off = 256
    for name, nb in (("off_rel", 24 * n), ("off_obs", 24 * n), ("off_reward", 4 * n), ("off_status", 4 * n), ("off_iou", 4 * n),
                     ("off_frame_ms", 4 * n), ("off_cnt_step", 4 * n), ("off_episode", 4 * n), ("off_target_heading", 8 * n),
                     ("off_target", 32 * n), ("off_lidar", 4 * n * beams)):
        setattr(lay, name, off)
        off += (nb + 255) & ~255
    lay.bytes, lay.n_env, lay.n_beams = off, n, beams
    base = np.zeros(off, np.uint8)
    fr = HostFrame(base, lay)
    assert fr.in_use()                 # not calibrated yet: the safe answer
    del base
    fr.calibrate()
    assert fr.obs.shape == (n, 6) and fr.rel.shape == (n, 3) and fr.target.shape == (n, 4, 2) and fr.lidar.shape == (n, beams)
    assert fr.status.shape == (n, 4) and fr.terminated.dtype == np.bool_ and fr.reward.shape == (n,)
    fr.status[2, 2] = 1
    assert fr.terminated.tolist() == [False, False, True, False, False]
    assert not fr.in_use()
    held = fr.obs                      # the section view itself
    assert fr.in_use()
    del held
    assert not fr.in_use()
    col = fr.obs[:, 0]                 # a view of a view: its .base is the frame's memory
    assert fr.in_use()
    del col
    info = {"state": {"x": fr.obs[:, 0]}, "lidar": fr.lidar}
    assert fr.in_use()
    del info
    t = fr.terminated
    assert fr.in_use()
    del t
    assert not fr.in_use()
    own = fr.copy(lidar=False)         # owns its memory, no lidar section
    assert own.lidar is None and own.obs.base is not fr.base and np.array_equal(own.status, fr.status)
    x = float(fr.obs[0, 0])            # scalars do not hold the frame
    assert not fr.in_use() and x == 0.0
    kept = [fr]                        # the HostFrame object itself, kept by a caller: no array reference, the frame is held
    assert fr.in_use()
    del kept
    assert not fr.in_use()


def test_a_kept_host_frame_object_is_not_handed_out_again():
    """ParticipantPool._pick_frame / _frame (no device needed: the frames are stood in by host arrays): a caller that keeps the
    HostFrame OBJECTS of several steps -- not views of their arrays -- never sees one of them filled again (round-5 advice)."""
    from tactics2d_amd._ffi import FrameLayout
    from tactics2d_amd.pool import HostFrame, ParticipantPool
    n = 3
    lay = FrameLayout()
    off = 256
    for name in ("off_rel", "off_obs", "off_reward", "off_status", "off_iou", "off_frame_ms", "off_cnt_step", "off_episode"):
        setattr(lay, name, off)
        off += 256
    lay.off_target = lay.off_target_heading = lay.off_lidar = -1
    lay.bytes, lay.n_env, lay.n_beams = off, n, 0
    pool = ParticipantPool.__new__(ParticipantPool)      # the bookkeeping alone
    pool.frame_layout, pool.n_frames, pool._frames, pool._frame_turn = lay, 4, [None] * 4, 0
    mem = [np.zeros(off, np.uint8) for _ in range(4)]

    def step_host(fresh=True):                           # what ParticipantPool.step_host does around the library call
        k, must_copy = pool._pick_frame(fresh)
        if pool._frames[k] is None:
            fr = HostFrame(mem[k], lay)
            pool._frames[k] = fr
            fr.calibrate()
        fr = pool._frames[k]
        fr.reward[:] += 1.0                              # "the library filled frame k"
        return (fr.copy() if must_copy else fr), k

    a, ka = step_host()
    b, kb = step_host()
    assert a is not b and ka != kb                       # `a` is held as an OBJECT: its frame is not picked again
    c, kc = step_host()
    d, kd = step_host()                                  # every frame but the last is held: the last one is filled and copied out
    assert len({ka, kb, kc}) == 3 and kd == 3 and d.base is not mem[3]
    e, ke = step_host()
    assert ke == 3 and e is not d and d.reward[0] == 1.0 and e.reward[0] == 2.0
    del a
    f, kf = step_host()
    assert kf == ka                                      # released: handed out again without a copy
    del pool._frames, pool


def test_parameter_tables_equal_the_reference_module():
    """tests/golden/templates.json = the numeric entries of the three dicts of the reference's participant_template.py, read by
    loading that FILE (oracle/gen_golden_tables.py): the build's own tables (tactics2d_amd/participant.py) hold the same numbers,
    field by field -- every template, none missing, none extra"""
    import json, os
    from tactics2d_amd import participant as P
    t = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "templates.json")))
    veh_fields = ("length", "width", "height", "wheel_base", "front_overhang", "rear_overhang", "kerb_weight", "max_speed", "0_100_km/h", "max_decel")
    assert set(t["vehicle"]) == set(P.VEHICLE_TEMPLATE) and len(t["vehicle"]) == 9
    for name, row in t["vehicle"].items():
        assert tuple(row[f] for f in veh_fields) == tuple(P.VEHICLE_TEMPLATE[name]), name
        assert set(row) == set(veh_fields), (name, sorted(row))
    cyc_fields = ("length", "width", "height", "max_steer", "max_speed", "max_accel", "max_decel")
    assert set(t["cyclist"]) == set(P.CYCLIST_TEMPLATE)
    for name, row in t["cyclist"].items():
        assert tuple(row[f] for f in cyc_fields) == tuple(P.CYCLIST_TEMPLATE[name]) and set(row) == set(cyc_fields), name
    ped_fields = ("length", "width", "height", "max_speed", "max_accel")
    assert set(t["pedestrian"]) == set(P.PEDESTRIAN_TEMPLATE)
    for name, row in t["pedestrian"].items():
        assert tuple(row[f] for f in ped_fields) == tuple(P.PEDESTRIAN_TEMPLATE[name]) and set(row) == set(ped_fields), name


def test_vehicle_rows_and_poses_equal_the_reference_methods_executed(oracle):
    """tests/golden/vehicle_templates_loaded.json: Vehicle.load_from_template and Vehicle.get_pose (participant/element/vehicle.py:
    179-221, 263-281) EXECUTED by oracle/gen_golden_tables.py on a blank holder per template -- max_accel (the rounded 0-100 km/h
    rule), speed / accel ranges, the bounding box and its vertex ORDER, and six poses each (the matrix the method hands to
    shapely.affinity.affine_transform, applied by that function's documented rule).  The build's parameter rows and the oracle's
    pose (what every kernel's pose phase is bit-identical to) agree: ranges exactly, vertices in the same order to <= 2e-13 m."""
    import json, os
    from tactics2d_amd import layout as L, participant as P
    v = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vehicle_templates_loaded.json")))
    assert set(v) == set(P.VEHICLE_TEMPLATE)
    # (a quirk of the reference kept on record: this one name is also an EPA class, mapped to a template that does not exist)
    assert [n for n in v if "error" in v[n]] == ["multi_purpose_car"]
    n_pose = 0
    for name, want in v.items():
        if "error" in want:
            continue
        r = P.vehicle_row(name, "kinematics")
        assert r[L.P_ACCEL_HI] == want["max_accel"] and [r[L.P_ACCEL_LO], r[L.P_ACCEL_HI]] == want["accel_range"], name
        assert [r[L.P_SPEED_LO], r[L.P_SPEED_HI]] == want["speed_range"], name
        Ln, W = P.VEHICLE_TEMPLATE[name][:2]
        assert want["bbox"] == [[0.5 * Ln, -0.5 * W], [0.5 * Ln, 0.5 * W], [-0.5 * Ln, 0.5 * W], [-0.5 * Ln, -0.5 * W]], name
        for p in want["poses"]:
            c, s = np.cos(p["heading"]), np.sin(p["heading"])
            assert p["matrix"] == [c, -s, s, c, p["x"], p["y"]]
            got = np.asarray(oracle.pose_obb(p["x"], p["y"], p["heading"], Ln, W, trig=1)).reshape(4, 2)
            assert np.abs(got - np.asarray(p["pose"])).max() <= 2e-13, (name, got, p["pose"])
            n_pose += 1
    assert n_pose == 48


def test_batched_trajectory_replays_the_reference_operation_by_operation():
    """tests/golden/trajectory_kats.json: 33 scripted sequences (457 operations, 61 of them raising) run on the reference's own
    `Trajectory` by oracle/gen_golden_trajectory.py -- add_state (duplicates, earlier stamps, uneven intervals, a non-State),
    get_state, has_state, get_trace, the three reset modes -- with what the reference answered after EVERY operation: the value or
    the exception's type, frames, len, stable_freq, first / last / current / initial frame, average_speed.  BatchedTrajectory (a
    batch of two: the second participant carries the same numbers shifted) answers the same."""
    import warnings
    from tactics2d_amd.physics import BatchedState
    from tactics2d_amd.history import BatchedTrajectory
    seqs = H.load_json("trajectory_kats.json")
    n_ops = 0
    mk = lambda frame, x, y, speed: BatchedState(frame=frame, x=[x, x + 1.0], y=[y, y - 1.0], heading=[0.0, 0.0], speed=[speed, 2.0 * speed])
    for si, seq in enumerate(seqs):
        t = BatchedTrajectory(id_=3)
        for oi, rec in enumerate(seq):
            op = rec["op"]; kind = op[0]; got = None; raised = None
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    if kind == "add":
                        t.add_state(mk(*op[1:]))
                    elif kind == "add_bad":
                        t.add_state("not a state")
                    elif kind == "get":
                        s = t.get_state(op[1])
                        got = None if s is None else dict(frame=s.frame, speed=float(s.speed[0]), x=float(s.x[0]))
                    elif kind == "has":
                        got = bool(t.has_state(op[1]))
                    elif kind == "trace":
                        tr = t.get_trace(None if op[1] is None else tuple(op[1]))
                        got = [[float(p[0][0]), float(p[1][0])] for p in tr]     # (a location is (x[batch], y[batch]))
                    elif kind == "reset":
                        t.reset(None if op[1] is None else mk(*op[1]), keep_history=op[2])
            except Exception as exc:   # noqa: BLE001 -- compared by type below
                raised = type(exc).__name__
            where = (si, oi, op)
            assert raised == rec.get("raises"), (where, raised, rec.get("raises"))
            if raised is None and kind in ("get", "has", "trace"):
                want = rec["result"]
                if kind == "get" and want is not None:
                    assert got["frame"] == want["frame"] and got["speed"] == np.float32(want["speed"]) and got["x"] == np.float32(want["x"]), (where, got, want)
                elif kind == "trace":
                    assert np.allclose(got, want, rtol=1e-6, atol=0) if want else got == [], (where, got, want)
                else:
                    assert got == want, (where, got, want)
            a = rec["after"]
            cur = t.get_state()
            assert list(t.frames) == a["frames"] and len(t) == a["n"] and bool(t.stable_freq) == a["stable_freq"], (where, list(t.frames), a)
            assert t.first_frame == a["first_frame"] and t.last_frame == a["last_frame"], where
            assert (None if cur is None else cur.frame) == a["current_frame"], (where, a)
            assert (None if t.initial_state is None else t.initial_state.frame) == a["initial_frame"], where
            assert (None if t.last_state is None else t.last_state.frame) == a["last_state_frame"], where
            if a["average_speed"] is not None and len(t):
                assert np.allclose(np.asarray(t.average_speed)[0], a["average_speed"], rtol=1e-6), (where, t.average_speed, a["average_speed"])
            n_ops += 1
    assert n_ops == 457


def test_env_constants_and_status_enums_equal_the_reference_literals():
    """tests/golden/parking_constants.json: the literal constants of envs/parking.py (read from the parsed file: the module needs
    gymnasium) and the two enums of traffic/status.py (loaded as a file) -- the mirror's action limits, discrete actions, default
    step limit, lidar beams / range and every status name and value are the reference's"""
    import inspect
    from tactics2d_amd import envs as E, traffic as T, layout as L
    c = H.load_json("parking_constants.json")
    assert (E.MAX_STEER, E.MAX_ACCEL) == (c["MAX_STEER"], c["MAX_ACCEL"])
    assert (E.VecParkingEnv._max_steer, E.VecParkingEnv._max_accel) == (c["ParkingEnv._max_steer"], c["ParkingEnv._max_accel"])
    assert {str(k): list(v) for k, v in E.VecParkingEnv._discrete_actions.items()} == c["ParkingEnv._discrete_actions"]
    sig = inspect.signature(E.VecParkingEnv.__init__).parameters
    d = c["ParkingEnv.__init__.defaults"]
    assert sig["max_step"].default == d["max_step"] and sig["continuous"].default == d["continuous"] and sig["type_proportion"].default == d["type_proportion"]
    assert sig["lidar_beams"].default == c["ParkingEnv._ParkingScenarioManager._lidar_line"]
    assert {e.name: int(e) for e in T.ScenarioStatus} == c["ScenarioStatus"]
    assert {e.name: int(e) for e in T.TrafficStatus} == c["TrafficStatus"]
    # the C ABI's status codes are the same numbers (include/t2d.h)
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "t2d.h")).read()
    abi = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define T2D_((?:SCENARIO|TRAFFIC)_[A-Z_]+)\s+(\d+)", hdr)}
    for name, val in c["ScenarioStatus"].items():
        assert abi["SCENARIO_" + name] == val, name
    for name in ("NORMAL", "COLLISION_STATIC", "COLLISION_DYNAMIC", "OFF_LANE"):
        assert abi["TRAFFIC_" + name] == c["TrafficStatus"][name], name
    assert abi["TRAFFIC_NO_ACTION_QUIRK"] == c["ScenarioStatus"]["NO_ACTION"]      # parking.py:373
