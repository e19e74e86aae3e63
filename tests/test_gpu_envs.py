"""The Gym-style / ScenarioManager mirrors running on the MI355X (through the C ABI)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vec_parking_env_step_contract(oracle):
    from tactics2d_amd import layout as L
    from tactics2d_amd.envs import InvalidAction, VecParkingEnv
    from tactics2d_amd.traffic import ScenarioStatus, TrafficStatus
    env = VecParkingEnv(256, max_step=50, seed=3)
    obs, infos = env.reset()
    assert obs.shape == (256, 6) and obs.dtype == np.float32
    assert infos["lidar"].shape == (256, 360) and np.isfinite(infos["lidar"]).mean() > 0.3
    assert (infos["scenario_status"] == ScenarioStatus.NORMAL).all()
    with pytest.raises(InvalidAction):
        env.step(np.tile([0.6, 0.0], (256, 1)))              # steering outside +-0.524 (parking.py:235-236)
    rng = np.random.default_rng(0)
    seen = set()
    sc = env._scene
    x, y, h, v = sc.x.copy(), sc.y.copy(), sc.heading.copy(), sc.speed.copy()
    for t in range(60):
        act = env.action_space.sample(rng, 256)
        act[:, 1] = np.abs(act[:, 1])                         # keep accelerating: reach walls / boundary
        obs, reward, terminated, truncated, infos = env.step(act)
        # physics parity of the env step against the oracle (teacher-forced on the pool's fp32 state)
        o = oracle.integrate(sc.rows, x, y, h, v, None, None, act[:, 1], act[:, 0], sc.type_id, sc.active, 100)
        assert np.abs(obs[:, :4] - o[:, :4]).max() <= 1e-5
        x, y, h, v = obs[:, 0].copy(), obs[:, 1].copy(), obs[:, 2].copy(), obs[:, 3].copy()
        assert reward.shape == (256,) and terminated.dtype == bool and truncated.dtype == bool
        st = infos["scenario_status"]; tr = infos["traffic_status"]
        assert (truncated == ((st != 1) | (tr != 1))).all()   # parking.py:247-248
        assert (reward[st == ScenarioStatus.TIME_EXCEEDED] == -1).all()
        assert (reward[st == ScenarioStatus.OUT_BOUND] == -5).all()
        assert (reward[tr == TrafficStatus.COLLISION_STATIC] == -5).all()
        assert (terminated == (st == ScenarioStatus.COMPLETED)).all()
        assert (reward[terminated] == 5).all()
        normal = (st == 1) & (tr == 1)
        assert (reward[normal] > -0.002).all() and (reward[normal] < 0.2).all()   # time penalty + shaping
        seen |= set(zip(st.tolist(), tr.tolist()))
    assert (3, 1) in seen and (1, 1) in seen                  # time exceeded after 50 steps
    env.close()


def test_parking_env_adapter_returns_the_reference_5_tuple():
    from tactics2d_amd.envs import InvalidAction, ParkingEnv
    from tactics2d_amd.traffic import ScenarioStatus, TrafficStatus
    env = ParkingEnv(max_step=20000, seed=1)
    obs, infos = env.reset()
    assert obs.shape == (6,)
    out = env.step(np.array([0.1, 1.0], np.float32))
    assert len(out) == 5
    obs, reward, terminated, truncated, infos = out
    assert isinstance(reward, float) and isinstance(terminated, bool) and isinstance(truncated, bool)
    assert isinstance(infos["scenario_status"], ScenarioStatus) and isinstance(infos["traffic_status"], TrafficStatus)
    assert infos["state"]["frame"] == 100 and abs(infos["state"]["speed"] - 0.1) < 1e-6   # a = 1 m/s^2 for 0.1 s
    with pytest.raises(InvalidAction):
        env.step(np.array([0.0, 2.5], np.float32))
    env.close()


def test_scenario_manager_update_then_check_status_equals_step():
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.traffic import BatchedScenarioManager
    sc = S.intersection(24, 32, seed=9)
    rng = np.random.default_rng(2)
    res = []
    for split in (False, True):
        m = BatchedScenarioManager(sc.n_env, sc.A, max_step=100, step_size=100)
        m.configure(sc.rows, check_dynamic=True, check_off_lane=True)
        m._static, m._boundary, m._lanes = sc.static, sc.boundary, sc.lanes
        m._push_geometry(); m.pool.set_lane_geometry(sc.lanes)
        m.reset(sc.x, sc.y, sc.heading, sc.speed, sc.type_id, sc.active)
        r = np.random.default_rng(5)
        for _ in range(5):
            a0, a1 = sc.sample_actions(r)
            if split:
                m.update(a0, a1)
                scen, traf = m.check_status()
            else:
                m.step(a0, a1)
        res.append([m.pool.download(f) for f in (L.F_X, L.F_HEADING, L.F_FLAGS, L.F_STATUS, L.F_REWARD, L.F_CNT_STEP)])
        assert len(m.get_active_participants()) == sc.n_env and m.get_observation().shape == (sc.n_env, 6)
        assert m.status_checklist["out_bound"].update().shape == (sc.n_env,)
        assert m.status_checklist["dynamic_collision"].update(ego_only=False).shape == (sc.n_env, sc.A)
        m.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    assert (res[0][5] == 5).all()


def test_parking_env_reaches_completed_when_parked_on_the_target():
    """Arrival: an ego standing on the target bay -> COMPLETED, terminated, reward +5 (arrival.py, parking.py:387-390)."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.envs import VecParkingEnv
    from tactics2d_amd.traffic import ScenarioStatus
    env = VecParkingEnv(64, seed=5)
    env.reset()
    sc = env._scene
    tc = sc.target.mean(1)
    pool = env.scenario_manager.pool
    pool.reset(tc[:, 0], tc[:, 1], sc.target_heading, np.zeros(64, np.float32), sc.type_id, sc.active)
    obs, reward, terminated, truncated, infos = env.step(np.zeros((64, 2), np.float32))
    assert terminated.all() and not truncated.any() and (reward == 5).all()
    assert (infos["scenario_status"] == ScenarioStatus.COMPLETED).all() and (infos["iou"] > 0.999).all()
    env.close()


def test_cfg1_single_parking_env_600_random_actions(oracle):
    """BASELINE.json configs[0] / the reference's own env test (tests/test_env.py:49-52: 600 samples of the action
    space through one ParkingEnv).  EVERY step, everything the 5-tuple carries is held against the oracle chain
    integrate -> collide -> status / reward / IoU -> 360-beam scan, teacher-forced on the env's own fp32 state:
    state within 1e-5 (default fast integrator), event flags, status bytes, terminated / truncated, NoAction counter
    and lidar bit for bit, reward and IoU to the last fp32 bit or two.  The env is reset whenever the episode ends."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.envs import ParkingEnv
    from tactics2d_amd.traffic import ScenarioStatus, TrafficStatus
    env = ParkingEnv(max_step=150, seed=0)
    obs, infos = env.reset()
    sc = env._vec._scene
    pool = env.scenario_manager.pool
    cfg = oracle.make_config(**{**sc.status, "max_step": 150})
    start_xy = np.stack([sc.x[:1], sc.y[:1]], 1)
    ep = oracle.EpisodeState(1, sc.target, None, start_xy)
    cnt = np.zeros(1, np.int32); frame = np.zeros(1, np.int32)
    rng = np.random.default_rng(0)
    n_done = 0
    seen = set()
    n_hits = 0
    prev = np.array([obs[0], obs[1], obs[2], obs[3]], np.float32)
    w0 = oracle.lidar(sc.rows, 1, 1, 0, prev[0:1], prev[1:2], prev[2:3], sc.type_id[:1], sc.active[:1], sc.static, 0, 360, 20.0)
    assert np.array_equal(np.float32(infos["lidar"]).view(np.uint32), w0[0].view(np.uint32))
    for t in range(600):
        # episodes cycle through: random actions (runs into the time limit), no action at all (NoAction after 100
        # still checks), full lock + full throttle (drives into the parked cars after 80 steps), random again
        mode = n_done % 3
        act = env.action_space.sample(rng)
        if mode == 1:
            act = np.float32([0.0, 0.0])
        elif mode == 2:
            act = np.float32([0.524, 2.0])
        obs, reward, terminated, truncated, infos = env.step(act)
        o = oracle.integrate(sc.rows, prev[0:1], prev[1:2], prev[2:3], prev[3:4], None, None,
                             np.float32([act[1]]), np.float32([act[0]]), sc.type_id[:1], sc.active[:1], 100)
        assert np.abs(np.asarray(obs[:4], np.float64) - o[0, :4]).max() <= 1e-5, (t, obs[:4], o[0, :4])
        assert isinstance(reward, float) and np.isfinite(reward)
        # events, status, reward, IoU and scan of the pose the env reports (fp32), by the oracle
        x, y, h = (np.float32([obs[k]]) for k in range(3))
        wf, _ = oracle.collide(sc.rows, 1, 1, x, y, h, sc.type_id[:1], sc.active[:1], sc.static, sc.boundary,
                               sc.boundary_valid, None, 0)
        assert np.array_equal(pool.download(L.F_FLAGS), wf), (t, wf)
        wst, wrw, wiou = oracle.status_ex(cfg, 1, wf, 100, cnt, frame, sc.rows, x, y, h, sc.type_id[:1], ep)
        assert (int(infos["scenario_status"]), int(infos["traffic_status"])) == (int(wst[0, 0]), int(wst[0, 1])), (t, wst)
        assert isinstance(infos["scenario_status"], ScenarioStatus) and isinstance(infos["traffic_status"], TrafficStatus)
        assert (terminated, truncated) == (bool(wst[0, 2]), bool(wst[0, 3])), (t, wst)
        assert abs(reward - float(wrw[0])) <= 2e-6, (t, reward, wrw)
        giou = np.float32(infos["iou"])
        assert np.isnan(giou) == np.isnan(wiou[0]) and (np.isnan(giou) or abs(float(giou) - float(wiou[0])) <= 1e-7), (t, giou, wiou)
        assert infos["state"]["frame"] == frame[0] and frame[0] % 100 == 0
        assert pool.download(L.F_CNT_NO_ACTION)[0] == ep.cnt_na[0] and pool.download(L.F_CNT_STEP)[0] == cnt[0]
        wl = oracle.lidar(sc.rows, 1, 1, 0, x, y, h, sc.type_id[:1], sc.active[:1], sc.static, 0, 360, 20.0)
        gl = np.float32(infos["lidar"])
        assert gl.shape == (360,) and np.array_equal(gl.view(np.uint32), wl[0].view(np.uint32)), (t, int((gl != wl[0]).sum()))
        n_hits += int(np.isfinite(gl).sum())
        seen.add((int(wst[0, 0]), int(wst[0, 1])))
        if terminated or truncated:
            n_done += 1
            obs, infos = env.reset()
            cnt[:] = 0; frame[:] = 0
            ep.reset_envs(np.ones(1, bool), start_xy)
        prev = np.array([obs[0], obs[1], obs[2], obs[3]], np.float32)
    env.close()
    print(f"cfg1: {n_done} episodes, statuses seen {sorted(seen)}, {n_hits} lidar returns")
    assert n_done >= 4 and {(1, 1), (3, 1), (1, 5), (6, 3)} <= seen and n_hits > 20000, (n_done, seen)


def test_vec_parking_env_device_resident_step_equals_the_host_step():
    """step_torch: actions in, observation tensors out, all on the device (zero-copy views + the lidar tensor);
    same numbers as the numpy step on the same actions."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd.envs import VecParkingEnv
    n = 300
    a = VecParkingEnv(n, max_step=30, seed=5, auto_reset=True); b = VecParkingEnv(n, max_step=30, seed=5, auto_reset=True)
    a.reset(); b.reset()
    rng = np.random.default_rng(2)
    dev = torch.device("cuda", 0)
    for t in range(40):
        act = a.action_space.sample(rng, n)
        obs, reward, term, trunc, infos = a.step(act)
        out = b.step_torch(torch.from_numpy(act).to(dev))
        torch.cuda.synchronize()
        assert out["lidar"].is_cuda and out["x"].is_cuda and out["lidar"].shape == (n, 360)
        for k, col in (("x", 0), ("y", 1), ("heading", 2), ("speed", 3), ("vx", 4), ("vy", 5)):
            assert np.array_equal(out[k].cpu().numpy(), obs[:, col]), (t, k)
        assert np.array_equal(out["reward"].cpu().numpy(), reward)
        st = out["status"].cpu().numpy()
        assert np.array_equal(st[:, 2].astype(bool), term) and np.array_equal(st[:, 3].astype(bool), trunc)
        assert np.array_equal(out["lidar"].cpu().numpy().view(np.uint32), infos["lidar"].view(np.uint32))
        assert np.array_equal(out["iou"].cpu().numpy(), infos["iou"], equal_nan=True)
    a.close(); b.close()


@pytest.mark.parametrize("source,zero_copy", [("layout", False), ("layout", True), ("generator", False), ("generator", True)])
def test_host_frame_equals_the_per_field_copies(source, zero_copy):
    """t2d_step_host's ONE packed frame against the per-field path it replaces (t2d_download of every column, a lidar scan
    into the pool's own buffer, t2d_get_parking_scenes): bit-identical, with copy commands and with the kernels
    reading / writing mapped host memory; the relative pose of _get_relative_pose (envs/parking.py:190-201) against numpy
    on the same fp32 state."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.envs import VecParkingEnv
    n = 257
    env = VecParkingEnv(n, max_step=25, seed=3, auto_reset=True, scene_source=source, zero_copy=zero_copy)
    obs, infos = env.reset()
    pool = env.scenario_manager.pool
    rng = np.random.default_rng(4)
    n_done = 0
    for t in range(60):
        if t:
            act = env.action_space.sample(rng, n)
            if t % 7 == 0:
                act[::3] = 0.0
            obs, reward, term, trunc, infos = env.step(act)
            assert np.array_equal(pool.download(L.F_ACT0)[:0], np.zeros(0, np.float32))   # (a download in between is harmless)
            st = pool.download(L.F_STATUS)
            assert np.array_equal(reward.view(np.uint32), pool.download(L.F_REWARD).view(np.uint32))
            assert np.array_equal(term, st[:, 2].astype(bool)) and np.array_equal(trunc, st[:, 3].astype(bool))
            assert np.array_equal(infos["scenario_status"], st[:, 0]) and np.array_equal(infos["traffic_status"], st[:, 1])
            n_done += int((term | trunc).sum())
        for k, f in enumerate((L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY)):
            assert np.array_equal(obs[:, k].view(np.uint32), pool.download(f).view(np.uint32)), (t, k)
        assert np.array_equal(infos["state"]["frame"], pool.download(L.F_FRAME_MS))
        assert np.array_equal(infos["iou"].view(np.uint32), pool.download(L.F_IOU).view(np.uint32))
        pool.lidar_scan()
        assert np.array_equal(infos["lidar"].view(np.uint32), pool.download(L.F_LIDAR).view(np.uint32)), t
        if source == "generator":
            g = pool.get_parking_scenes()
            assert np.array_equal(infos["target_area"], g.target) and np.array_equal(infos["target_heading"], g.target_heading)
            assert np.array_equal(infos["episode"], g.episode)
            tc = _area_centroid(g.target)
            th = g.target_heading
        else:
            tc = _area_centroid(env._scene.target)
            th = np.float64(env._scene.target_heading)
            assert np.array_equal(infos["target_area"], env._scene.target)
        x, y, h = (obs[:, k].astype(np.float64) for k in range(3))
        assert np.allclose(infos["diff_position"], np.hypot(tc[:, 0] - x, tc[:, 1] - y), rtol=1e-12, atol=1e-12)
        assert np.allclose(infos["diff_angle"], np.arctan2(tc[:, 1] - y, tc[:, 0] - x) - h, rtol=1e-12, atol=1e-12)
        assert np.array_equal(infos["diff_heading"], th - h)
    assert n_done > n   # episodes ended and restarted under the frame
    env.close()


def _area_centroid(quads):
    """shapely Polygon.centroid of a batch of quads (n, 4, 2), fp64 -- what t2d_set_target_areas computes."""
    q = np.asarray(quads, np.float64)
    nxt = np.roll(q, -1, 1)
    w = q[:, :, 0] * nxt[:, :, 1] - nxt[:, :, 0] * q[:, :, 1]
    a = w.sum(1) / 2
    return np.stack([((q[:, :, 0] + nxt[:, :, 0]) * w).sum(1), ((q[:, :, 1] + nxt[:, :, 1]) * w).sum(1)], 1) / (6 * a[:, None])


def test_host_frame_views_without_copy_and_without_lidar():
    """copy=False hands out views of the two alternating pinned frames (valid until the step after next); info_lidar=False
    leaves the scan out of the host path."""
    from tactics2d_amd.envs import VecParkingEnv
    n = 64
    a = VecParkingEnv(n, max_step=30, seed=5, auto_reset=True)
    b = VecParkingEnv(n, max_step=30, seed=5, auto_reset=True, copy=False, info_lidar=False)
    a.reset(); obs_b0, infos_b = b.reset()
    assert infos_b["lidar"] is None
    rng = np.random.default_rng(2)
    prev = None
    for t in range(12):
        act = a.action_space.sample(rng, n)
        oa, ra, ta, ua, ia = a.step(act)
        ob, rb, tb, ub, ib = b.step(act)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb) and np.array_equal(ta, tb) and np.array_equal(ua, ub)
        assert np.array_equal(ia["iou"], ib["iou"], equal_nan=True) and ib["lidar"] is None
        if prev is not None:   # the previous step's views are still intact (two frames alternate)
            assert np.array_equal(prev[0], prev[1])
        prev = (ob, oa.copy())
    a.close(); b.close()


def test_default_step_results_are_never_overwritten():
    """copy=True (default): what step() returns behaves like fresh arrays although no memcpy happens while pinned frames are
    free -- a caller that keeps EVERY step's results by reference still finds them intact later."""
    from tactics2d_amd.envs import VecParkingEnv
    n = 33
    env = VecParkingEnv(n, max_step=30, seed=7, auto_reset=True)
    env.reset()
    rng = np.random.default_rng(1)
    kept, saved = [], []
    for t in range(12):
        out = env.step(env.action_space.sample(rng, n))
        kept.append(out)
        saved.append((out[0].copy(), out[1].copy(), out[4]["lidar"].copy(), out[4]["state"]["x"].copy()))
    for (obs, rew, term, trunc, infos), (o, r, l, x) in zip(kept, saved):
        assert np.array_equal(obs, o) and np.array_equal(rew, r) and np.array_equal(infos["lidar"], l)
        assert np.array_equal(infos["state"]["x"], x)
    # ... and once the caller lets go, the pinned frames are handed out again (no copies: the arrays are views of them)
    del kept, out, obs, rew, term, trunc, infos
    pool = env.scenario_manager.pool
    bases = set()
    for t in range(6):
        o = env.step(env.action_space.sample(rng, n))[0]
        bases.add(o.base.__array_interface__["data"][0])
        del o
    assert len(bases) <= 2 and bases <= {fr.base.__array_interface__["data"][0] for fr in pool._frames if fr is not None}
    env.close()


def test_lidar_beams_is_a_regular_subset_of_the_reference_scan_and_copy_always_owns_its_memory():
    """VecParkingEnv(lidar_beams=120): the scan in info["lidar"] (host path) and in step_torch() equals every third beam of the
    360-beam scan bit for bit -- what docs/tutorial/train_parking_demo.ipynb keeps of envs/parking.py:422-431's observation --;
    copy="always": the arrays handed out own their memory (nothing of the pinned frames is aliased)."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd.envs import VecParkingEnv
    n = 96
    full = VecParkingEnv(n, max_step=40, seed=11, auto_reset=True)
    third = VecParkingEnv(n, max_step=40, seed=11, auto_reset=True, lidar_beams=120, copy="always")
    tenth = VecParkingEnv(n, max_step=40, seed=11, auto_reset=True, lidar_beams=36)
    _, i0 = full.reset(); _, i1 = third.reset(); _, i2 = tenth.reset()
    assert i1["lidar"].shape == (n, 120) and i2["lidar"].shape == (n, 36)
    assert np.array_equal(i0["lidar"][:, ::3].view(np.uint32), i1["lidar"].view(np.uint32))
    assert np.array_equal(i0["lidar"][:, ::10].view(np.uint32), i2["lidar"].view(np.uint32))
    rng = np.random.default_rng(4)
    pool = third.scenario_manager.pool
    frames = lambda: {fr.base.__array_interface__["data"][0] for fr in pool._frames if fr is not None}
    kept = []
    hits = 0
    for t in range(10):
        act = full.action_space.sample(rng, n)
        o0, r0, _, _, f0 = full.step(act)
        o1, r1, _, _, f1 = third.step(act)
        _, _, _, _, f2 = tenth.step(act)
        assert np.array_equal(o0, o1) and np.array_equal(r0, r1)
        assert np.array_equal(f0["lidar"][:, ::3].view(np.uint32), f1["lidar"].view(np.uint32)), t
        assert np.array_equal(f0["lidar"][:, ::10].view(np.uint32), f2["lidar"].view(np.uint32)), t
        hits += int(np.isfinite(f1["lidar"]).sum())
        base = o1.base if o1.base is not None else o1
        while getattr(base, "base", None) is not None:
            base = base.base
        assert base.__array_interface__["data"][0] not in frames()          # a real copy: not one of the pinned frames
        kept.append((o1, f1["lidar"], o1.copy(), f1["lidar"].copy()))
    assert hits > 1000
    for o, l, oc, lc in kept:                                                   # ... and never touched again
        assert np.array_equal(o, oc) and np.array_equal(l, lc, equal_nan=True)
    dev = torch.device("cuda", 0)
    a = torch.from_numpy(full.action_space.sample(rng, n)).to(dev)
    t0, t1 = full.step_torch(a), third.step_torch(a)
    torch.cuda.synchronize()
    assert t1["lidar"].shape == (n, 120)
    assert np.array_equal(t0["lidar"].cpu().numpy()[:, ::3].view(np.uint32), t1["lidar"].cpu().numpy().view(np.uint32))
    with pytest.raises(ValueError):
        VecParkingEnv(4, lidar_beams=100)
    with pytest.raises(ValueError):
        VecParkingEnv(4, copy="sometimes")
    full.close(); third.close(); tenth.close()


def test_a_new_frame_configuration_never_leaves_the_pool_reading_freed_action_buffers():
    """t2d_step_host binds the pool's actions to its staging buffers; t2d_frame_config with ANOTHER configuration frees them.
    The pool falls back to its own action fields (round-5 advice: it kept reading the freed device / mapped memory)."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    for zero_copy in (False, True):
        sc = S.parking(32, seed0=3)
        pool = ParticipantPool(sc.n_env, 1)
        sc.load(pool)
        pool.lidar_config(360, 20.0)
        a0, a1 = sc.sample_actions(np.random.default_rng(1))
        pool.set_actions(a0, a1)                                 # the pool's own action fields hold THESE
        pool.snapshot()
        pool.step(100)
        want = [pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED)]
        pool.restore(False)
        pool.frame_config(lidar=False, zero_copy=zero_copy)
        other = np.ascontiguousarray(np.stack([-a1, -a0], 1), np.float32)   # (steering, accel): different actions
        pool.step_host(other, 100)
        pool.restore(False)
        pool.frame_config(lidar=True, zero_copy=zero_copy, n_frames=3)       # another configuration: staging buffers freed
        pool.step(100)                                            # must read the pool's own fields, not freed memory
        got = [pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED)]
        assert all(np.array_equal(g, w) for g, w in zip(got, want)), zero_copy
        # ... and a rejected action row is not staged: a plain step afterwards uses the last ACCEPTED actions
        pool.restore(False)
        ok = np.ascontiguousarray(np.stack([a1, a0], 1), np.float32)
        box = np.float32([-0.524, 0.524, -2.0, 2.0])
        pool.step_host(ok, 100, action_box=box)
        first = [pool.download(f) for f in (L.F_X, L.F_Y)]
        pool.restore(False)
        bad = ok.copy(); bad[5, 0] = np.nan
        with pytest.raises(Exception):
            pool.step_host(bad, 100, action_box=box)
        pool.step_host(None, 100)                                 # the actions already in place: the accepted ones
        assert all(np.array_equal(g, w) for g, w in zip([pool.download(f) for f in (L.F_X, L.F_Y)], first)), zero_copy
        pool.close()


def test_nan_actions_are_invalid_actions_and_step_nothing():
    """`action_space.contains(nan)` is False in the reference (envs/parking.py:235-236): the host path raises InvalidAction
    from the staging pass of t2d_step_host and nothing is stepped -- vector env and single env alike."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.envs import InvalidAction, ParkingEnv, VecParkingEnv
    env = VecParkingEnv(16, max_step=30, seed=2)
    env.reset()
    pool = env.scenario_manager.pool
    a = env.action_space.sample(np.random.default_rng(0), 16)
    env.step(a)
    x0, cnt0 = pool.download(L.F_X), pool.download(L.F_CNT_STEP)
    for bad in (np.nan, np.inf, -np.inf, 0.6):
        b = a.copy(); b[7, 0] = bad
        with pytest.raises(InvalidAction):
            env.step(b)
    assert np.array_equal(pool.download(L.F_X), x0) and np.array_equal(pool.download(L.F_CNT_STEP), cnt0)
    env.step(a)
    assert (pool.download(L.F_CNT_STEP) == cnt0 + 1).all()
    env.close()
    one = ParkingEnv(seed=1)
    one.reset()
    for bad in ([np.nan, 0.0], [0.0, np.nan], [0.0, 2.5], [0.0], "x"):
        with pytest.raises(InvalidAction):
            one.step(bad)
    obs, reward, term, trunc, info = one.step(np.float32([0.1, 0.5]))
    assert np.isfinite(obs).all() and info["state"]["frame"] == 100
    one.close()


def test_scenario_manager_step_host_equals_step_plus_downloads():
    """BatchedScenarioManager.step_host on a multi-agent pool (mixed scene, ego = agent 0): the frame of ONE call against
    set_actions + step + one download per field."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.traffic import BatchedScenarioManager
    sc = S.mixed(48, 64, seed=9)
    rng = np.random.default_rng(3)

    def make():
        m = BatchedScenarioManager(sc.n_env, sc.A, max_step=20, step_size=sc.interval_ms)
        sc.load(m.pool)
        m.pool.set_status_config(**{**sc.status, "max_step": 20})
        m.pool.set_auto_reset(True)
        return m

    a, b = make(), make()
    for t in range(25):
        a0, a1 = sc.sample_actions(rng)
        acts = np.stack([a1, a0], 1)                     # (steering, accel) per participant
        if t >= 5 and t % 2:                             # ... written straight into the pool's pinned staging buffer: no copy
            buf = a.pool.host_action_buffer()
            buf[:] = acts
            acts = buf
        fr = a.step_host(acts)
        b.step(a0, a1)
        obs = b.get_observation()
        assert np.array_equal(fr.obs[:, :4], obs[:, :4]), t
        st = b.pool.download(L.F_STATUS)
        assert np.array_equal(fr.status, st) and np.array_equal(fr.reward, b.pool.download(L.F_REWARD))
        assert np.array_equal(fr.cnt_step, b.pool.download(L.F_CNT_STEP)) and np.array_equal(fr.frame_ms, b.pool.download(L.F_FRAME_MS))
        assert np.array_equal(a.flags(), b.flags())
    assert (st[:, 2] | st[:, 3]).any() or t > 20
    a.close(); b.close()


def test_single_env_results_survive_later_steps_and_resets():
    """What ParkingEnv.reset / step hand out never aliases the pinned frame later calls fill: a reset's observation and lidar
    are intact after the next steps, a step's after the following ones and after a reset (which keeps the frames: the same
    frame configuration is asked for again)."""
    from tactics2d_amd.envs import ParkingEnv
    env = ParkingEnv(seed=4, max_step=30)
    obs0, info0 = env.reset()
    keep0 = (obs0.copy(), info0["lidar"].copy())
    rng = np.random.default_rng(0)
    kept = []
    for t in range(8):
        o, r, te, tr, info = env.step(env.action_space.sample(rng))
        kept.append((o, o.copy(), info["lidar"], info["lidar"].copy()))
    assert np.array_equal(obs0, keep0[0]) and np.array_equal(info0["lidar"], keep0[1])
    obs1, info1 = env.reset()
    for _ in range(3):
        env.step(env.action_space.sample(rng))
    for o, oc, l, lc in kept:
        assert np.array_equal(o, oc) and np.array_equal(l, lc)
    assert np.array_equal(obs1, obs0)          # the same scene, the same start
    env.close()
