"""ParkingLotGenerator (scope row f4).  Parity with the reference is UNPINNED (numpy global stream + shapely
predicates cannot run here), so the CPU tests check the restatement against the properties the reference's
algorithm guarantees (generate_parking_lot.py:207-237, 239-444) and against its own exact-rational predicates;
the GPU tests pin the kernel to the restatement bit for bit."""
from fractions import Fraction

import numpy as np
import pytest

N = 6000


@pytest.fixture(scope="module")
def scenes(oracle):
    return oracle.generate_parking(20260927, N, 0.5, (4.284, 1.81))


def _exact_intersects(a, b):
    """Closed convex-quad intersection in exact rational arithmetic (separating edge with strict separation)."""
    def ccw(q):
        q = [(Fraction(float(x)), Fraction(float(y))) for x, y in q]
        area = sum(q[i][0] * q[(i + 1) % 4][1] - q[(i + 1) % 4][0] * q[i][1] for i in range(4))
        return q if area >= 0 else q[::-1]
    A, B = ccw(a), ccw(b)
    for P, Q in ((A, B), (B, A)):
        for i in range(4):
            p, q = P[i], P[(i + 1) % 4]
            if all((q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0]) < 0 for r in Q):
                return False
    return True


def _box(x, y, h, L, W):
    c, s = np.cos(h), np.sin(h)
    base = np.array([[L / 2, -W / 2], [L / 2, W / 2], [-L / 2, W / 2], [-L / 2, -W / 2]])
    return base @ np.array([[c, s], [-s, c]]) + [x, y]


def test_flags_and_mode_statistics(oracle, scenes):
    info = scenes["info"]
    assert not (info & (oracle.GEN_UNVERIFIED | oracle.GEN_START_UNVERIFIED | oracle.GEN_NONCONVEX | oracle.GEN_OVERFLOW)).any()
    bay = (info & oracle.GEN_BAY) != 0
    assert abs(bay.mean() - 0.5) < 0.03                       # type_proportion :256
    flipped = (info & oracle.GEN_START_FLIPPED) != 0
    assert abs(flipped.mean() - 0.5) < 0.03                   # :412
    tflip = (info & oracle.GEN_TARGET_FLIPPED) != 0
    assert (tflip == (flipped & ~bay)).all()                  # only parallel targets flip :421-431
    assert ((info >> 8) & 255).min() >= 1 and ((info >> 16) & 255).min() >= 1
    # proportions 0 / 1 give one mode only; out-of-range values are clipped (:57)
    assert not (oracle.generate_parking(1, 500, 0.0)["info"] & oracle.GEN_BAY).any()
    assert (oracle.generate_parking(1, 500, 1.0)["info"] & oracle.GEN_BAY).all()
    assert (oracle.generate_parking(1, 500, 7.0)["info"] & oracle.GEN_BAY).all()
    assert not (oracle.generate_parking(1, 500, -2.0)["info"] & oracle.GEN_BAY).any()


def test_heading_and_position_ranges(oracle, scenes):
    info = scenes["info"]
    bay = (info & oracle.GEN_BAY) != 0
    flipped = (info & oracle.GEN_START_FLIPPED) != 0
    th = scenes["target_heading"] - np.where((info & oracle.GEN_TARGET_FLIPPED) != 0, np.pi, 0.0)
    assert (th[bay] >= np.pi * 4 / 9).all() and (th[bay] <= np.pi * 5 / 9).all()          # :33-36
    assert (th[~bay] >= -np.pi / 18).all() and (th[~bay] <= np.pi / 18).all()
    sh = scenes["start"][:, 2] - np.where(flipped, np.pi, 0.0)
    assert (np.abs(sh) <= np.pi / 18 + 1e-12).all()                                       # :227
    assert (np.abs(scenes["start"][:, 0]) <= 7.5 + 1e-9).all()                            # :399
    # clipped Gaussians actually hit both clip ends and the interior
    assert (th[bay] == np.pi * 4 / 9).any() and (th[bay] == np.pi * 5 / 9).any()
    assert 0.2 < (np.abs(th[bay] - np.pi / 2) < np.pi / 54).mean() < 0.9
    # target centre on x = 0, 0.8 .. 1.6 m above the back wall line (:108-113)
    t = scenes["target"].astype(np.float64)
    nf = (info & oracle.GEN_TARGET_FLIPPED) == 0
    assert np.abs(t[nf].mean(axis=1)[:, 0]).max() < 1e-6
    low = t.min(axis=1)[:, 1]
    assert (low >= 0.8 - 1e-5).all() and (low <= 1.6 + 1e-5).all()


def test_boundary_is_floor_ceil_of_start_and_target(oracle, scenes):
    t = scenes["target"].astype(np.float64)
    # the reference's centre = mean of the ring (:410-412); the fp32 ring here is within 1e-6 of the fp64 one
    tx, ty = t.mean(axis=1).T
    sx, sy = scenes["start"][:, 0], scenes["start"][:, 1]
    b = scenes["boundary"].astype(np.float64)
    near = lambda v: np.abs(v - np.round(v)) < 1e-5    # a value that close to an integer may floor either way
    for col, want, val in ((0, np.floor(np.minimum(sx, tx) - 13), np.minimum(sx, tx)), (1, np.ceil(np.maximum(sx, tx) + 13), np.maximum(sx, tx)),
                           (2, np.floor(np.minimum(sy, ty) - 13), np.minimum(sy, ty)), (3, np.ceil(np.maximum(sy, ty) + 13), np.maximum(sy, ty))):
        ok = (b[:, col] == want) | near(val)
        assert ok.all(), col
    assert (b[:, 1] - b[:, 0] >= 26).all() and (b[:, 3] - b[:, 2] >= 26).all()


def test_obstacle_ids_follow_map_add_area(oracle, scenes):
    ids, n = scenes["quad_id"], scenes["n_quads"]
    for e in range(0, N, 7):
        row = ids[e, :n[e]]
        assert (row >= 0).all() and (ids[e, n[e]:] == -1).all()
        assert len(set(row.tolist())) == n[e]                  # dict keys: one area per id (map.py:444-453)
    # wall on the left (id 1 with vertices on x = -15) => no further vehicle "0003" on that side (:272-285): an
    # id-3 area can then only be the far wall (:347-356), which spans the whole scene width.  (Ids 5 / 7 may still
    # appear: the perturbed vehicles are numbered from len(obstacles) + 1, :371, and Map.add_area lets such an
    # id replace a parked vehicle with the same id -- kept as the reference does it.)
    q = scenes["quads"]
    n_wall = 0
    for e in range(N):
        row = ids[e, :n[e]].tolist()
        if 1 in row and (q[e, row.index(1), :, 0] == -15.0).any():
            n_wall += 1
            if 3 in row:
                xs = q[e, row.index(3), :, 0]
                assert xs.min() == -15.0 and xs.max() == 15.0
    assert 0.1 * N < n_wall < 0.3 * N                          # p = 0.2, minus the 5 % drop
    assert 3 <= n.min() or (n < 3).mean() < 0.01
    assert n.max() <= oracle.GEN_MAX_QUADS


def test_accepted_scenes_satisfy_the_reference_predicates(oracle, scenes):
    """_verify_obstacles / _verify_start_state: the target touches none of back wall / neighbours, the start
    footprint touches no obstacle and not the target -- re-checked in exact rational arithmetic on the fp32
    outputs (a 1e-6 shift cannot create contact: the reference's own margins are >= 0.8 m)."""
    L, W = 4.284, 1.81
    info = scenes["info"]
    flipped = (info & oracle.GEN_START_FLIPPED) != 0
    for e in range(0, N, 9):
        n = scenes["n_quads"][e]
        ids = scenes["quad_id"][e, :n].tolist()
        tq = scenes["target"][e]
        for k in range(n):
            if ids[k] in (0, 1, 2):
                assert not _exact_intersects(tq, scenes["quads"][e, k]), (e, ids[k])
        sx, sy, sh = scenes["start"][e]
        if flipped[e]:   # the flip mirrors (x, y) through the box centre = itself, and turns the heading by pi
            sh -= np.pi
        body = _box(sx, sy, sh, L, W)
        for k in range(n):
            assert not _exact_intersects(body, scenes["quads"][e, k]), (e, ids[k])
        assert not _exact_intersects(body, tq)


def test_streams_are_per_scene_and_reproducible(oracle):
    a = oracle.generate_parking(5, 300, 0.5)
    b = oracle.generate_parking(5, 100, 0.5, first_env=200)
    for k in a:
        assert np.array_equal(a[k][200:], b[k]), k          # scene e depends on (seed, first_env + e) only
    c = oracle.generate_parking(6, 300, 0.5)
    assert not np.array_equal(a["start"], c["start"])
    assert len({tuple(r) for r in a["start"].round(9).tolist()}) == 300   # scenes differ from one another
    # thread count does not change the result
    oracle.set_threads(4)
    try:
        d = oracle.generate_parking(5, 300, 0.5)
    finally:
        oracle.set_threads(1)
    for k in a:
        assert np.array_equal(a[k], d[k]), k


def test_libm_and_deterministic_trig_agree(oracle):
    """trig=0 uses libm (what numpy calls), trig=1 the deterministic spec the GPU shares: same scenes up to
    1-ulp effects (a rejection decision could flip in principle; none does on this sample)."""
    a = oracle.generate_parking(11, 2000, 0.5, trig=0)
    b = oracle.generate_parking(11, 2000, 0.5, trig=1)
    same = (a["info"] == b["info"]) & (a["n_quads"] == b["n_quads"])
    assert same.mean() > 0.999
    assert np.abs(a["start"][same] - b["start"][same]).max() < 1e-9
    assert np.abs(a["quads"][same] - b["quads"][same]).max() < 1e-5


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("seed,n_env,prop,size", [(1, 4096, 0.5, (4.284, 1.81)), (2, 1000, 1.0, (5.3, 2.5)),
                                                   (3, 1000, 0.0, (5.3, 2.5)), (2**63 + 5, 257, 0.3, (4.5, 2.0)),
                                                   (4, 1, 0.5, (5.3, 2.5))])
def test_kernel_matches_oracle_bit_for_bit(oracle, seed, n_env, prop, size):
    from tactics2d_amd.generator import ParkingLotGenerator
    got = ParkingLotGenerator(size, prop).generate(n_env, seed, first_env=17)
    want = oracle.generate_parking(seed, n_env, prop, size, first_env=17, trig=1)
    assert np.array_equal(got.info, want["info"])
    assert np.array_equal(got.n_quads, want["n_quads"]) and np.array_equal(got.quad_id, want["quad_id"])
    assert np.array_equal(got.start, want["start"])                     # fp64, bit for bit
    assert np.array_equal(got.target_heading, want["target_heading"])
    assert np.array_equal(got.quads.view(np.uint32), want["quads"].view(np.uint32))
    assert np.array_equal(got.target.view(np.uint32), want["target"].view(np.uint32))
    assert np.array_equal(got.boundary, want["boundary"])


@pytest.mark.gpu
def test_invalid_vehicle_size_falls_back_and_args_are_checked(oracle):
    from tactics2d_amd import _ffi
    from tactics2d_amd.generator import ParkingLotGenerator
    g = ParkingLotGenerator((2.0, 3.0), 0.5)                              # length < width :45-50
    assert g.vehicle_size == (5.3, 2.5)
    got = g.generate(64, 9)
    want = oracle.generate_parking(9, 64, 0.5, (5.3, 2.5))
    assert np.array_equal(got.start, want["start"])
    assert ParkingLotGenerator().generate(0, 1).n_env == 0
    with pytest.raises(_ffi.T2DError):
        _ffi.check(_ffi.lib().t2d_generate_parking(0, 1, 0, 4, 0.5, 5.3, 2.5, *([None] * 8)))


@pytest.mark.gpu
def test_generated_scenes_load_and_step(oracle):
    """The generated batch goes through the ordinary boundary (set_static_geometry / set_target_areas / reset) and
    the step kernel's events agree with the oracle on it; nobody starts in collision or out of bounds."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.generator import ParkingLotGenerator
    from tactics2d_amd.pool import ParticipantPool
    n_env = 2048
    scenes = ParkingLotGenerator((4.284, 1.81), 0.5).generate(n_env, 31)
    sc = scenes.scene()
    pool = ParticipantPool(n_env, 1)
    sc.load(pool)
    pool.collide()
    assert not pool.download(L.F_FLAGS).any()                              # _verify_start_state held
    rng = np.random.default_rng(0)
    cfg = oracle.make_config(**sc.status)
    hit = 0
    for step in range(40):
        a0, a1 = sc.sample_actions(rng)
        pool.set_actions(a0, a1)
        pool.step(100)
        gx, gy, gh = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        wf, we = oracle.collide(sc.rows, n_env, 1, gx, gy, gh, sc.type_id, sc.active, sc.static, sc.boundary, None, None, 1)
        assert (pool.download(L.F_FLAGS) == wf).all()
        hit += int((wf != 0).sum())
    pool.close()


@pytest.mark.gpu
def test_vec_parking_env_with_generated_scenes():
    """VecParkingEnv(scene_source="generator"): reset() = generate + install + agent reset (envs/parking.py:397-441);
    a second reset with another seed gives other scenes; episodes run and terminate through the normal path."""
    from tactics2d_amd.envs import ParkingEnv, VecParkingEnv
    env = VecParkingEnv(512, max_step=50, scene_source="generator", type_proportion=0.5, seed=3)
    obs, infos = env.reset()
    start = env.generated.start
    assert np.allclose(obs[:, 0], start[:, 0], atol=1e-5) and np.allclose(obs[:, 1], start[:, 1], atol=1e-5)
    assert np.array_equal(infos["target_heading"], np.float32(env.generated.target_heading))
    assert set(np.unique(env.generated.mode)) == {"bay", "parallel"}
    lidar = infos["lidar"]
    assert np.isfinite(lidar).any(axis=1).mean() > 0.9              # obstacles within 20 m of nearly every start
    rng = np.random.default_rng(0)
    done = np.zeros(512, bool)
    for _ in range(60):
        obs, reward, term, trunc, infos = env.step(env.action_space.sample(rng, 512))
        done |= term | trunc
    assert done.all()                                               # max_step = 50 truncates whoever survived
    obs2, _ = env.reset(seed=4)
    assert not np.allclose(obs2[:, :2], start[:, :2])
    env.close()
    one = ParkingEnv(type_proportion=1.0, scene_source="generator", seed=11)
    o, info = one.reset()
    assert one._vec.generated.mode[0] == "bay" and o.shape == (6,)
    one.close()


def _pool_for(sc, n_env):
    from tactics2d_amd.pool import ParticipantPool
    pool = ParticipantPool(n_env, 1)
    pool.set_param_table(sc.rows)
    pool.set_status_config(**sc.status)
    return pool


_CMP = None


def _fields():
    from tactics2d_amd import layout as L
    return [L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_IDS, L.F_FLAGS, L.F_ENV_FLAGS, L.F_CNT_STEP,
            L.F_FRAME_MS, L.F_STATUS, L.F_REWARD, L.F_IOU, L.F_CNT_NO_ACTION]


@pytest.mark.gpu
@pytest.mark.parametrize("n_env", [700, 4096])
def test_device_installed_scenes_equal_the_host_install(n_env):
    """t2d_parking_scenes (generate + install in one launch, capacity-layout geometry) against the same scenes taken
    through the host boundary (generate -> set_static_geometry / set_target_areas / reset / snapshot): every pool
    field equal after the install and after each of 40 steps, lidar scans included."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.generator import ParkingLotGenerator
    size = (4.284, 1.81)
    scenes = ParkingLotGenerator(size, 0.5).generate(n_env, 77)
    sc = scenes.scene(max_step=25)
    host = _pool_for(sc, n_env)
    sc.load(host)
    dev = _pool_for(sc, n_env)
    dev.parking_scenes(77, 0.5, size, regenerate=False)
    got = dev.get_parking_scenes()
    assert np.array_equal(got.start, scenes.start) and np.array_equal(got.quads, scenes.quads)
    assert np.array_equal(got.info, scenes.info) and not got.episode.any()
    for p in (host, dev):
        p.set_auto_reset(True)
        p.lidar_config(360, 20.0, include_participants=False)
    rng = np.random.default_rng(5)
    ever_done = False
    for step in range(41):
        ever_done |= bool(host.download(L.F_STATUS)[:, 2:].any())
        for f in _fields():
            assert np.array_equal(host.download(f), dev.download(f), equal_nan=True), (step, f)
        if step % 8 == 0:
            host.lidar_scan(); dev.lidar_scan()
            assert np.array_equal(host.download(L.F_LIDAR), dev.download(L.F_LIDAR)), step
        a0, a1 = sc.sample_actions(rng)
        for p in (host, dev):
            p.set_actions(a0, a1)
            p.step(100)
    assert ever_done                                       # episodes did end (and were auto-reset) on the way
    host.close(); dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [True, "inline"])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_scene_regeneration_follows_the_episode_streams(oracle, auto_reset, mode):
    """regenerate=True: an env whose episode ended gets the scene of stream first_env + e + k * stride right after the
    step; everything it installs is checked against the oracle's scene for that stream, the step's own flags against
    the oracle's events on the scene that was in place during the step."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.generator import ParkingLotGenerator
    n_env, seed, size, first, stride = 320, 12, (4.284, 1.81), 1000, 5000
    scenes = ParkingLotGenerator(size, 0.5).generate(n_env, seed, first_env=first)
    sc = scenes.scene(max_step=3 if mode is True else 12)
    pool = _pool_for(sc, n_env)
    pool.parking_scenes(seed, 0.5, size, regenerate=mode, first_env=first, env_stride=stride)
    if auto_reset:
        pool.set_auto_reset(True)
    cur = pool.get_parking_scenes()
    assert np.array_equal(cur.start, scenes.start)
    episode = np.zeros(n_env, np.int64)
    rng = np.random.default_rng(2)
    n_regen = 0
    for step in range(120 if mode is True else 45):     # staged: long enough to go round the record ring
        a0, a1 = sc.sample_actions(rng)
        pool.set_actions(a0, a1)
        before = cur
        pool.profile_enable(step == 0)
        pool.step(100)
        status = pool.download(L.F_STATUS)
        done = (status[:, 2] | status[:, 3]).astype(bool)
        cur = pool.get_parking_scenes()
        # flags of this step were computed on `before`
        flags = pool.download(L.F_FLAGS)
        live = ~done
        gx, gy, gh = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        if live.any() and not auto_reset:
            idx = np.nonzero(live)[0]
            sub = ParkingScenesSubset(before, idx)
            wf, _ = oracle.collide(sc.rows, len(idx), 1, gx[idx], gy[idx], gh[idx], sc.type_id[idx], sc.active[idx],
                                   sub.static_csr(), sub.boundary, None, None, 1)
            assert np.array_equal(flags[idx], wf)
        for e in np.nonzero(done)[0]:
            episode[e] += 1
            want = oracle.generate_parking(seed, 1, 0.5, size, first_env=first + e + episode[e] * stride, trig=1)
            assert cur.episode[e] == episode[e]
            assert np.array_equal(cur.start[e], want["start"][0]) and np.array_equal(cur.quads[e], want["quads"][0])
            assert np.array_equal(cur.target[e], want["target"][0]) and cur.info[e] == want["info"][0]
            assert np.array_equal(cur.boundary[e], want["boundary"][0])
            assert gx[e] == np.float32(want["start"][0, 0]) and gy[e] == np.float32(want["start"][0, 1])
            assert gh[e] == np.float32(want["start"][0, 2])
            n_regen += 1
        assert np.array_equal(cur.episode, episode)
        keep = ~done
        assert np.array_equal(cur.start[keep], before.start[keep])      # nobody else's scene moved
        assert (pool.download(L.F_CNT_STEP)[done] == 0).all()
    assert n_regen > n_env                                                # every env went through > 1 episode on average
    pool.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ego_kernel", [True, False])
def test_staged_scenes_survive_the_shortest_episodes_in_a_row(oracle, ego_kernel):
    """The staging ring holds 16 episodes per env and is topped up every 8 steps (t2d_api.hip regenerate_done_scenes).  The
    shortest episode the status rules allow in a generated lot is two steps (max_step = 1: cnt_step > max_step at the second
    step, parking.py:271; the no-action detector needs a previous pose as well, and the generator rules out a start in
    collision): EVERY env ends an episode at EVERY second step, for three times round the ring, with host synchronisation
    only now and then (the refill stream runs behind the steps).  Every commit (sixteen lanes per env, scene_commit_kernel)
    is checked field for field against the oracle's scene of that episode's stream, the lidar edges and their culling bytes
    through a scan against the oracle's on the same lot.  ego_kernel: the commit in the ego step kernel's epilogue, or -- the
    pool kept on the general step kernel -- the same device function in a launch of its own behind the step."""
    from tactics2d_amd import layout as L
    from tactics2d_amd.generator import ParkingLotGenerator
    n_env, seed, size, first, stride = 96, 5, (4.284, 1.81), 40, 1000
    scenes = ParkingLotGenerator(size, 0.5).generate(n_env, seed, first_env=first)
    sc = scenes.scene(max_step=1)
    pool = _pool_for(sc, n_env)
    pool.set_ego_kernel(ego_kernel)
    pool.parking_scenes(seed, 0.5, size, regenerate=True, first_env=first, env_stride=stride)
    pool.lidar_config(120, 20.0, False)
    parked = np.zeros(n_env, np.float32)
    for step in range(1, 101):
        pool.set_actions(parked, parked)
        pool.step(100)
        if step % 14 and step < 96:
            continue
        status = pool.download(L.F_STATUS)
        assert (status[:, 2] | status[:, 3]).astype(bool).all() == (step % 2 == 0)
        cur = pool.get_parking_scenes()
        assert (cur.episode == step // 2).all()
        for e in range(n_env):
            want = oracle.generate_parking(seed, 1, 0.5, size, first_env=first + e + (step // 2) * stride, trig=1)
            for key in ("start", "quads", "quad_id", "n_quads", "target", "target_heading", "boundary", "info"):
                assert np.array_equal(getattr(cur, key)[e], want[key][0]), (step, e, key)
        gx, gy, gh = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        assert np.array_equal(gx, cur.start[:, 0].astype(np.float32)) and np.array_equal(gy, cur.start[:, 1].astype(np.float32))
        if step % 2 == 0:                               # (a stepped heading is wrapped into [0, 2 pi), the start's is as generated)
            assert np.array_equal(gh, cur.start[:, 2].astype(np.float32))
        assert (pool.download(L.F_CNT_STEP) == step % 2).all() and (pool.download(L.F_SPEED) == 0).all()
        # the lidar edges (and their culling bytes) the commit wrote: the scan equals the oracle's brute force on the same lots
        pool.lidar_scan()
        got = pool.download(L.F_LIDAR)
        ref = oracle.lidar(sc.rows, n_env, 1, 0, gx, gy, gh, sc.type_id, sc.active, cur.static_csr(), 0, 120, 20.0, trig=0)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), step
    pool.close()


class ParkingScenesSubset:
    """Rows `idx` of a ParkingScenes as the CSR / boundary arguments of the oracle."""

    def __init__(self, scenes, idx):
        from tactics2d_amd.generator import ParkingScenes
        self._s = ParkingScenes(scenes.quads[idx], scenes.quad_id[idx], scenes.n_quads[idx], scenes.start[idx],
                                scenes.target[idx], scenes.target_heading[idx], scenes.boundary[idx], scenes.info[idx], None)
        self.boundary = np.ascontiguousarray(scenes.boundary[idx], np.float32)

    def static_csr(self):
        return self._s.static_csr()


@pytest.mark.gpu
def test_vec_env_auto_reset_moves_on_to_new_scenes():
    """auto_reset + generated scenes: a finished episode continues in the scene of the env's next episode (the
    reference generates a new lot at every reset), infos follow, and step_torch keeps everything on the device."""
    import torch
    from tactics2d_amd.envs import VecParkingEnv
    env = VecParkingEnv(256, max_step=10, scene_source="generator", auto_reset=True, seed=8)
    obs, infos = env.reset()
    first_targets = infos["target_area"].copy()
    start0 = env.generated.start.copy()
    rng = np.random.default_rng(1)
    for _ in range(12):
        obs, reward, term, trunc, infos = env.step(env.action_space.sample(rng, 256))
    assert env.generated.episode.min() >= 1                       # max_step = 10: everyone is in a later episode
    moved = np.abs(env.generated.start[:, :2] - start0[:, :2]).max(axis=1) > 1e-3
    assert moved.mean() > 0.95
    assert not np.array_equal(infos["target_area"], first_targets)
    assert np.array_equal(infos["target_area"], env.generated.target)
    act = torch.zeros((256, 2), dtype=torch.float32, device="cuda")
    out = env.step_torch(act)
    torch.cuda.synchronize()
    assert out["lidar"].shape == (256, 360) and torch.isfinite(out["lidar"]).any()
    env.close()


@pytest.mark.gpu
def test_parking_scenes_argument_and_state_checks():
    """Error behaviour at the boundary: wrong pool shape, call order, mode mixing -- and leaving the generated-scene
    mode through t2d_set_static_geometry gives an ordinary host-described pool again."""
    from tactics2d_amd import _ffi
    from tactics2d_amd.generator import ParkingLotGenerator
    from tactics2d_amd.pool import ParticipantPool
    size = (4.284, 1.81)
    sc = ParkingLotGenerator(size, 0.5).generate(64, 3).scene(max_step=30)
    p2 = ParticipantPool(8, 2)
    p2.set_param_table(sc.rows)
    with pytest.raises(_ffi.T2DError):
        p2.parking_scenes(1)                                   # one participant per env only
    p2.close()
    raw = ParticipantPool(64, 1)
    with pytest.raises(_ffi.T2DError):
        raw.parking_scenes(1)                                  # parameter table first
    with pytest.raises(_ffi.T2DError):
        raw.get_parking_scenes()                               # nothing generated yet
    raw.close()
    pool = _pool_for(sc, 64)
    with pytest.raises(_ffi.T2DError):
        _ffi.check(_ffi.lib().t2d_parking_scenes(pool._h, 1, 0, 64, 0.5, size[0], size[1], 7), pool._h)   # regenerate in 0..2
    with pytest.raises(_ffi.T2DError):
        _ffi.check(_ffi.lib().t2d_parking_scenes(pool._h, 1, 0, -1, 0.5, size[0], size[1], 0), pool._h)  # stride >= 0
    pool.parking_scenes(3, 0.5, size, regenerate=True)
    with pytest.raises(_ffi.T2DError):
        pool.set_target_areas(sc.target)                       # targets belong to the generated scenes
    with pytest.raises(_ffi.T2DError):
        pool.set_lane_geometry(sc.static)
    rng = np.random.default_rng(0)
    for _ in range(5):
        pool.set_actions(*sc.sample_actions(rng)); pool.step(100)
    # back to a host-described scene: same pool, ordinary path, results equal to a fresh pool's
    sc.load(pool)
    fresh = _pool_for(sc, 64); sc.load(fresh)
    for _ in range(35):
        a0, a1 = sc.sample_actions(rng)
        for q in (pool, fresh):
            q.set_actions(a0, a1); q.step(100)
    for f in _fields():
        assert np.array_equal(pool.download(f), fresh.download(f), equal_nan=True), f
    with pytest.raises(_ffi.T2DError):
        pool.get_parking_scenes()                              # left the generated-scene mode
    pool.close(); fresh.close()


def test_layout_relations_of_the_reference_construction(oracle, scenes):
    """Relations the reference's construction implies (generate_parking_lot.py:127-205, 338-346, 396-407): the left
    obstacle (id 1) lies left of the target bay and the right one (id 2) right of it with the 0.8 m clearance, the
    back wall (id 0) spans the scene below y = 0, and the start pose sits at least 1.8 m above every obstacle that
    was placed before the start range was fixed (walls and parked vehicles)."""
    q, ids, n = scenes["quads"].astype(np.float64), scenes["quad_id"], scenes["n_quads"]
    t = scenes["target"].astype(np.float64)
    checked = {0: 0, 1: 0, 2: 0}
    for e in range(0, N, 5):
        row = ids[e, :n[e]].tolist()
        tx0, tx1 = t[e, :, 0].min(), t[e, :, 0].max()
        if 0 in row:
            w = q[e, row.index(0)]
            assert w[:, 0].min() == -15.0 and w[:, 0].max() == 15.0 and w[:, 1].max() == 0.0 and -1.5 <= w[:, 1].min() <= -0.5
            checked[0] += 1
        if 1 in row:
            assert q[e, row.index(1), :, 0].max() < tx0 - 0.3          # rotated boxes: corner gaps shrink below 0.8
            checked[1] += 1
        if 2 in row:
            assert q[e, row.index(2), :, 0].min() > tx1 + 0.3
            checked[2] += 1
        # placed before the start range was fixed = walls and parked vehicles: they stand on the back-wall side
        # (lowest vertex below y = 3); the far wall and the perturbed vehicles are beyond the start range
        early = [k for k in range(len(row)) if q[e, k, :, 1].min() < 3.0]
        if early:
            top = max(q[e, k, :, 1].max() for k in early)
            assert scenes["start"][e, 1] >= top + 1.8 - 1e-6
    assert min(checked.values()) > 0.8 * len(range(0, N, 5))          # the 5 % drop removes a few


def test_restated_generator_replays_the_reference_draw_for_draw(oracle):
    """tests/golden/generator_replay.npz: the reference's OWN ParkingLotGenerator class (map/generator/generate_parking_lot.py:19-444,
    executed where it lies by oracle/gen_golden_generator.py, with exact stand-ins for the shapely calls it makes) run on 240 seeds
    of numpy's global stream, every value np.random handed it recorded.  Fed the same draws, the build's restatement
    (t2do_generate_parking -- what the device generator is bit-identical to) must ask for them in the same ORDER and of the same
    kind, consume all of them, and produce the same scene: bay / parallel, every obstacle of Map.areas in order with its id (the
    stale side vehicles of rejected attempts, the "0003" id collision, the 5 % drops included), the target, its heading, the start
    pose (flipped or not) and the boundary."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "generator_replay.npz"))
    n = len(g["seed"])
    assert n == 240
    n_draws = n_areas = 0
    worst_q = worst_s = 0.0
    for c in range(n):
        t0, t1 = g["tape_off"][c], g["tape_off"][c + 1]
        sc, used, desync = oracle.generate_parking_replay(g["tape_kind"][t0:t1], g["tape_val"][t0:t1], float(g["type_proportion"][c]))
        assert desync == -1 and used == t1 - t0, (c, desync, used, int(t1 - t0))
        a0, a1 = g["area_off"][c], g["area_off"][c + 1]
        k = int(sc["n_quads"][0])
        assert k == a1 - a0 and sc["quad_id"][0, :k].tolist() == g["area_id"][a0:a1].tolist(), (c, sc["quad_id"][0, :k].tolist(), g["area_id"][a0:a1].tolist())
        assert bool(sc["info"][0] & oracle.GEN_BAY) == bool(g["bay"][c]), c
        if k:
            worst_q = max(worst_q, float(np.abs(sc["quads"][0, :k].astype(np.float64) - g["area_quad"][a0:a1]).max()))
        worst_q = max(worst_q, float(np.abs(sc["target"][0].astype(np.float64) - g["target"][c]).max()))
        worst_s = max(worst_s, float(np.abs(sc["start"][0] - g["start"][c]).max()), abs(float(sc["target_heading"][0]) - float(g["target_heading"][c])))
        assert np.array_equal(sc["boundary"][0], np.float32(g["boundary"][c])), (c, sc["boundary"][0], g["boundary"][c])
        n_draws += int(t1 - t0); n_areas += k
    assert worst_q < 4e-6 and worst_s < 1e-9, (worst_q, worst_s)      # (the scene's quads are fp32)
    assert n_draws > 10000 and n_areas > 1500
    assert 0.3 < g["bay"].mean() < 0.8 and set(np.unique(g["type_proportion"]).tolist()) == {0.0, 0.5, 0.8, 1.0}
