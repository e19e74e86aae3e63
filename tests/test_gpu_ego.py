"""The single-ego step kernel (t2d_ego.hip: sixteen lanes per environment, ParkingEnv / BASELINE config 2) against the general
kernel (one lane per participant) on identical pools: every field the step writes must agree bit for bit, step after step,
through collisions, arrivals, no-action stretches, time limits and auto-resets.  (The general kernel is held against the
oracle in tests/test_gpu_collide.py, test_iou_events.py and test_gpu_envs.py; step_torch, which runs this kernel, against
the oracle-checked numpy step in test_gpu_envs.py.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields():
    from tactics2d_amd import layout as L
    return [L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_APPLIED0, L.F_APPLIED1, L.F_FLAGS, L.F_ENV_FLAGS,
            L.F_STATUS, L.F_REWARD, L.F_IOU, L.F_CNT_NO_ACTION, L.F_CNT_STEP, L.F_FRAME_MS, L.F_RECORD, L.F_IDS]


def _pair(sc, variant, auto_reset=True):
    from tactics2d_amd.pool import ParticipantPool
    pools = []
    for ego in (True, False):
        p = ParticipantPool(sc.n_env, sc.A)
        sc.load(p)
        p.set_integrator_variant(variant)
        p.set_auto_reset(auto_reset)
        p.set_ego_kernel(ego)
        pools.append(p)
    return pools


def _compare(a, b, t):
    for f in _fields():
        x, y = a.download(f), b.download(f)
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"step {t}: field {f} differs in {int((x != y).sum())} places"


def test_fast_integrator_agrees_to_rounding_noise():
    """The fast integrator picks its sub-step loop per WAVE (a wave in which some lane clips its speed takes the plain
    loop: DESIGN.md 3), and the two kernels put different environments into a wave: from identical states one step
    may differ in the last bits of the fp64 result -- never by more than an fp32 ulp of the stored state."""
    from tactics2d_amd import layout as L, scenarios as S
    sc = S.parking(2000, seed0=55)
    rng = np.random.default_rng(1)
    sc.speed[:] = rng.uniform(-0.5, 0.5, sc.n).astype(np.float32)
    a, b = _pair(sc, "fast", auto_reset=False)
    a0, a1 = sc.sample_actions(rng)
    for p in (a, b):
        p.set_actions(a0, a1); p.step(100)
    for f in (L.F_X, L.F_Y, L.F_SPEED, L.F_VX, L.F_VY):
        assert np.abs(a.download(f).astype(np.float64) - b.download(f)).max() <= 2e-6, f
    assert np.abs(a.download(L.F_HEADING).astype(np.float64) - b.download(L.F_HEADING)).max() <= 1e-6
    assert (a.download(L.F_FLAGS) != b.download(L.F_FLAGS)).mean() < 1e-3
    a.close(); b.close()


@pytest.mark.parametrize("variant", ["exact"])
def test_wave_per_env_kernel_equals_the_general_kernel_on_parking_scenes(variant):
    from tactics2d_amd import scenarios as S
    n = 1500
    sc = S.parking(n, seed0=321)
    rng = np.random.default_rng(3)
    # a third of the egos starts on its target bay (arrival), a third never acts (no-action), the rest drive
    tc = sc.target.mean(1)
    on = np.arange(n) % 3 == 0
    sc.x[on] = tc[on, 0] + rng.normal(0, 0.03, on.sum()).astype(np.float32)
    sc.y[on] = tc[on, 1] + rng.normal(0, 0.03, on.sum()).astype(np.float32)
    sc.heading[on] = sc.target_heading[on]
    sc.boundary[::5, 1] = sc.boundary[::5, 0] + 18.0           # some maps are tight: out-of-bound fires
    sc.status.update(max_step=60, no_action_max_step=6)
    a, b = _pair(sc, variant)
    still = np.arange(n) % 3 == 1
    seen = set()
    for t in range(140):
        a0, a1 = sc.sample_actions(rng)
        a0[still | on] = 0.0
        if t % 9 == 4:
            a0[:] = 2.0; a1[::2] = 0.524                        # full lock, full throttle: into the parked cars
        for p in (a, b):
            p.set_actions(a0, a1)
            p.step(100)
        _compare(a, b, t)
        from tactics2d_amd import layout as L
        seen |= set(map(tuple, a.download(L.F_STATUS)[:, :2].tolist()))
    a.close(); b.close()
    assert {(1, 1), (2, 1), (1, 5), (3, 1), (4, 1)} <= seen, seen     # (static collisions: the next test)


def test_wave_per_env_kernel_on_generated_lots_and_odd_sizes():
    """Device-generated parking lots (capacity layout: 12 quad slots per env, unused ones boxed out), env counts that do
    not fill the last workgroup, triangles among the obstacles, no target areas, no boundary."""
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    rng = np.random.default_rng(8)
    # (a) generated scenes
    pools = []
    for ego in (True, False):
        p = ParticipantPool(1027, 1)
        sc = S.parking(1, seed0=0)
        p.set_param_table(sc.rows)
        p.set_status_config(**dict(sc.status, max_step=40))
        p.parking_scenes(77, 0.5, (4.284, 1.799), regenerate=False)
        p.set_auto_reset(True)
        p.set_integrator_variant("exact")
        p.set_ego_kernel(ego)
        pools.append(p)
    a, b = pools
    ended = 0
    for t in range(60):
        a0 = rng.uniform(-2, 2, 1027).astype(np.float32); a1 = rng.uniform(-0.524, 0.524, 1027).astype(np.float32)
        for p in (a, b):
            p.set_actions(a0, a1); p.step(100)
        _compare(a, b, t)
        ended += int(a.download(L.F_STATUS)[:, 3].sum())
    assert ended >= 1027                                       # every env ran into the 40-step limit at least once
    a.close(); b.close()
    # (b) plain static scene with triangles, no targets / boundary / IoU events; up to 18 obstacles per env (two rounds of 16 lanes)
    import helpers as H
    n = 203
    sc = H.random_scene(rng, n, 1, (30.0, 20.0), n_static=18, with_peds=False, inactive_frac=0.05, bounded=False)
    eo, vo, xy = sc["static"]
    keep = np.ones(len(xy), bool)
    for q in range(0, len(vo) - 1, 3):                         # every third quad loses a vertex: a triangle
        keep[vo[q + 1] - 1] = False
    cnt = np.diff(vo).copy(); cnt[0::3] -= 1
    vo2 = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    static = (eo, vo2, xy[keep])
    rows = sc["rows"].copy(); rows[:, 0] = 0; rows[:, 10] = 0  # kinematics, unbounded ranges
    pools = []
    for ego in (True, False):
        p = ParticipantPool(n, 1)
        p.set_param_table(rows)
        p.set_static_geometry(static, None, None)
        p.set_status_config(max_step=30)
        p.reset(sc["x"], sc["y"], sc["heading"], np.full(n, 2.0, np.float32), sc["type_id"], sc["active"])
        p.snapshot(); p.set_auto_reset(True); p.set_ego_kernel(ego); p.set_integrator_variant("exact")
        pools.append(p)
    a, b = pools
    for t in range(45):
        a0 = rng.uniform(-2, 2, n).astype(np.float32); a1 = rng.uniform(-0.5, 0.5, n).astype(np.float32)
        for p in (a, b):
            p.set_actions(a0, a1); p.step(100)
        _compare(a, b, t)
    fl = a.download(L.F_FLAGS)
    assert 0.02 < (fl & 2).astype(bool).mean() < 0.98
    a.close(); b.close()
