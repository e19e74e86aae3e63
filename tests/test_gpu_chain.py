"""t2d_step_n: n steps enqueued by one call -- for pools stepped by the fused kernel alone ONE launch in which workgroup
(g, k) takes step k of the envs of workgroup g, ordered after (g, k - 1) by a counter in device memory.  The contract is
exact: every pool field, and every slot of the record ring, equals what n separate t2d_step calls leave behind.
(Reference loop: envs/parking.py:219-256, one env.step() after the other.)"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields():
    from tactics2d_amd import layout as L
    return (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_APPLIED0, L.F_APPLIED1, L.F_IDS, L.F_FLAGS,
            L.F_ENV_FLAGS, L.F_CNT_STEP, L.F_FRAME_MS, L.F_STATUS, L.F_REWARD, L.F_RECORD, L.F_IOU, L.F_CNT_NO_ACTION)


def _pool(sc, variant, ego_kernel=True):
    from tactics2d_amd.pool import ParticipantPool
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    pool.set_integrator_variant(variant)
    pool.set_auto_reset(True)
    if not ego_kernel:
        pool._ck(pool._lib.t2d_set_ego_kernel(pool._h, 0))
    return pool


def _compare(sc, n_steps, variant, calls=(None,), ego_kernel=True, same_actions=False, chaining=1, form=None, split=True):
    """reference: n_steps t2d_step calls on an action ring; candidate: the same ring through t2d_step_n, split into `calls`"""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    n = sc.n
    sets = [sc.sample_actions(rng) for _ in range(1 if same_actions else n_steps)]
    a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()   # [steps][N]
    a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
    stride = 0 if same_actions else n
    ref = _pool(sc, variant, ego_kernel)
    for k in range(n_steps):
        ref.bind_actions(a0.data_ptr() + 4 * stride * k, a1.data_ptr() + 4 * stride * k)
        ref.step(sc.interval_ms)
    want = [ref.download(f) for f in _fields()]
    ref.close()
    got_pool = _pool(sc, variant, ego_kernel)
    got_pool.set_step_chaining(chaining)
    got_pool.set_split_step(split)
    if form is not None:
        assert got_pool.step_form(max(2, n_steps)) == form
    done = 0
    for c in calls:
        c = n_steps - done if c is None else c
        got_pool.bind_actions(a0.data_ptr() + 4 * stride * done, a1.data_ptr() + 4 * stride * done)
        got_pool.step_n(c, sc.interval_ms, stride)
        done += c
    assert done == n_steps
    got = [got_pool.download(f) for f in _fields()]
    got_pool.close()
    for f, g, w in zip(_fields(), got, want):
        assert np.array_equal(g, w, equal_nan=True), (f, int((g != w).sum()))
    return want


@pytest.mark.parametrize("variant", ["exact", "fast"])
def test_chained_steps_equal_single_steps_on_the_mixed_scene(variant):
    from tactics2d_amd import layout as L, scenarios as S
    sc = S.mixed(203, 64, seed=5)          # 51 workgroups: the grid is padded to 56, the last workgroup is ragged
    want = _compare(sc, 24, variant)
    st = want[_fields().index(L.F_STATUS)]
    rec = want[_fields().index(L.F_RECORD)].reshape(L.RECORD_RING, sc.n_env, 2)
    assert (rec[:24, :, 1] >> 16).astype(bool).any(), "no episode ended in 24 steps: the auto-reset path was not exercised"
    assert st.shape == (sc.n_env, 4)


def test_chained_steps_in_several_calls_and_past_the_record_ring():
    from tactics2d_amd import scenarios as S
    sc = S.intersection(64, 32, seed=8)    # A = 32: two envs per wave, eight per workgroup
    _compare(sc, 120, "exact", calls=(3, 90, 1, 26))   # 90 > the ring of 64 slots: two launches inside one call; 1: plain step


def test_chained_steps_repeat_one_action_set():
    from tactics2d_amd import scenarios as S
    sc = S.highway(40, 64, seed=2)
    _compare(sc, 12, "fast", same_actions=True)


def test_chained_steps_with_iou_events_on_the_general_kernel():
    """parking envs (Arrival / NoAction IoU, shaped reward) kept on the general kernel: the IoU instantiation of the chain;
    with the single-ego kernel t2d_step_n takes ordinary launches -- same answers either way"""
    from tactics2d_amd import scenarios as S
    sc = S.parking(300)
    _compare(sc, 20, "exact", ego_kernel=False)
    _compare(sc, 20, "exact", ego_kernel=True)


def test_chaining_off_and_idm_pools_fall_back_to_single_launches():
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.controller import IDMController, install
    from tactics2d_amd.pool import ParticipantPool
    sc = S.highway(24, 64, seed=3)
    rng = np.random.default_rng(0)
    a0, a1 = sc.sample_actions(rng)
    outs = []
    for mode in ("steps", "step_n", "step_n_unchained"):
        pool = ParticipantPool(sc.n_env, sc.A)
        sc.load(pool)
        cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
        cid[:, 1:] = 0
        install(pool, [IDMController(desired_speed=25.0, horizon=120.0)], cid.reshape(-1))
        pool.set_actions(a0, a1)
        if mode == "step_n_unchained":
            pool.set_step_chaining(False)
        if mode == "steps":
            for _ in range(6):
                pool.step(100)
        else:
            pool.step_n(6, 100, 0)
        outs.append([pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_RECORD)])
        pool.close()
    for o in outs[1:]:
        for g, w in zip(o, outs[0]):
            assert np.array_equal(g, w)


def test_chained_steps_at_the_metric_size():
    """4096 x 64: 1024 workgroups per step = one wave-round of the GPU, so step k + 1's workgroups start as step k's retire"""
    from tactics2d_amd import scenarios as S
    sc = S.mixed(4096, 64, seed=3)
    _compare(sc, 16, "fast")


def test_single_ego_pools_loop_through_arrivals_no_action_and_time_limits():
    """The single-ego kernel's LOOP form carries the state, the counters and the detector history (previous pose, NoAction
    counter, _max_iou, _min_dist) in registers from one step to the next: egos parked on the bay (Arrival), egos that never
    move (NoAction after 3 steps), a 12-step time limit, auto-reset on -- 64 steps as fragments of 1..32 against 64 calls."""
    from tactics2d_amd import scenarios as S
    sc = S.parking(777, seed0=5)
    rng = np.random.default_rng(12)
    tc = sc.target.mean(1)
    on = np.arange(sc.n_env) % 3 == 0
    sc.x[on] = tc[on, 0] + rng.normal(0, 0.03, on.sum()).astype(np.float32)
    sc.y[on] = tc[on, 1] + rng.normal(0, 0.03, on.sum()).astype(np.float32)
    sc.heading[on] = sc.target_heading[on]
    sc.status.update(max_step=12, no_action_max_step=3)
    still = np.arange(sc.n_env) % 3 == 1

    class Calm:   # an action source whose `still` / `on` egos never accelerate
        def __init__(self, sc):
            self.sc = sc
        def __getattr__(self, k):
            return getattr(self.sc, k)
        def sample_actions(self, r):
            a0, a1 = self.sc.sample_actions(r)
            a0[still | on] = 0.0
            return a0, a1
    # (both multi-step forms of the single-ego kernel: integrator waves a step ahead of the event waves, and the plain loop)
    _compare(Calm(sc), 64, "exact", calls=(1, 2, 29, 32), chaining=3, form="ego_loop")
    want = _compare(Calm(sc), 64, "exact", calls=(1, 2, 29, 32), form="ego_loop_pipe")
    from tactics2d_amd import layout as L
    rec = want[_fields().index(L.F_RECORD)].reshape(L.RECORD_RING, sc.n_env, 2)
    seen = set(map(tuple, np.stack([rec[..., 1] & 0xff, (rec[..., 1] >> 8) & 0xff], -1).reshape(-1, 2).tolist()))
    assert {(1, 1), (2, 1), (1, 5), (3, 1)} <= seen, seen     # normal, completed, no-action quirk, time exceeded
    _compare(Calm(sc), 40, "fast", calls=(40,))


def test_one_workgroup_per_env_form_equals_one_wave_per_env():
    """Small pools of 64-agent envs with obstacles and lanes run each env's event stages on four waves side by side
    (collide_kernel<..., SPLIT>): every field after 40 steps -- as single launches and as chained fragments -- equals what the
    one-wave-per-env form leaves behind, auto-resets included.  A highway pool (no static obstacles) does not take that form."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    dev = torch.device("cuda", 0)
    for sc, A_real in ((S.mixed(150, 64, seed=11), 64), (S.mixed(37, 40, seed=12), 40)):   # 40: 24 empty lanes per env
        rng = np.random.default_rng(7)
        sets = [sc.sample_actions(rng) for _ in range(40)]
        a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
        a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
        outs = {}
        for split in (False, True):
            for chained in (False, True):
                pool = _pool(sc, "exact")
                pool.set_split_step(split)
                pool.set_step_chaining(2)    # (the chained form: by default a pool this small loops with integrator waves)
                assert pool.step_form(1) == ("step_split" if split else "step")
                assert pool.step_form(8) == ("chain_split" if split else "chain")
                if chained:
                    done = 0
                    for c in (7, 1, 32):
                        pool.bind_actions(a0.data_ptr() + 4 * sc.n * done, a1.data_ptr() + 4 * sc.n * done)
                        pool.step_n(c, sc.interval_ms, sc.n)
                        done += c
                else:
                    for k in range(40):
                        pool.bind_actions(a0.data_ptr() + 4 * sc.n * k, a1.data_ptr() + 4 * sc.n * k)
                        pool.step(sc.interval_ms)
                outs[(split, chained)] = [pool.download(f) for f in _fields()]
                pool.close()
        want = outs[(False, False)]
        rec = want[_fields().index(L.F_RECORD)].reshape(L.RECORD_RING, sc.n_env, 2)
        assert (rec[:40, :, 1] >> 16).astype(bool).any(), "no episode ended: the reset path of the split form was not exercised"
        fl = want[_fields().index(L.F_FLAGS)]
        assert (fl & 1).any() and (fl & 8).any(), "no collision / off-lane flag in the comparison"
        for key, got in outs.items():
            for f, g, w in zip(_fields(), got, want):
                assert np.array_equal(g, w, equal_nan=True), (key, f, int((g != w).sum()))
    hw = _pool(S.highway(64, 64, seed=2), "exact")
    assert hw.step_form(1) == "step" and hw.step_form(8) == "loop_pipe"
    hw.set_step_chaining(2)
    assert hw.step_form(8) == "chain"
    hw.close()


@pytest.mark.parametrize("chaining,form", [(1, "loop_pipe"), (4, "loop_pipe"), (3, "loop"), (2, "chain")])
def test_every_multi_step_form_of_a_small_pool_equals_single_steps(chaining, form):
    """A pool of at most one workgroup per CU has three multi-step forms: workgroups chained per step, resident workgroups
    looping over the steps, and the loop with integrator waves a step ahead of the event waves (the default; with lane
    polygons in the pool a third set of waves takes the lane stage -- 4 keeps that off).  Each against
    the same steps as single launches -- auto-resets in most steps (the integrator waves' speculation is wrong there and
    they integrate again from the snapshot), fragments of 1..33 steps, 64-agent envs (one per wave) and 32-agent envs (two per
    wave, verdicts of both in one integrator wave)."""
    from tactics2d_amd import layout as L, scenarios as S
    for sc, variant, calls in ((S.mixed(203, 64, seed=5), "fast", (1, 2, 33, 4)), (S.intersection(100, 32, seed=8), "exact", (17, 23)),
                               (S.highway(77, 64, seed=2), "exact", (40,))):
        if form == "chain" and sc.A == 64 and sc.name != "highway":
            form_here = "chain_split"     # (pools with static obstacles and lanes chain one workgroup per env)
        else:
            form_here = form
        want = _compare(sc, sum(calls), variant, calls=calls, chaining=chaining, form=form_here, split=chaining != 3)
        rec = want[_fields().index(L.F_RECORD)].reshape(L.RECORD_RING, sc.n_env, 2)
        ended = (rec[:sum(calls), :, 1] >> 16).astype(bool)
        assert ended.any(1).sum() >= sum(calls) // 2, "too few steps with an episode end: the reset path was hardly exercised"


@pytest.mark.parametrize("name", ["cfg3", "cfg4", "cfg5"])
def test_baseline_shards_at_full_size_as_fragments(name):
    """BASELINE.json's configs 3-5 at their per-GPU sizes (1024 x 64 highway, 512 x 32 intersection, 1024 x 64 mixed: one
    workgroup per CU and fewer) as t2d_step_n fragments -- the form bench.py's `configs` times -- against single launches."""
    from tactics2d_amd import scenarios as S
    sc = {"cfg3": lambda: S.highway(1024, 64, seed=1), "cfg4": lambda: S.intersection(512, 32, seed=2),
          "cfg5": lambda: S.mixed(1024, 64, seed=3)}[name]()
    _compare(sc, 48, "fast", calls=(32, 16), form="loop_pipe")


@pytest.mark.parametrize("bound,chaining,form", [(False, 1, "loop_pipe"), (True, 1, "loop_pipe"), (True, 2, "chain")])
def test_idm_pools_run_their_controllers_inside_the_fragment(bound, chaining, form):
    """A pool with installed IDM controllers (t2d_step = idm_kernel + step launch per step): as t2d_step_n the PIPE form's
    integrator waves run the controllers themselves ahead of every step, and so does every workgroup of the chained form
    (larger pools) ahead of its integrator -- leaders, accelerations (the pool's action field), states, flags and records
    equal to the separate launches over 40 steps with auto-resets, with the other participants' actions in the pool's own
    fields or bound as a device-resident ring."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.controller import IDMController, install
    dev = torch.device("cuda", 0)
    for sc in (S.highway(70, 64, seed=4), S.intersection(50, 32, seed=6)):
        rng = np.random.default_rng(3)
        n_steps = 40
        sets = [sc.sample_actions(rng) for _ in range(n_steps)]
        a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
        a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
        veh = (sc.rows[sc.type_id, L.P_MODEL] != L.MODEL_POINTMASS).reshape(sc.n_env, sc.A)
        cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
        cid[:, 1:] = np.where(veh[:, 1:], np.arange(sc.A - 1)[None, :] % 2, L.IDM_NONE)   # two controllers; slot 0 = the caller's
        outs = []
        for mode in ("steps", "fragments", "fused_steps"):
            pool = _pool(sc, "exact")
            pool.set_step_chaining(0 if mode == "steps" else chaining)   # 0: idm_kernel stays a launch of its own
            install(pool, [IDMController(desired_speed=25.0, horizon=120.0), IDMController(desired_speed=12.0, horizon=60.0, time_headway=1.0)],
                    cid.reshape(-1))
            if not bound:
                pool.set_actions(sets[0][0], sets[0][1])
            if mode != "fragments":
                assert pool.step_form(1) == ("unfused" if mode == "steps" else "step")
                for k in range(n_steps):
                    if bound:
                        pool.bind_actions(a0.data_ptr() + 4 * sc.n * k, a1.data_ptr() + 4 * sc.n * k)
                    pool.step(sc.interval_ms)
            else:
                assert pool.step_form(8) == form
                done = 0
                for c in (3, 32, 5):
                    if bound:
                        pool.bind_actions(a0.data_ptr() + 4 * sc.n * done, a1.data_ptr() + 4 * sc.n * done)
                    pool.step_n(c, sc.interval_ms, sc.n if bound else 0)
                    done += c
            fields = _fields() + (L.F_ACT0, L.F_ACT1, L.F_LEADER)
            outs.append([pool.download(f) for f in fields])
            pool.close()
        for got in outs[1:]:
            for f, g, w in zip(fields, got, outs[0]):
                assert np.array_equal(g, w, equal_nan=True), (sc.name, bound, f, int((g != w).sum()))
        lead = outs[0][fields.index(L.F_LEADER)]
        rec = outs[0][fields.index(L.F_RECORD)].reshape(L.RECORD_RING, sc.n_env, 2)
        assert (lead >= 0).mean() > 0.2 and (rec[:n_steps, :, 1] >> 16).astype(bool).any()


@pytest.mark.parametrize("A", [2, 3, 7, 16, 24, 33, 50])
def test_fragments_of_pools_of_any_env_width(A):
    """Envs of 2..64 participants map 32..1 envs to a wave (the env's lanes are padded to a power of two), and the last
    workgroup of a pool is ragged: the looping forms with their integrator / lane waves against single launches for widths
    that are not powers of two, pools of one env, and env counts that fill no workgroup."""
    from tactics2d_amd import mapgeom as MG, scenarios as S
    ran = 0
    for n_env, maker in ((1, S.intersection), (5, S.highway), (67, S.intersection), (130, S.highway)):
        sc = maker(n_env, A, seed=A + n_env)
        # (narrow envs put up to 128 of them into a workgroup: their polygons may not fit its LDS record -- such a pool steps
        # through the HBM grid tier, one launch per stage: none of the looping forms, tests/test_gpu_mapgrid.py)
        if not MG.geometry_budget(sc.n_env, sc.A, static=sc.static, lanes=sc.lanes)["fits"]:
            assert A < 16
            continue
        for chaining in (1, 3):
            _compare(sc, 24, "exact", calls=(1, 19, 4), chaining=chaining, form="loop_pipe" if chaining == 1 else "loop", split=False)
        ran += 1
    assert ran >= 2


def test_idm_pool_of_more_workgroups_than_cus_chains_with_its_controllers():
    """1100 x 64 = 275 workgroups: too many for the looping forms with integrator waves; the chained form runs the controllers"""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.controller import IDMController, install
    sc = S.mixed(1100, 64, seed=9)
    rng = np.random.default_rng(1)
    a0, a1 = sc.sample_actions(rng)
    veh = (sc.rows[sc.type_id, L.P_MODEL] != L.MODEL_POINTMASS).reshape(sc.n_env, sc.A)
    cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
    cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
    outs = []
    fields = _fields() + (L.F_ACT0, L.F_ACT1, L.F_LEADER)
    for mode in ("steps", "fragments"):
        pool = _pool(sc, "fast")
        install(pool, [IDMController(desired_speed=20.0, horizon=100.0)], cid.reshape(-1))
        pool.set_actions(a0, a1)
        if mode == "steps":
            pool.set_step_chaining(0)   # idm_kernel + step launch per step
            assert pool.step_form(1) == "unfused"
            for _ in range(24):
                pool.step(sc.interval_ms)
        else:
            assert pool.step_form(1) == "step"   # (a single t2d_step runs the controllers in its launch as well)
            assert pool.step_form(24) == "chain"
            pool.step_n(24, sc.interval_ms, 0)
        outs.append([pool.download(f) for f in fields])
        pool.close()
    for f, g, w in zip(fields, outs[1], outs[0]):
        assert np.array_equal(g, w, equal_nan=True), (f, int((g != w).sum()))


def test_a_declared_action_extent_stops_a_fragment_that_would_read_past_the_ring():
    """t2d_set_action_extent: the library cannot see how large caller-owned action memory is; told, t2d_step_n refuses the fragment
    whose last step would read beyond it (T2D_ERR_INVALID, nothing launched) instead of faulting on the device; a rebind forgets
    the extent; the pool's own action fields take none"""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd._ffi import T2DError
    dev = torch.device("cuda", 0)
    sc = S.highway(8, 64, seed=3)
    n, K = sc.n, 4
    rng = np.random.default_rng(1)
    sets = [sc.sample_actions(rng) for _ in range(K)]
    a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
    a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
    pool = _pool(sc, "exact")
    with pytest.raises(T2DError):                       # nothing bound: the pool reads its own fields
        pool._ck(pool._lib.t2d_set_action_extent(pool._h, K * n))
    pool.bind_actions(a0.data_ptr(), a1.data_ptr(), extent=K * n)
    pool.step_n(K, sc.interval_ms, n)                   # the whole ring: fine
    count = pool.step_count()
    before = pool.download(L.F_X).copy()
    with pytest.raises(T2DError, match="extent"):
        pool.step_n(K + 1, sc.interval_ms, n)           # one set too many
    assert pool.step_count() == count and np.array_equal(pool.download(L.F_X), before)   # refused before anything ran
    pool.step_n(K + 1, sc.interval_ms, 0)               # one set repeated reads the first set only
    with pytest.raises(T2DError):                       # an extent that does not hold one set
        pool._ck(pool._lib.t2d_set_action_extent(pool._h, n - 1))
    pool.bind_actions(a0.data_ptr(), a1.data_ptr())    # a rebind: the extent is unknown again, the caller is on their own
    pool.step_n(K, sc.interval_ms, n)
    pool.close()
