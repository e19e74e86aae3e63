"""Error behaviour and call-order rules of the C ABI (include/t2d.h), plus partial reset and the
zero-copy paths.  The reference raises Python exceptions on its path (ValueError / KeyError /
RuntimeError / InvalidAction); the ABI returns status codes that tactics2d_amd._ffi turns into T2DError."""
import ctypes as C

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _pool(n_env=4, A=2):
    from tactics2d_amd.pool import ParticipantPool
    return ParticipantPool(n_env, A)


def test_call_order_is_enforced():
    from tactics2d_amd import _ffi
    p = _pool()
    z = np.zeros(8, np.float32)
    with pytest.raises(_ffi.T2DError) as e:
        p.reset(z, z, z, z, np.zeros(8, np.uint8))
    assert e.value.code == _ffi.ERR_STATE and "t2d_set_param_table" in str(e.value)
    p.set_param_table(H.shape_rows())
    for fn in (p.integrate, p.collide, p.step, p.check_status):
        with pytest.raises(_ffi.T2DError) as e:
            fn()
        assert e.value.code == _ffi.ERR_STATE
    with pytest.raises(_ffi.T2DError) as e:
        p.set_auto_reset(True)
    assert e.value.code == _ffi.ERR_STATE and "t2d_snapshot" in str(e.value)
    with pytest.raises(_ffi.T2DError):
        p.restore()
    p.reset(z, z, z, z, np.zeros(8, np.uint8))
    p.step(100)
    p.close()


def test_invalid_arguments_are_rejected_with_messages():
    from tactics2d_amd import _ffi
    from tactics2d_amd.pool import ParticipantPool
    for bad in ((0, 1), (4, 0), (4, 257), (-1, 1)):
        with pytest.raises(_ffi.T2DError) as e:
            ParticipantPool(*bad)
        assert e.value.code == _ffi.ERR_INVALID
    p = _pool()
    rows = H.shape_rows()
    with pytest.raises(_ffi.T2DError):
        p.set_param_table(np.zeros((33, 24)))                 # more than T2D_MAX_TYPES
    bad_rows = rows.copy(); bad_rows[0, 17] = 0               # delta_t < 1 ms
    with pytest.raises(_ffi.T2DError):
        p.set_param_table(bad_rows)
    bad_rows = rows.copy(); bad_rows[0, 0] = 7                # unknown model id
    with pytest.raises(_ffi.T2DError):
        p.set_param_table(bad_rows)
    p.set_param_table(rows)
    z = np.zeros(8, np.float32)
    with pytest.raises(_ffi.T2DError) as e:                   # type id outside the table
        p.reset(z, z, z, z, np.full(8, 31, np.uint8))
    assert e.value.code == _ffi.ERR_INVALID
    p.reset(z, z, z, z, np.zeros(8, np.uint8))
    with pytest.raises(_ffi.T2DError):
        p.integrate(0)                                        # interval must be positive
    with pytest.raises(_ffi.T2DError):
        p.set_status_config(ego_index=2)                      # >= max_agents
    with pytest.raises(_ffi.T2DError):
        p.set_integrator_variant(4)
    lib = _ffi.lib()
    buf = np.zeros(3, np.float32)                             # wrong size for a field
    assert lib.t2d_download(p._h, 0, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == _ffi.ERR_INVALID
    assert lib.t2d_download(p._h, 99, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == _ffi.ERR_INVALID
    assert b"bytes" in lib.t2d_last_error(p._h)
    assert lib.t2d_bind_actions(p._h, C.c_void_p(16), None) == _ffi.ERR_INVALID
    assert lib.t2d_get_field(p._h, -1, None, None) == _ffi.ERR_INVALID
    assert lib.t2d_destroy(None) == 0 and lib.t2d_sync(None) == _ffi.ERR_INVALID
    p.close()


def test_geometry_validation():
    from tactics2d_amd import _ffi
    p = _pool(2, 1)
    quad = np.float32([[0, 0], [2, 0], [2, 2], [0, 2]])
    nine = np.float32([[np.cos(a), np.sin(a)] for a in np.linspace(0, 2 * np.pi, 9, endpoint=False)])
    for polys, code in (([[nine], []], _ffi.ERR_GEOMETRY),                                   # > 8 vertices
                        ([[np.float32([[0, 0], [1, 1]])], []], _ffi.ERR_GEOMETRY),           # < 3 vertices
                        ([[np.float32([[0, 0], [1, 1], [2, 2]])], []], _ffi.ERR_GEOMETRY),   # zero area
                        ([[np.float32([[0, 0], [4, 0], [1, 1], [0, 4]])], []], _ffi.ERR_GEOMETRY)):  # not convex
        with pytest.raises(_ffi.T2DError) as e:
            p.set_static_geometry(H.to_csr(polys))
        assert e.value.code == code, polys
    eo, vo, xy = H.to_csr([[quad], [quad]])
    with pytest.raises(_ffi.T2DError):                        # CSR offsets must start at 0 and be monotone
        p._ck(p._lib.t2d_set_static_geometry(p._h, (eo + 1).ctypes.data_as(C.c_void_p), vo.ctypes.data_as(C.c_void_p),
                                             xy.ctypes.data_as(C.c_void_p), None, None))
    with pytest.raises(ValueError):
        p.set_static_geometry((eo, vo[:-1], xy))              # host-side shape check
    with pytest.raises(_ffi.T2DError):
        p.set_target_areas(np.float32([[[0, 0], [4, 0], [1, 1], [0, 4]]] * 2))   # dart target
    p.set_static_geometry(H.to_csr([[quad[::-1]], [quad]]))   # clockwise input is accepted
    p.close()


def test_partial_reset_only_touches_masked_envs():
    from tactics2d_amd import layout as L
    p = _pool(6, 4)
    rows = H.load_npz("kin_random.npz")["rows"][:2]
    p.set_param_table(rows)
    n = 24
    x = np.arange(n, dtype=np.float32); tid = (np.arange(n) % 2).astype(np.uint8)
    p.reset(x, x + 1, x * 0.1, x * 0 + 2, tid)
    p.set_actions(np.ones(n, np.float32), np.zeros(n, np.float32))
    for _ in range(3):
        p.step(100)
    before = {f: p.download(f) for f in (L.F_X, L.F_SPEED, L.F_CNT_STEP, L.F_FRAME_MS, L.F_IDS)}
    mask = np.array([0, 1, 0, 0, 1, 0], np.uint8)
    p.reset(x * 0 - 5, x * 0, x * 0, x * 0 + 9, tid[::-1].copy(), env_mask=mask)
    after = {f: p.download(f) for f in before}
    sel = np.repeat(mask.astype(bool), 4)
    assert np.array_equal(after[L.F_X][~sel], before[L.F_X][~sel]) and (after[L.F_X][sel] == -5).all()
    assert (after[L.F_SPEED][sel] == 9).all() and np.array_equal(after[L.F_SPEED][~sel], before[L.F_SPEED][~sel])
    assert after[L.F_CNT_STEP].tolist() == [3, 0, 3, 3, 0, 3] and after[L.F_FRAME_MS].tolist() == [300, 0, 300, 300, 0, 300]
    assert np.array_equal(after[L.F_IDS][~sel], before[L.F_IDS][~sel])
    p.close()


def test_zero_copy_views_and_bound_actions():
    """t2d_get_field pointers as torch tensors (the only PyTorch touch point) and t2d_bind_actions reading the
    actions straight from a caller-owned device tensor."""
    import torch
    from tactics2d_amd import layout as L
    d = H.load_npz("kin_random.npz")
    m = d["timing"][:, 0] == 100
    st, act, tid = d["state"][m][:512], d["action"][m][:512], d["type_id"][m][:512]
    ut = np.unique(tid); remap = {int(t): i for i, t in enumerate(ut)}
    tid2 = np.array([remap[int(t)] for t in tid], np.uint8)
    p = _pool(512, 1)
    p.set_param_table(d["rows"][ut])
    p.reset(st[:, 0], st[:, 1], st[:, 2], st[:, 3], tid2)
    a0 = torch.from_numpy(act[:, 0].copy()).cuda(); a1 = torch.from_numpy(act[:, 1].copy()).cuda()
    p.bind_actions(a0.data_ptr(), a1.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    p.integrate(100, s)
    xt = torch.as_tensor(p.device_array(L.F_X), device="cuda")
    torch.cuda.synchronize()
    assert xt.dtype == torch.float32 and xt.shape == (512,)
    assert np.array_equal(xt.cpu().numpy(), p.download(L.F_X))
    want = H.gpu_physics(d["rows"], tid, st, act, 100, "fast", "kin")
    assert np.array_equal(xt.cpu().numpy(), want[:, 0])
    xt += 1.0                                                  # writes through to the pool
    torch.cuda.synchronize()
    assert np.array_equal(p.download(L.F_X), want[:, 0] + np.float32(1.0))
    p.bind_actions(None, None)                                 # back to the pool's own buffers
    p.close()


def test_idm_verify_and_drift_argument_checks():
    from tactics2d_amd import _ffi, layout as L
    p = _pool()
    rows = H.shape_rows()
    p.set_param_table(rows)
    z = np.zeros(8, np.float32)
    cid = np.zeros(8, np.uint8)
    good = np.array([[10.0, 1.5, 2.0, 1.0, 3.0, 4.0, 1.875, np.inf]])
    with pytest.raises(_ffi.T2DError) as e:                   # nothing installed yet
        p.idm_actions()
    assert e.value.code == _ffi.ERR_STATE
    for bad, why in ((good[:, :6], "8 columns"), (np.r_[good[0, :3], 0.0, good[0, 4:]][None], "positive"),
                     (np.r_[good[0, :7], -1.0][None], "horizon")):
        with pytest.raises(_ffi.T2DError) as e:
            p.set_idm(bad, cid)
        assert e.value.code == _ffi.ERR_INVALID and why in str(e.value), (why, str(e.value))
    with pytest.raises(_ffi.T2DError) as e:                   # controller id outside the table
        p.set_idm(good, np.full(8, 3, np.uint8))
    assert "out of range" in str(e.value)
    p.set_idm(good, cid)
    with pytest.raises(_ffi.T2DError) as e:                   # needs a reset first
        p.idm_actions()
    assert e.value.code == _ffi.ERR_STATE
    p.reset(z, z, z, z, np.zeros(8, np.uint8))
    p.idm_actions()
    lib = _ffi.lib()
    assert lib.t2d_verify_state(p._h, None, None, None, None, 100, None, None) == _ffi.ERR_INVALID
    ptr = C.c_void_p(p.field_ptr(L.F_X)[0]) if not isinstance(p.field_ptr(L.F_X)[0], C.c_void_p) else p.field_ptr(L.F_X)[0]
    assert lib.t2d_verify_state(p._h, ptr, ptr, ptr, ptr, -5, ptr, None) == _ffi.ERR_INVALID
    drift = np.zeros(L.PARAM_COLS); drift[[L.P_MODEL, L.P_LF, L.P_LR, L.P_WB, L.P_DELTA_T_MS]] = L.MODEL_DRIFT, 1.2, 1.3, 2.5, 5
    with pytest.raises(_ffi.T2DError) as e:                   # mass / I_z / radius / I_yw missing
        p.set_param_table(drift[None])
    assert "SingleTrackDrift" in str(e.value)
    p.close()


def test_metric_scene_keeps_four_workgroups_per_cu():
    """The 4096 x 64 metric launch is exactly one wave-round: 1024 workgroups of 4 waves on 256 CUs need 4 resident
    workgroups per CU, and LDS (static tables + the workgroup's geometry record incl. the lane-union boundary pieces)
    is what decides it -- a record that grows past the budget silently doubles the step time."""
    from tactics2d_amd import scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    sc = S.mixed(96, 64, seed=3)
    pool = ParticipantPool(sc.n_env, sc.A)
    sc.load(pool)
    blocks, lds = pool.step_occupancy()
    pool.close()
    print(f"step kernel: {blocks} workgroups / CU, {lds} B of LDS per workgroup")
    assert blocks >= 4, (blocks, lds)


def test_bound_action_memory_is_read_only_and_uploads_end_a_binding():
    """(1) IDM agents must not write into caller-owned action tensors bound with t2d_bind_actions (the
    controlled lanes take their action from the pool's own fields instead); (2) t2d_upload of an action field ends a
    binding -- the kernels would otherwise keep reading the caller's stale tensors; (3) the lidar buffer survives a
    reconfiguration with the same beam count (zero-copy views stay valid)."""
    torch = pytest.importorskip("torch")
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.controller import IDMController, install
    from tactics2d_amd.pool import ParticipantPool
    sc = S.highway(24, 64, seed=4)
    rng = np.random.default_rng(0)
    a0, a1 = sc.sample_actions(rng)
    cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
    cid[:, 1::2] = 0                                           # every other participant follows its leader
    pools = []
    for _ in range(2):
        p = ParticipantPool(sc.n_env, sc.A)
        sc.load(p)
        p.set_integrator_variant("exact")
        install(p, [IDMController(desired_speed=25.0, horizon=120.0)], cid.reshape(-1))
        pools.append(p)
    a, b = pools
    t0, t1 = torch.from_numpy(a0).cuda(), torch.from_numpy(a1).cuda()
    keep0, keep1 = t0.clone(), t1.clone()
    a.bind_actions(t0.data_ptr(), t1.data_ptr())               # pool a: caller-owned tensors
    b.set_actions(a0, a1)                                      # pool b: the pool's own fields
    for _ in range(3):
        a.step(100); b.step(100)
    torch.cuda.synchronize()
    assert torch.equal(t0, keep0) and torch.equal(t1, keep1), "IDM wrote into bound action memory"
    for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_LEADER):
        assert np.array_equal(a.download(f), b.download(f)), f
    act = a.download(L.F_ACT0).reshape(sc.n_env, sc.A)
    assert (act[:, 1::2] != a0.reshape(sc.n_env, sc.A)[:, 1::2]).any()    # the IDM accelerations live in the pool's field
    # (2) an upload ends the binding
    z = np.zeros(sc.n, np.float32)
    a.set_actions(z, z); b.set_actions(z, z)
    t0.fill_(3.0)                                              # a stale tensor nobody should read any more
    a.step(100); b.step(100)
    for f in (L.F_X, L.F_SPEED, L.F_APPLIED0):
        assert np.array_equal(a.download(f), b.download(f)), f
    a.close(); b.close()
    # (3) the lidar buffer is kept across a reconfiguration with the same beam count
    sp = S.parking(16)
    p = ParticipantPool(sp.n_env, sp.A); sp.load(p)
    p.lidar_config(360, 20.0, False)
    ptr = p.field_ptr(L.F_LIDAR)[0]
    p.lidar_config(360, 20.0, False)
    assert p.field_ptr(L.F_LIDAR)[0] == ptr
    p.lidar_config(180, 20.0, False)
    assert p.field_ptr(L.F_LIDAR)[1] == 16 * 180 * 4
    p.close()


@pytest.mark.gpu
def test_step_placement_never_changes_a_result():
    """t2d_debug_set_step_placement (a hook of libt2d_hip_debug.so, include/t2d_debug.h): any permutation of the step launch's workgroups, with any wave rotations, gives the
    same state, flags, status and rewards step after step (auto-reset on); malformed maps are rejected."""
    from tactics2d_amd import scenarios as S, layout as L
    from tactics2d_amd import debug as D
    from tactics2d_amd._ffi import T2DError
    rng = np.random.default_rng(3)
    # 1060 envs: 265 workgroups of 4 envs (four waves each: rotations apply); 96 envs: fewer workgroups than compute units at
    # four envs each, so the pool steps with one env -- one wave -- per workgroup (t2d_api.hip envs_per_workgroup)
    for n_env, n_wg, max_rot in ((1060, 265, 4), (96, 96, 1)):
        sc = S.mixed(n_env, 64, 5)
        acts = [sc.sample_actions(rng) for _ in range(6)]

        def rollout(wgmap):
            pool = D.pool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
            if wgmap is not None:
                D.set_step_placement(pool, wgmap)
            out = []
            for a0, a1 in acts:
                pool.set_actions(a0, a1); pool.step(100)
                out.append([pool.download(f).copy() for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_STATUS, L.F_REWARD)])
            pool.close()
            return out
        want = rollout(None)
        perm = rng.permutation(n_wg).astype(np.uint32)
        rot = rng.integers(0, max_rot, n_wg).astype(np.uint32)
        got = rollout(perm | (rot << 16))
        for w, g in zip(want, got):
            for a, b in zip(w, g):
                assert np.array_equal(a, b, equal_nan=True)
        pool = D.pool(sc.n_env, sc.A); sc.load(pool)
        bad = perm.copy(); bad[0] = bad[1]
        for m in (bad, perm[:-1], perm | np.uint32(4 << 16)):
            with pytest.raises(T2DError):
                D.set_step_placement(pool, m)
        D.set_step_placement(pool, None)
        pool.close()


@pytest.mark.gpu
def test_strided_action_binding_equals_split_arrays():
    """t2d_bind_actions_strided: a policy's [N, 2] (steering, accel) tensor read in place gives the step the split
    contiguous arrays give -- fused step, two-launch step, single-ego kernel; stride < 1 is rejected."""
    import torch
    from tactics2d_amd import scenarios as S, layout as L
    from tactics2d_amd.pool import ParticipantPool
    from tactics2d_amd._ffi import T2DError
    dev = torch.device("cuda", 0)
    for sc, fused in ((S.mixed(48, 64, 7), True), (S.mixed(48, 64, 7), False), (S.parking(200), True)):
        rng = np.random.default_rng(11)
        outs = []
        for strided in (False, True):
            pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_fused_step(fused)
            res = []
            keep = []
            for k in range(5):
                a0, a1 = sc.sample_actions(np.random.default_rng(100 + k))
                if strided:
                    t = torch.from_numpy(np.stack([a1, a0], 1).copy()).to(dev)     # [N, 2] = (steering, accel)
                    pool.bind_actions(t.data_ptr() + 4, t.data_ptr(), stride=2)
                else:
                    t = (torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev))
                    pool.bind_actions(t[0].data_ptr(), t[1].data_ptr())
                keep.append(t)
                pool.step(100)
                res.append([pool.download(f).copy() for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_APPLIED0, L.F_APPLIED1, L.F_FLAGS)])
            pool.sync()
            outs.append(res)
            if strided:
                with pytest.raises(T2DError):
                    pool.bind_actions(keep[-1].data_ptr() + 4, keep[-1].data_ptr(), stride=0)
            pool.close()
        for ra, rb in zip(*outs):
            for a, b in zip(ra, rb):
                assert np.array_equal(a, b, equal_nan=True)
