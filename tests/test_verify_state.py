"""verify_state ("very rough check", single_track_kinematics.py:200-250 & twins, point_mass.py:234-259):
oracle vs golden vectors produced by running the reference (oracle/gen_golden_verify.py), HIP kernel vs
oracle through the C ABI."""
import numpy as np
import pytest

import helpers as H


def _cases():
    g = H.load_npz("verify_state.npz")
    return g, sorted(set(int(i) for i in g["interval"]))


def test_oracle_matches_reference_golden_vectors(oracle):
    g, intervals = _cases()
    assert 0 < g["valid"].mean() < 1
    for trig in (0, 1):   # libm and deterministic trig agree away from the thresholds (margin >= 1e-7)
        for iv in intervals:
            m = g["interval"] == iv
            got = oracle.verify_state(g["rows"], g["type_id"][m], g["last"][m], g["cand"][m], iv, trig=trig)
            assert np.array_equal(got, g["valid"][m].astype(bool)), (trig, iv, int((got != g["valid"][m]).sum()))


def test_quirks_are_kept(oracle):
    g, _ = _cases()
    rows = g["rows"]
    last = np.array([[0.0, 0.0, 0.0, 5.0, 5.0, 0.0]]); cand = np.array([[0.5, 0.0, 0.0, 5.0]])
    # interval 0 -> True whatever the candidate; unbounded accel_range (row 3) -> True
    assert oracle.verify_state(rows, [0], last, np.array([[99.0, 99.0, 3.0, 50.0]]), 0)[0]
    assert oracle.verify_state(rows, [3], last, np.array([[99.0, 99.0, 3.0, 50.0]]), 100)[0]
    # heading pi: cos < 0 makes the x window empty (x_range[0] > x_range[1], strict inequalities) -> always False
    lastw = np.array([[0.0, 0.0, np.pi, 5.0, -5.0, 0.0]])
    assert not oracle.verify_state(rows, [0], lastw, np.array([[-0.5, 0.0, np.pi, 5.0]]), 100)[0]


@pytest.mark.gpu
def test_gpu_matches_oracle_and_reference(oracle):
    from tactics2d_amd import layout as L
    from tactics2d_amd.pool import ParticipantPool
    g, intervals = _cases()
    for iv in intervals:
        m = np.nonzero(g["interval"] == iv)[0]
        n = len(m)
        pool = ParticipantPool(n, 1)
        try:
            pool.set_param_table(g["rows"])
            last, cand = g["last"][m], g["cand"][m]
            pool.reset(last[:, 0], last[:, 1], last[:, 2], last[:, 3], g["type_id"][m], vx=last[:, 4], vy=last[:, 5])
            x0 = pool.download(L.F_X).copy()
            got = pool.verify_state(cand[:, 0], cand[:, 1], cand[:, 2], cand[:, 3], iv)
            assert np.array_equal(pool.download(L.F_X), x0)            # a pure check: state untouched
        finally:
            pool.close()
        want = oracle.verify_state(g["rows"], g["type_id"][m], last, cand, iv, trig=1)
        assert np.array_equal(got, want), (iv, int((got != want).sum()))
        assert np.array_equal(got, g["valid"][m].astype(bool))


@pytest.mark.gpu
def test_gpu_mirror_verify_state_and_reference_invariant():
    """tests/test_physics.py:302-303 of the reference: a step of the UNconstrained model from a state
    outside the constrained model's envelope is rejected by the constrained model's verify_state."""
    from tactics2d_amd.physics import BatchedState, SingleTrackKinematics
    con = SingleTrackKinematics(lf=1.262, lr=1.375, steer_range=(-0.524, 0.524), speed_range=(-16.67, 69.44),
                                accel_range=(-11.0, 3.121), interval=100, delta_t=5)
    unc = SingleTrackKinematics(lf=1.262, lr=1.375, interval=100, delta_t=5)
    s0 = BatchedState(frame=0, x=[10.0, 10.0], y=[10.0, 10.0], heading=[0.0, 0.0], speed=[5.0, 5.0])
    ok_next, _, _ = con.step(s0, np.float32([1.0, 1.0]), np.float32([0.1, -0.1]))
    wild, _, _ = unc.step(s0, np.float32([15.0, -15.0]), np.float32([0.0, 0.0]))
    try:
        v_ok = con.verify_state(ok_next, s0)
        v_wild = con.verify_state(wild, s0)
    finally:
        con.close(); unc.close()
    assert v_ok.all() and not v_wild.any()
