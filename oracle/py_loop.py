"""Reference-style per-participant Python loop (TEST INFRASTRUCTURE; CPU baseline of bench.py only).

SURVEY.md 8(d) asks for the call pattern of the reference -- one Python-level `step` per participant with
numpy *scalar* ufunc calls inside the 20 sub-steps (physics/single_track_kinematics.py:126-198) -- timed next
to the C restatement, because that per-call overhead is where the reference spends its time.  The reference
itself cannot travel to the GPU box, so this is the build's own restatement of SingleTrackKinematics.step
(Appendix A.2 of SURVEY.md); it is checked against the golden vectors in tests/test_oracle_physics.py.
"""
import numpy as np

P_LR, P_WB, P_STEER_LO, P_STEER_HI, P_SPEED_LO, P_SPEED_HI, P_ACCEL_LO, P_ACCEL_HI, P_FLAGS, P_DT = 2, 3, 4, 5, 6, 7, 8, 9, 10, 17


def kinematics_step(row, x, y, phi, v, accel, delta, interval):
    flags = int(row[P_FLAGS])
    if flags & 4:
        accel = np.clip(accel, row[P_ACCEL_LO], row[P_ACCEL_HI])
    if flags & 1:
        delta = np.clip(delta, row[P_STEER_LO], row[P_STEER_HI])
    lr, wb, delta_t = row[P_LR], row[P_WB], int(row[P_DT])
    beta = np.arctan(lr / wb * np.tan(delta))
    dts = [float(delta_t) / 1000] * (interval // delta_t)
    if interval % delta_t > 0:
        dts.append(float(interval % delta_t) / 1000)
    for dt in dts:
        dx = v * np.cos(phi + beta)
        dy = v * np.sin(phi + beta)
        dphi = v / wb * np.tan(delta) * np.cos(beta)
        x += dx * dt
        y += dy * dt
        phi += dphi * dt
        v += accel * dt
        if flags & 2:
            v = np.clip(v, row[P_SPEED_LO], row[P_SPEED_HI])
    return x, y, np.mod(phi, 2 * np.pi), v, v * np.cos(phi), v * np.sin(phi), accel, delta


def time_loop(row, n_steps=2000, seed=0):
    """-> participant-steps/s of the loop above on one core (physics only: the reference's event detectors
    live in shapely, which is not available here)."""
    import time
    rng = np.random.default_rng(seed)
    acts = np.stack([rng.uniform(-3, 2, n_steps), rng.normal(0, 0.02, n_steps)], 1)
    x, y, phi, v = 0.0, 0.0, 0.0, 5.0
    t0 = time.perf_counter()
    for k in range(n_steps):
        x, y, phi, v, _, _, _, _ = kinematics_step(row, x, y, phi, v, acts[k, 0], acts[k, 1], 100)
    return n_steps / (time.perf_counter() - t0)
