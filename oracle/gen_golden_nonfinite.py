#!/usr/bin/env python3
"""Golden vectors for NON-FINITE inputs, made by importing the reference physics (TEST INFRASTRUCTURE; build container only).

RL policies do emit NaN; the reference lets it through: `np.clip(nan, lo, hi)` is nan (single_track_kinematics.py:192-193),
`np.clip(+-inf, lo, hi)` is the bound, `np.mod(+-inf, 2 pi)` is nan.  This script steps SingleTrackKinematics /
SingleTrackDynamics / PointMass once from a finite base case with ONE input (a state field or an action component) replaced
by nan, +inf or -inf and records what the reference returns -- data only -> tests/golden/nonfinite.npz:

    rows[n_types, 24], type_id[n], model[n] (0 kin, 1 dyn, 2 pm), state[n, 4], action[n, 2], interval[n],
    out[n, 6] (x, y, heading, speed, vx, vy: fp64, nan where the reference has nan or None), applied[n, 2]

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_nonfinite.py [--ref /root/reference]
"""
import argparse
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import DYN, KIN, PM, row_from_model, state_out  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden", "nonfinite.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    from tactics2d.participant.trajectory import State
    from tactics2d.physics import PointMass, SingleTrackDynamics, SingleTrackKinematics
    warnings.simplefilter("ignore")   # (np.mod(inf), inf - inf: the reference computes on)

    MED = dict(lf=4.284 / 2 - 0.880, lr=4.284 / 2 - 0.767)
    med = dict(steer_range=(-0.524, 0.524), speed_range=(-16.67, 69.44), accel_range=(-11.0, 3.121))
    park = dict(steer_range=(-0.524, 0.524), speed_range=(-0.5, 0.5), accel_range=(-2.0, 2.0))
    rigs = [
        (KIN, SingleTrackKinematics(**MED, **med, interval=100), [(3.0, -2.0, 0.7, 6.0), (1.0, 0.1)]),
        (KIN, SingleTrackKinematics(**MED, **park, interval=100), [(1.25, -3.5, 0.3, 0.2), (1.0, 0.2)]),
        (KIN, SingleTrackKinematics(**MED, interval=100), [(3.0, -2.0, 0.7, 6.0), (1.0, 0.1)]),              # no range at all
        (KIN, SingleTrackKinematics(**MED, **med, interval=9, delta_t=5), [(3.0, -2.0, 0.7, 6.0), (1.0, 0.1)]),  # remainder sub-step
        (DYN, SingleTrackDynamics(**MED, mass=1620, mass_height=0.726, **med, interval=100), [(0.0, 0.0, 1.0, 12.0), (1.0, 0.05)]),
        (DYN, SingleTrackDynamics(**MED, mass=1620, mass_height=0.726, **med, interval=100), [(0.0, 0.0, 1.0, 0.05), (1.0, -0.2)]),
        (DYN, SingleTrackDynamics(**MED, mass=1620, mass_height=0.726, interval=100), [(0.0, 0.0, 1.0, 12.0), (1.0, 0.05)]),
        (PM, PointMass(speed_range=(-7.0, 7.0), accel_range=(-1.5, 1.5), interval=100), [(0.0, 0.0, 1.0, 0.5), (1.0, -0.5)]),
        (PM, PointMass(speed_range=(0.5, 1.2), interval=100), [(0.0, 0.0, 1.0, 0.5), (3.0, 1.0)]),
        (PM, PointMass(interval=100), [(1.0, 2.0, -0.3, 0.4), (2.0, -1.0)]),
    ]
    rows, type_id, model_id, st, act, ivl, out, app, raised = [], [], [], [], [], [], [], [], 0
    for tid, (mid, model, (s0, a0)) in enumerate(rigs):
        rows.append(row_from_model(model, mid))
        interval = int(model.interval)
        s0 = [float(np.float32(v)) for v in s0]   # (inputs are fp32 values: what the pool stores)
        a0 = [float(np.float32(v)) for v in a0]
        cases = [(list(s0), list(a0))]
        for bad in (np.nan, np.inf, -np.inf):
            for k in range(4):
                s = list(s0); s[k] = bad
                cases.append((s, list(a0)))
            for k in range(2):
                a = list(a0); a[k] = bad
                cases.append((list(s0), a))
        for s, a in cases:
            try:
                if mid == PM:
                    o = model.step(State(0, x=s[0], y=s[1], vx=s[2], vy=s[3]), (a[0], a[1]))
                    res, ap_ = [o.x, o.y, o.heading, o.speed, o.vx, o.vy], [a[0], a[1]]
                else:
                    o, aa, dd = model.step(State(0, x=s[0], y=s[1], heading=s[2], speed=s[3]), a[0], a[1])
                    res, ap_ = state_out(o), [aa, dd]
            except Exception as exc:   # noqa: BLE001  (recorded by its absence: the fixture holds what the reference RETURNS)
                raised += 1
                print(f"reference raised {type(exc).__name__} for model {mid} state {s} action {a}: {exc}")
                continue
            type_id.append(tid); model_id.append(mid); st.append(s); act.append(a); ivl.append(interval)
            out.append([np.nan if v is None else float(v) for v in res]); app.append([float(v) for v in ap_])
    np.savez_compressed(OUT, rows=np.array(rows), type_id=np.array(type_id, np.int32), model=np.array(model_id, np.int32),
                        state=np.array(st, np.float64), action=np.array(act, np.float64), interval=np.array(ivl, np.int32),
                        out=np.array(out, np.float64), applied=np.array(app, np.float64))
    print(f"{len(out)} cases -> {OUT} ({raised} inputs made the reference raise and are left out)")


if __name__ == "__main__":
    main()
