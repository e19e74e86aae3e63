#!/usr/bin/env python3
"""Golden vectors for PointMass(backend="euler") made by importing the reference physics (TEST INFRASTRUCTURE; build container
only).  The euler back-end is the reference's selectable second PointMass integrator (physics/point_mass.py:28, 177-207,
226-229): sub-steps of delta_t (+ a remainder), the clipped speed re-projected onto the PREVIOUS sub-step's heading.  Seeded
single steps over several range / timing rigs (inputs rounded to fp32 first: what the pool stores) -> tests/golden/pm_euler.npz:

    rows[n_types, 24] (model id 4 = T2D_MODEL_POINTMASS_EULER), type_id[n], state[n, 5] (x, y, heading, vx, vy), action[n, 2],
    timing[n, 2] (interval, delta_t), out[n, 6] (x, y, heading, speed, vx, vy: the reference's fp64)

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_pm_euler.py [--ref /root/reference]
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import f32, row_from_model  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden", "pm_euler.npz")
PM_EULER = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    from tactics2d.participant.trajectory import State
    from tactics2d.physics import PointMass

    rng = np.random.default_rng(20260)
    rigs = [((-7.0, 7.0), (-1.5, 1.5), 100, None),      # -> [0, 7]: the upper bound clips
            ((0.5, 1.2), None, 100, 5),                 # both bounds bite
            (None, None, 100, 5),                       # unconstrained
            (3.0, 2.0, 100, 5),                         # float r -> [0, r]
            ((0.2, 2.5), None, 9, 5),                   # one full sub-step + a 4 ms remainder
            ((0.0, 1.0), None, 50, 3),                  # 16 sub-steps + 2 ms
            ((0.8, 6.0), None, 3, 5)]                   # delta_t clamped to the interval
    rows, type_id, st, act, tim, out = [], [], [], [], [], []
    for tid, (sr, ar, interval, dt) in enumerate(rigs):
        m = PointMass(sr, ar, interval, dt, "euler")
        assert m.backend == "euler"
        rows.append(row_from_model(m, PM_EULER))
        n = 48
        x, y = f32(rng.uniform(-50, 50, n)), f32(rng.uniform(-50, 50, n))
        h = f32(rng.uniform(-np.pi, np.pi, n))
        sp = rng.uniform(0.0, 3.0, n) * (rng.random(n) > 0.1)            # a tenth at rest
        va = np.where(rng.random(n) < 0.5, h, rng.uniform(-np.pi, np.pi, n))   # half of them move along their heading
        vx, vy = f32(sp * np.cos(va)), f32(sp * np.sin(va))
        ax, ay = f32(rng.uniform(-3, 3, n)), f32(rng.uniform(-3, 3, n))
        for i in range(n):
            s = m.step(State(frame=0, x=float(x[i]), y=float(y[i]), heading=float(h[i]), vx=float(vx[i]), vy=float(vy[i])),
                       (float(ax[i]), float(ay[i])), interval)
            out.append([s.x, s.y, s.heading, s.speed, s.vx, s.vy])
        type_id += [tid] * n
        st.append(np.stack([x, y, h, vx, vy], 1)); act.append(np.stack([ax, ay], 1)); tim += [[interval, m.delta_t]] * n
    np.savez_compressed(OUT, rows=np.array(rows), type_id=np.array(type_id, np.int32), state=np.concatenate(st),
                        action=np.concatenate(act), timing=np.array(tim, np.int32), out=np.array(out))
    print("wrote", OUT, len(out), "cases")


if __name__ == "__main__":
    main()
