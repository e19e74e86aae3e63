"""Stand-alone `integrate_kernel` (kernel id 0) per physics model and pool size: µs per launch and the
HBM rate on SURVEY §8(d)'s 44-B per participant-step figure.  The north_star's roofline target is
quoted on THIS kernel; the sweep shows where each model sits once the pool fills the chip.

    python scripts/time_integrate.py [exact]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from tactics2d_amd import layout as L
from tactics2d_amd import scenarios as S
from tactics2d_amd.pool import ParticipantPool

variant = sys.argv[1] if len(sys.argv) > 1 else "fast"
ROWS, NAMES = S.full_type_table()


def rows_for(model):
    """(rows, type ids) of the type-table rows that use `model` (None = every row)."""
    m = ROWS[:, L.P_MODEL].astype(int)
    return ROWS, np.nonzero((m == model) if model is not None else (m < L.MODEL_DRIFT))[0]


MODELS = {"kinematics": L.MODEL_KINEMATICS, "dynamics": L.MODEL_DYNAMICS, "pointmass": L.MODEL_POINTMASS}
print(f"variant={variant}")
print(f"{'model':11s} {'participants':>12s} {'us/launch':>10s} {'part-steps/s':>13s} {'GB/s (44 B)':>12s} {'% of 8 TB/s':>11s}")
for label, model in list(MODELS.items()) + [("mixed", None)]:
    for n_env in [int(v) for v in os.environ.get('T2D_TI_ENVS', '4096,16384,65536').split(',')]:
        A = 64
        n = n_env * A
        rng = np.random.default_rng(1)
        rows, ids = rows_for(model)
        if ids.size == 0:
            continue
        tid = ids[rng.integers(0, ids.size, n)].astype(np.uint8)
        pool = ParticipantPool(n_env, A)
        pool.set_param_table(rows)
        pool.set_integrator_variant(variant)
        # speeds inside every type's range, moderate steering: the state a driven scene is in.  Every timed
        # launch starts from the same snapshot (a free run with constant random steering spins the dynamics
        # model up until its trig arguments leave the fast reduction range -- not a driving workload)
        pool.reset(np.float32(rng.uniform(-100, 100, n)), np.float32(rng.uniform(-100, 100, n)),
                   np.float32(rng.uniform(0, 6.28, n)), np.float32(rng.uniform(0.5, 1.4, n) if model == L.MODEL_POINTMASS
                                                                 else rng.uniform(2.0, 7.5, n)), tid)
        pool.set_actions(np.float32(rng.uniform(-1.0, 1.0, n)), np.float32(rng.uniform(-0.08, 0.08, n)))
        pool.snapshot()
        for _ in range(3):
            pool.integrate(100)
        pool.profile_enable(True)
        for _ in range(20):
            pool.restore()
            pool.integrate(100)
        ms, launches = pool.profile_read(0)
        us = 1e3 * ms / launches
        gbs = 44.0 * n / (us * 1e-6) / 1e9
        print(f"{label:11s} {n:12d} {us:10.2f} {n / (us * 1e-6):13.3e} {gbs:12.1f} {100 * gbs / 8000:10.1f}%")
        pool.close()
