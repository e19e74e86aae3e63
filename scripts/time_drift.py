"""Drift kernel timing (kernel id 5) at two pool sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import helpers as H
from tactics2d_amd import layout as L
from tactics2d_amd.pool import ParticipantPool
g = H.load_npz("drift.npz")
for n_env, A in ((1024, 64), (4096, 64)):
    n = n_env * A
    rng = np.random.default_rng(0)
    pool = ParticipantPool(n_env, A); pool.set_param_table(g["rows"][:1])
    v = np.float32(rng.uniform(5, 25, n))
    pool.reset(rng.uniform(-100, 100, n), rng.uniform(-100, 100, n), rng.uniform(0, 6.28, n), v, np.zeros(n, np.uint8))
    pool.upload(L.F_OMEGA_F, v / 0.344); pool.upload(L.F_OMEGA_R, v / 0.344)
    pool.set_actions(np.float32(rng.uniform(-2, 2, n)), np.float32(rng.normal(0, 0.05, n)))
    pool.profile_enable(True)
    for k in range(20): pool.integrate(100)
    ms, l = pool.profile_read(5)
    print(n, "drift participants: kernel avg us", 1e3 * ms / l, "-> participant-steps/s %.3e" % (n / (ms / l * 1e-3)))
    pool.close()
