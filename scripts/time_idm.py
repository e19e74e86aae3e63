"""IDM kernel alone (t2d_idm_actions) on the metric scene: every non-pedestrian participant but the ego IDM-controlled."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tactics2d_amd import scenarios as S, layout as L
from tactics2d_amd.pool import ParticipantPool
from tactics2d_amd.controller import IDMController, install
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 1024]
short = "short" in sys.argv   # counter passes: a few dozen launches are enough
for n_env in sizes:
    sc = S.mixed(n_env, 64, 3)
    pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool)
    cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
    veh = (sc.rows[sc.type_id, L.P_MODEL] != L.MODEL_POINTMASS).reshape(sc.n_env, sc.A)
    cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
    install(pool, [IDMController(desired_speed=25.0, horizon=120.0)], cid.reshape(-1))
    for _ in range(60 if short else 3000): pool.idm_actions()        # clock ramp
    pool.sync()
    t = time.perf_counter(); n = 60 if short else 2000
    for _ in range(n): pool.idm_actions()
    pool.sync()
    print(f"{n_env} x 64: idm_kernel {1e6 * (time.perf_counter() - t) / n:.2f} us per launch (back to back, wall)")
    pool.close()
