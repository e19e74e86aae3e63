import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd.envs import VecParkingEnv
rng = np.random.default_rng(0)
for kw in (dict(lidar_beams=120), dict(lidar_beams=120, zero_copy=False), dict(lidar_beams=360), dict(lidar_beams=360, zero_copy=False), dict(info_lidar=False)):
    env = VecParkingEnv(4096, max_step=200, auto_reset=True, seed=1, **kw); env.reset()
    acts = [env.action_space.sample(rng, 4096) for _ in range(8)]
    for k in range(300): env.step(acts[k & 7])
    t = time.perf_counter()
    for k in range(2000): env.step(acts[k & 7])
    print(kw, round(1e6 * (time.perf_counter() - t) / 2000, 1), "us per step", flush=True)
    env.close()
