import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, ctypes as C
from tactics2d_amd.envs import ParkingEnv
for kw in (dict(), dict(info_lidar=False)):
    env = ParkingEnv(seed=0, **kw); env.reset()
    rng = np.random.default_rng(0)
    acts = [env.action_space.sample(rng) * 0.2 for _ in range(64)]
    for k in range(300): env.step(acts[k & 63])
    pool = env.scenario_manager.pool
    a = env._abuf; box = env._vec._action_box
    lib = pool._lib; h = pool._h; fp = pool._frame_ptr
    ap, bp = a.ctypes.data, box.ctypes.data
    N = 3000
    t = time.perf_counter()
    for k in range(N): lib.t2d_step_host(h, ap, bp, 100, None, 0, C.byref(fp))
    tc = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for k in range(N): pool.step_host(a, 100, action_box=box)
    tp = (time.perf_counter() - t) / N
    t = time.perf_counter()
    for k in range(N):
        o, r, te, tr, info = env.step(acts[k & 63])
        if te or tr: env.reset()
    te_ = (time.perf_counter() - t) / N
    print(kw, "raw C call %.1f us, pool.step_host %.1f us, ParkingEnv.step %.1f us" % (1e6*tc, 1e6*tp, 1e6*te_))
    env.close()
