"""ego_step_kernel at 4096 envs with action sets that steer the kinematic lanes into one integrator path: random actions (7 % of
the egos reach a speed bound inside the step: their waves take the piecewise loop), zero acceleration (every lane linear), and
accelerations that keep every ego pinned on a bound."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd import scenarios as S
from tactics2d_amd.pool import ParticipantPool
dev = torch.device("cuda", 0)
sc = S.parking(4096)
rng = np.random.default_rng(5)
for label in ("random", "accel=0", "accel=+2 (pinned at the upper bound after a few steps)", "accel small (+-0.02)"):
    pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
    a0, a1 = sc.sample_actions(rng)
    if label == "accel=0": a0[:] = 0
    elif label.startswith("accel=+2"): a0[:] = 2.0
    elif label.startswith("accel small"): a0[:] = rng.uniform(-0.02, 0.02, a0.shape)
    t0 = torch.from_numpy(a0).to(dev); t1 = torch.from_numpy(a1).to(dev)
    pool.bind_actions(t0.data_ptr(), t1.data_ptr())
    st = torch.cuda.Stream(device=dev)
    for _ in range(600): pool.step(100, st.cuda_stream)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        for _ in range(2000): pool.step(100, st.cuda_stream)
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t) / 2000)
    print(f"{label:60s} {best:6.2f} us per launch")
    pool.close()
