"""A short steady-state run of the metric scene for counter passes (scripts/sq_cmd.sh TAG python scripts/count_step.py [chain]):
300 steps with auto-reset from the snapshot, device-resident action ring; `chain` = as t2d_step_n fragments of 20."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd import scenarios as S
from tactics2d_amd.pool import ParticipantPool
dev = torch.device("cuda", 0)
name = os.environ.get("T2D_COUNT_CONFIG", "metric")
def _kind(k):   # 4096 envs of ONE of the metric scene's three env kinds (0 highway, 1 roundabout, 2 intersection + pedestrians)
    fn = [lambda e, rng, A, by: S._highway_env(rng, A, by, True), lambda e, rng, A, by: S._roundabout_env(rng, A, by),
          lambda e, rng, A, by: S._intersection_env(rng, A, by, 0.10)][k]
    return S._assemble("mixed", 4096, 64, 3, fn)
sc = {"hw": lambda: _kind(0), "rb": lambda: _kind(1), "ix": lambda: _kind(2), "metric": lambda: S.mixed(4096, 64, seed=3), "cfg3": lambda: S.highway(1024, 64, seed=1),
      "cfg4": lambda: S.intersection(512, 32, seed=2), "cfg5": lambda: S.mixed(1024, 64, seed=3)}[name]()
rng = np.random.default_rng(5)
sets = [sc.sample_actions(rng) for _ in range(4)]
a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
if "state_only" in sys.argv:
    pool.set_outputs(velocity=False, applied=False)
n = int(os.environ.get("T2D_COUNT_STEPS", "300"))
if "chain" in sys.argv:
    pool.bind_actions(a0.data_ptr(), a1.data_ptr())
    for _ in range(n // 20):
        pool.step_n(20, sc.interval_ms, 0)
else:
    for k in range(n):
        pool.bind_actions(a0.data_ptr() + 4 * sc.n * (k & 3), a1.data_ptr() + 4 * sc.n * (k & 3))
        pool.step(sc.interval_ms)
pool.sync()
pool.close()
