"""VecParkingEnv.step_torch throughput (step kernel + lidar, actions generated on the device, no host sync)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tactics2d_amd.envs import VecParkingEnv
dev = torch.device("cuda", 0)
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [4096, 32768]
short = "short" in sys.argv   # counter passes: a few dozen steps are enough
sources = [a for a in sys.argv[1:] if a in ("layout", "generator")] or ["layout", "generator"]   # (one source: a per-source kernel trace)
for n, source in [(n_, s_) for n_ in sizes for s_ in sources]:
    # "layout": fixed bay layout, finished episodes restart from the snapshot; "generator": ParkingLotGenerator scenes
    # installed on the device, every finished episode continues in a new scene (staged ahead on the pool's stream)
    env = VecParkingEnv(n, max_step=200, auto_reset=True, seed=1, scene_source=source); env.reset()
    lo = torch.tensor([-0.524, -2.0], device=dev); hi = torch.tensor([0.524, 2.0], device=dev)
    acts = [lo + (hi - lo) * torch.rand((n, 2), device=dev) for _ in range(8)]
    for k in range(50): out = env.step_torch(acts[k & 7])
    torch.cuda.synchronize()
    t = time.perf_counter(); steps = 40 if short else 500
    for k in range(steps): out = env.step_torch(acts[k & 7])
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    print(f"{n} envs, scenes = {source}: {1e6 * el / steps:.1f} us per vector step, {n * steps / el:.3e} env-steps/s (state + 360-beam lidar on the device)")
    env.close()
