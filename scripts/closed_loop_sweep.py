#!/usr/bin/env python3
"""Closed loop at the metric size (4096 envs x 64 participants): us per step of ALL envs for G env groups x the three ways
of enqueuing (policy kernel -> t2d_step) per group and step (tactics2d_amd/csrc/t2d_loop.hip).

    GPU_MAX_HW_QUEUES=8 python scripts/closed_loop_sweep.py [steps] [out.json]

Prints one JSON object: {"single": {...}, "sweep": [{groups, launcher, us_per_step, us_per_step_20}, ...]}."""
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tactics2d_amd import scenarios as S  # noqa: E402
from tactics2d_amd.debug import ClosedLoop, env_groups as EnvGroups  # noqa: E402
from tactics2d_amd.pool import ParticipantPool  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
out_path = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
sc = S.mixed(4096, 64, seed=3)
res = dict(hw_queues=os.environ["GPU_MAX_HW_QUEUES"], steps=steps, sweep=[])


def ramp(fn, n=2500):
    fn(n)
    torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / n


# ---- one pool, one stream: t2d_step alone, and policy -> t2d_step from the host (the closed loop without groups)
pool = ParticipantPool(sc.n_env, sc.A)
sc.load(pool)
pool.set_auto_reset(True)
act = torch.zeros((sc.n, 2), dtype=torch.float32, device=dev)
rng = np.random.default_rng(0)
a0, a1 = sc.sample_actions(rng)
ta0, ta1 = torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)
st = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()


def single_open(n):
    pool.bind_actions(ta0.data_ptr(), ta1.data_ptr())
    for _ in range(n):
        pool.step(100, st.cuda_stream)


def single_closed(n):
    pool.bind_actions(act.data_ptr() + 4, act.data_ptr(), 2)
    f = pool._lib.t2d_debug_feedback_policy   # (pools of debug.env_groups live in libt2d_hip_debug.so)
    for _ in range(n):
        f(pool._h, act.data_ptr(), 12.0, 0.5, 0.04, st.cuda_stream)
        pool.step(100, st.cuda_stream)


ramp(single_open)
res["single"] = dict(open_loop_us=timed(single_open, steps), open_loop_us_20=timed(single_open, 20))
ramp(single_closed)
res["single"].update(closed_loop_us=timed(single_closed, steps), closed_loop_us_20=timed(single_closed, 20))
flags = pool.download(9)
res["single"]["flag_rates_closed"] = [float((flags & b).astype(bool).mean()) for b in (1, 2, 4, 8)]
pool.close()
print(json.dumps(res["single"]), flush=True)

for G in (1, 2, 4, 8, 16):
    eg = EnvGroups(sc, G)
    eg.configure(lambda p: p.set_auto_reset(True))
    for launcher in ("thread", "threads", "graph"):
        if launcher == "threads" and G == 1:
            continue
        try:
            loop = ClosedLoop(eg, launcher, 100, graph_steps=64)
            ramp(loop.run, 2560)
            us = min(timed(loop.run, steps) for _ in range(3))
            ramp(loop.run, 640)
            us64 = min(timed(loop.run, 64) for _ in range(3))
            if launcher == "graph":   # (the driver's 20-step region needs a graph of its own length)
                loop.close()
                loop = ClosedLoop(eg, launcher, 100, graph_steps=20)
                ramp(loop.run, 640)
            us20 = min(timed(loop.run, 20) for _ in range(5))
            loop.close()
            row = dict(groups=G, launcher=launcher, us_per_step=us, us_per_step_20=us20, us_per_step_64=us64)
        except Exception as e:  # noqa: BLE001
            row = dict(groups=G, launcher=launcher, error=str(e))
        res["sweep"].append(row)
        print(json.dumps(row), flush=True)
    eg.close()
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
