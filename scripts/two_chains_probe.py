"""Do two CHAINED launches side by side beat one?  The metric scene (4096 x 64) as ONE pool stepped by t2d_step_n fragments, against
the same envs cut into G env groups (contiguous blocks), each a pool of its own, its fragments on a stream of its own.

    GPU_MAX_HW_QUEUES=8 python scripts/two_chains_probe.py

A wave slot of the chained form idles ~1.2 us per step between two workgroups (dispatch) and the fragment's last step drains over
one workgroup life; launches from several hardware queues could fill those holes -- or not.  Timed: wall time
per step of all 4096 envs over back-to-back fragments, and of single synchronised fragments of 20 (the driver's shape)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
scene = B.build_scene("metric", 4096, 64, seed=0)
FRAG = 20


def make(groups):
    per = scene.n_env // groups
    runs = []
    for g in range(groups):
        sc = scene if groups == 1 else scene.shard(g * per, (g + 1) * per)
        r = B.Runner(sc, dev, "fast", seed=5 + g)
        r.pool.set_step_chaining(2)       # the chained form whatever the pool's size
        runs.append(r)
    return runs


def fragment(runs):
    for r in runs:
        r.steps_chain(FRAG, FRAG)


def timed(runs, synced, n=150, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(30):
            fragment(runs)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fragment(runs)
            if synced:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t) / (n * FRAG))
    return best


res = {}
for groups in (1, 2, 4):
    runs = make(groups)
    forms = [r.pool.step_form(FRAG) for r in runs]
    res[groups] = dict(form=forms[0], back_to_back_us=round(timed(runs, False), 3), synced_each_us=round(timed(runs, True), 3))
    print(groups, res[groups], flush=True)
    for r in runs:
        r.close()
print("TWO_CHAINS_PROBE", res)
