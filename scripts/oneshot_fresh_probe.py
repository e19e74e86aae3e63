"""bench.py's timed region on a FRESH pool -- the way the driver's single run sees it -- against the same on a pool that has run
before (scripts/oneshot_probe.py: 16.65 us per step with bench.py's own sequence, where the driver-like line says 17.7-19.1).

    python scripts/oneshot_fresh_probe.py

Every repetition builds a new Runner (pool + action ring), then: clock ramp on a scratch pool, W warm-up steps as one fragment,
synchronize, ONE timed fragment of 20 -- and a SECOND timed fragment of 20 right behind it.  Variants say what else is done to the
fresh pool before the ramp:
  plain          nothing (bench.py's sequence)
  touch_actions  the whole action ring read once by a torch kernel (sum): its pages mapped / in the MALL
  prior_frag20   one untimed fragment of 20 on the measured pool before the ramp (NOT what --warmup 5 allows: a diagnostic)"""
import os
import statistics as st
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
scene = B.build_scene("metric", 4096, 64, seed=0)
warm = B.Runner(scene, dev, "fast", seed=7)
FRAG = 20
pc = time.perf_counter


def one(run):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(run.stream)
    t = pc()
    run.steps_chain(FRAG, FRAG)
    e1.record(run.stream)
    torch.cuda.synchronize()
    w = pc() - t
    return 1e6 * w / FRAG, 1e3 * e0.elapsed_time(e1) / FRAG


res = {}
for rep in range(6):
    for name in ("plain", "touch_actions", "prior_frag20", "warmup20"):
        run = B.Runner(scene, dev, "fast", seed=100 + rep)
        if name == "touch_actions":
            _ = float((run.a0.sum() + run.a1.sum()).item())
        if name == "prior_frag20":
            run.steps_chain(FRAG, FRAG)
            torch.cuda.synchronize()
        warm.steps_single(3000)
        torch.cuda.synchronize()
        run.steps_chain(20 if name == "warmup20" else 5, FRAG)
        first = one(run)
        second = one(run)
        res.setdefault(name, []).append((first, second))
        run.close()
out = {}
for k, v in res.items():
    out[k] = dict(first_wall=round(st.median(x[0][0] for x in v), 2), first_span=round(st.median(x[0][1] for x in v), 2),
                  second_wall=round(st.median(x[1][0] for x in v), 2), second_span=round(st.median(x[1][1] for x in v), 2))
    print(k, out[k])
print("per repetition (first fragment: wall, span)", {k: [(round(x[0][0], 2), round(x[0][1], 2)) for x in v] for k, v in res.items()})
print("ONESHOT_FRESH_PROBE", out)
warm.close()
