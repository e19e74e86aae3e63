"""Why is ONE synchronised fragment of 20 steps, run the way bench.py's timed region runs it (the driver's `--steps 20 --warmup 5`),
slower than the same fragment repeated (scripts/launch_latency_probe.py: 16.7 us per step wall, 16.1 span)?

    python scripts/oneshot_probe.py

Each variant: [what precedes] -> synchronize -> t0, event, ONE t2d_step_n(20), event, synchronize, t1; 12 repetitions, median wall
and HIP-event span per step.  What precedes:
  steady        the same fragment, just before
  bench         3000 single steps of a SCRATCH pool holding the same scene (bench.py's untimed clock ramp), then 5 warm-up steps
                of the measured pool as one fragment -- bench.py's sequence
  bench_w20     the same with a 20-step warm-up fragment
  ramp_chained  the ramp on the scratch pool as chained fragments (the measured kernel's own code), then 5 warm-up steps
  idle_w5       50 ms of idle GPU, then 5 warm-up steps (no ramp)
  self_ramp     the ramp on the MEASURED pool itself (3000 of its own steps as fragments), then the timed fragment"""
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
run = B.Runner(scene, dev, "fast")
warm = B.Runner(scene, dev, "fast", seed=7)
FRAG = 20
pc = time.perf_counter


def one():
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(run.stream)
    t = pc()
    run.steps_chain(FRAG, FRAG)
    e1.record(run.stream)
    torch.cuda.synchronize()
    w = pc() - t
    return 1e6 * w / FRAG, 1e3 * e0.elapsed_time(e1) / FRAG


def pre_steady():
    run.steps_chain(FRAG, FRAG)


def pre_bench(w=5):
    warm.steps_single(3000)
    torch.cuda.synchronize()
    run.steps_chain(w, FRAG)


def pre_ramp_chained():
    warm.steps_chain(3000, FRAG)
    torch.cuda.synchronize()
    run.steps_chain(5, FRAG)


def pre_idle():
    torch.cuda.synchronize()
    time.sleep(0.05)
    run.steps_chain(5, FRAG)


def pre_self():
    run.steps_chain(3000, FRAG)


variants = [("steady", pre_steady), ("bench", pre_bench), ("bench_w20", lambda: pre_bench(20)), ("ramp_chained", pre_ramp_chained),
            ("idle_w5", pre_idle), ("self_ramp", pre_self)]
res = {}
run.steps_chain(600, FRAG)
for rep in range(12):
    for name, pre in variants:
        pre()
        res.setdefault(name, []).append(one())
out = {k: dict(wall_us_per_step=round(st.median(x[0] for x in v), 2), span_us_per_step=round(st.median(x[1] for x in v), 2),
               wall_min=round(min(x[0] for x in v), 2), wall_max=round(max(x[0] for x in v), 2)) for k, v in res.items()}
for k, v in out.items():
    print(k, v)
print("ONESHOT_PROBE", out)
run.close()
warm.close()
