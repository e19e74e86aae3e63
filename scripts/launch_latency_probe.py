"""Where the host-side microseconds of ONE synchronised fragment go (the driver's `bench.py --steps 20`: t0, one t2d_step_n call,
synchronize, t1).

    python scripts/launch_latency_probe.py

Per piece, median of 200: a trivial library call through ctypes (t2d_step_count), the t2d_step_n call itself with the GPU idle
(its host time: validation, record slots, the launch), the time from the call's return to the end of synchronize, an empty
synchronize, and the HIP-event span of the fragment.  wall = call + wait; GPU span + launch-to-start + completion-to-wake = wait."""
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
pool, s = run.pool, run.stream.cuda_stream
FRAG = 20
run.steps_chain(600, FRAG)
torch.cuda.synchronize()
pc = time.perf_counter
t_triv, t_call, t_wait, t_sync0, t_span, t_bind = [], [], [], [], [], []
for _ in range(200):
    torch.cuda.synchronize()
    a = pc(); pool.step_count(); b = pc()
    t_triv.append(b - a)
    a = pc(); torch.cuda.synchronize(); b = pc()
    t_sync0.append(b - a)
    a = pc(); pool.bind_actions(run.a0.data_ptr(), run.a1.data_ptr(), extent=run.a0.numel()); b = pc()
    t_bind.append(b - a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(run.stream)
    a = pc()
    pool.step_n(FRAG, scene.interval_ms, scene.n, s)
    b = pc()
    e1.record(run.stream)
    torch.cuda.synchronize()
    c = pc()
    t_call.append(b - a)
    t_wait.append(c - b)
    t_span.append(e0.elapsed_time(e1) * 1e-3)
med = lambda v: 1e6 * st.median(v)
res = dict(trivial_ctypes_call_us=med(t_triv), empty_synchronize_us=med(t_sync0), bind_actions_with_extent_us=med(t_bind),
           step_n_host_call_us=med(t_call), return_to_end_of_synchronize_us=med(t_wait), hip_event_span_us=med(t_span),
           wall_us=med(t_call) + med(t_wait), wall_us_per_step=(med(t_call) + med(t_wait)) / FRAG, span_us_per_step=med(t_span) / FRAG)
print("LAUNCH_LATENCY_PROBE", {k: round(v, 2) for k, v in res.items()})
run.close()
