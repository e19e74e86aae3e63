"""Feasibility probe for the ParkingEnv vector step: what would the step cost if the ego step kernel and the
lidar scan of the SAME step ran side by side?

    GPU_MAX_HW_QUEUES=8 python scripts/overlap_probe.py

The scan needs the pose the ego step produces, so the shipped order is ego_step_kernel -> lidar_kernel on one stream (27.5 us at
4096 envs).  An overlap would need the scan to take the new pose from a word the ego kernel publishes after its integrator.  Before
building that: this launches the two kernels of a step on TWO streams with no dependency inside the step (the scan reads whatever
pose is there -- its RESULTS mean nothing here) and a two-way event join between steps, and times the pair.  That is the best
case of the overlap: if the pair is not clearly faster than the sum, the hand-off cannot be either."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import bench as B  # noqa: E402
from tactics2d_amd import scenarios as S  # noqa: E402

dev = torch.device("cuda", 0)
sc = S.parking(4096)
r = B.Runner(sc, dev, "fast")
r.pool.lidar_config(360, 20.0, False)
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
r.pool.bind_actions(r.a0.data_ptr(), r.a1.data_ptr())


def timed(fn, n=400, warm=100, reps=3):
    best = 1e9
    for _ in range(reps):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t) / n)
    return best


def serial():
    r.pool.step(sc.interval_ms, sa.cuda_stream)
    r.pool.lidar_scan(None, sa.cuda_stream)


def ego_only():
    r.pool.step(sc.interval_ms, sa.cuda_stream)


def scan_only():
    r.pool.lidar_scan(None, sa.cuda_stream)


ev_a = [torch.cuda.Event() for _ in range(2)]
ev_b = [torch.cuda.Event() for _ in range(2)]
k = [0]


def side_by_side(ego_first=True):
    i = k[0] & 1
    k[0] += 1
    if ego_first:
        r.pool.step(sc.interval_ms, sa.cuda_stream)
        r.pool.lidar_scan(None, sb.cuda_stream)
    else:
        r.pool.lidar_scan(None, sb.cuda_stream)
        r.pool.step(sc.interval_ms, sa.cuda_stream)
    ev_a[i].record(sa)
    ev_b[i].record(sb)
    sa.wait_event(ev_b[i])   # the next step's kernels start when BOTH of this step's have ended
    sb.wait_event(ev_a[i])


res = dict(form=r.pool.step_form(1))
res["ego_step_us"] = timed(ego_only)
res["scan_us"] = timed(scan_only)
res["serial_one_stream_us"] = timed(serial)
res["two_streams_ego_first_us"] = timed(lambda: side_by_side(True))
res["two_streams_scan_first_us"] = timed(lambda: side_by_side(False))
# the join's own cost: two empty-handed streams joining each step (two events + two waits around the ego step alone)


def join_only():
    i = k[0] & 1
    k[0] += 1
    r.pool.step(sc.interval_ms, sa.cuda_stream)
    ev_a[i].record(sa)
    ev_b[i].record(sb)
    sa.wait_event(ev_b[i])
    sb.wait_event(ev_a[i])


res["ego_step_plus_join_us"] = timed(join_only)
print("OVERLAP_PROBE", res)
r.close()
