"""What the host's way of waiting costs a SHORT timed region (the driver's `bench.py --steps 20`: one chained fragment of 20 steps,
~330 us of GPU work, between two torch.cuda.synchronize()).

    python scripts/host_wait_probe.py

hipDeviceSynchronize spins on the completion signal for 100 us and then blocks on the interrupt (the runtime's default,
hipDeviceScheduleAuto); with hipSetDeviceFlags(hipDeviceScheduleSpin) it spins until the signal moves.  Measured here on the
metric scene: wall time per step of 20-step fragments, each bracketed by synchronize, under both settings, in alternation; beside
it the HIP-event span of the same fragments (what the GPU itself took)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import bench as B  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
scene = B.build_scene("metric", 4096, 64, seed=0)
run = B.Runner(scene, dev, "fast")
hip = C.CDLL("libamdhip64.so.7" if os.path.exists("/opt/rocm/lib/libamdhip64.so.7") else "libamdhip64.so")
hip.hipSetDeviceFlags.argtypes = [C.c_uint]
hip.hipGetDeviceFlags.argtypes = [C.POINTER(C.c_uint)]
AUTO, SPIN, BLOCKING = 0, 1, 4


def flags():
    f = C.c_uint(0)
    rc = hip.hipGetDeviceFlags(C.byref(f))
    return rc, f.value


def measure(frag=20, n=200):
    run.steps_chain(600, 20)       # (fragments no longer than the action ring: bench.ACTION_SETS)
    torch.cuda.synchronize()
    wall = span = 0.0
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        e0.record(run.stream)
        run.steps_chain(frag, frag)
        e1.record(run.stream)
        torch.cuda.synchronize()
        wall += time.perf_counter() - t
        span += e0.elapsed_time(e1) * 1e-3
    return 1e6 * wall / (n * frag), 1e6 * span / (n * frag)


res = {}
print("flags at start", flags())
for rep in range(3):
    for name, fl in (("auto", AUTO), ("spin", SPIN), ("blocking", BLOCKING)):
        rc = hip.hipSetDeviceFlags(fl)
        w, s = measure()
        res.setdefault(name, []).append((round(w, 3), round(s, 3)))
        print(name, "rc", rc, "flags", flags(), "wall us/step %.3f  event span us/step %.3f" % (w, s), flush=True)
print("HOST_WAIT_PROBE", res)
run.close()
