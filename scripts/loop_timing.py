"""Per-phase cycles of the LOOP form of t2d_step_n (a workgroup walks through the steps itself), wave by wave (-DT2D_TIMING build:
the stamps accumulate over every step since the pool was created).

    T2D_LIB_NAME=libt2d_loop4_timing.so python scripts/loop_timing.py [frag] [fragments]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
frag = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n_frag = int(sys.argv[2]) if len(sys.argv) > 2 else 10
import torch  # noqa: E402
import bench as B  # noqa: E402
from tactics2d_amd import _ffi  # noqa: E402

dev = torch.device("cuda", 0)
n_env = int(os.environ.get("T2D_CT_ENVS", 4096))
scene = B.build_scene("metric", n_env, 64, seed=0)
warm = B.Runner(scene, dev, "fast")
warm.steps_chain(400, frag)          # clocks up on a scratch pool: the measured pool's stamps hold its own steps only
torch.cuda.synchronize()
run = B.Runner(scene, dev, "fast")
form = run.pool.step_form(frag)
lib = _ffi.lib()
lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(run.stream)
run.steps_chain(n_frag * frag, frag)
e1.record(run.stream)
torch.cuda.synchronize()
us = 1e3 * e0.elapsed_time(e1) / (n_frag * frag)
n_waves = n_env
buf = np.zeros(n_waves * 16, np.uint64)
assert lib.t2d_debug_read(run.pool._h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
v = buf.reshape(n_waves, 16).astype(np.float64) / (n_frag * frag)
names = ["0 start-up / trip head", "1 pose", "2 sync (b)", "3 pair broad phase", "4 pair compaction + SAT", "5 static box sweep",
         "6 static SAT", "7 lane box sweep", "8 lane narrow", "9 off-lane stage 2", "10 sync (c)", "11 reduce + sync (d)", "12 epilogue",
         "13 fused integrator"]
tot = v[:, :14].sum(1)
res = dict(form=form, us_per_step_events=us, cycles_per_wave_and_step=float(tot.mean()),
           phases={n: float(v[:, k].mean()) for k, n in enumerate(names)},
           by_kind={nm: dict(total=float(tot[np.arange(n_waves) % 3 == t].mean()),
                             phases={n: float(v[np.arange(n_waves) % 3 == t, k].mean()) for k, n in enumerate(names)})
                    for t, nm in enumerate(("highway", "roundabout", "intersection"))})
print(json.dumps(res, indent=1))
