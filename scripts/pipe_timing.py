"""Who waits for whom in the PIPE form of t2d_step_n (-DT2D_TIMING build: T2D_LIB_NAME=libt2d_hip_timing.so):
    python scripts/pipe_timing.py cfg3|cfg4|cfg5 [chaining mode: 1 = default, 4 = no lane waves]
cycles per step of every wave of a workgroup -- the event waves' phases incl. their wait for the integrator's commit, the lane
waves', and the integrator waves' integrate / wait for the verdict / integrate again after a reset / commit."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]

def pipe_report():
    """t2d_step_n on a small pool (PIPE form): cycles per step and role, accumulated over the launch (-DT2D_TIMING build)."""
    import torch
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    sc = {"cfg3": lambda: S.highway(1024, 64, seed=1), "cfg4": lambda: S.intersection(512, 32, seed=2), "cfg5": lambda: S.mixed(1024, 64, seed=3)}[name]()
    pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True); pool.set_step_chaining(mode)
    rng = np.random.default_rng(0)
    a0, a1 = sc.sample_actions(rng); pool.set_actions(a0, a1)
    n = 32
    for _ in range(40): pool.step_n(n, sc.interval_ms, 0)
    pool.sync()
    log2A = int(np.ceil(np.log2(sc.A))); epb = 256 >> log2A
    n_wg = (sc.n_env + epb - 1) // epb
    sets = 3 if (name != "cfg3" and mode == 1) else 2
    wpw = 4 * sets
    buf = np.zeros(n_wg * wpw * 16, np.uint64)
    lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    steps = 41 * n if False else 40 * n
    v = buf.reshape(n_wg, wpw, 16).astype(np.float64) / steps
    print(name, "chaining mode", mode, "form", pool.step_form(n), "sets of waves", sets, "| cycles per step, mean over workgroups")
    ev = ["0 stage/clear", "1 pose", "2 (b)", "3 broad", "4 pair narrow", "5 static box", "6 static narrow", "7 lane box", "8 lane narrow", "9 exit",
          "10 (c)+wait lanes", "11 reduce", "12 epilogue", "13 WAIT commit"]
    for w in range(4):
        print(f" event wave {w}: total {v[:, w, :14].sum(1).mean():7.0f} | " + " ".join(f"{ev[k].split()[0]}:{v[:, w, k].mean():.0f}" for k in range(14)))
    if sets == 3:
        for w in range(4, 8):
            print(f" lane wave  {w - 4}: total {v[:, w, :14].sum(1).mean():7.0f} | " + " ".join(f"{ev[k].split()[0]}:{v[:, w, k].mean():.0f}" for k in range(14)))
    for w in range(wpw - 4, wpw):
        x = v[:, w, :4].mean(0)
        print(f" integrator {w - (wpw - 4)}: integrate {x[0]:.0f}  WAIT verdict {x[1]:.0f}  again after reset {x[2]:.0f}  commit {x[3]:.0f}  | sum {x.sum():.0f}")
    pool.close()


pipe_report()
