"""Per-wave cost without contention (-DT2D_TIMING build): the metric scene stepped once as one pool of 4096 envs (four waves
per SIMD: the contended finish times) and as four pools of 1024 envs one after the other (one wave per SIMD: what each wave
costs on its own).  Dumps both for offline analysis (is a SIMD's finish time predictable from its waves' solo costs, and what
would a cost-aware env -> SIMD map buy)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
sc = S.mixed(4096, 64, 3)
rng = np.random.default_rng(0)
a0, a1 = sc.sample_actions(rng)
def run(scene, act0, act1, n_warm=3):
    pool = ParticipantPool(scene.n_env, scene.A); scene.load(pool); pool.set_auto_reset(True)
    pool.set_actions(act0, act1)
    out = []
    for k in range(n_warm):        # the same state every time: restore the snapshot before each step
        pool.restore()
        pool.step(100); pool.sync()
        buf = np.zeros(scene.n_env * 16, np.uint64)
        lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
        out.append(buf.reshape(scene.n_env, 16).copy())
    pool.close()
    return out[-1]
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/solo_full.npy", run(sc, a0, a1))
parts = []
for g in range(4):
    lo, hi = g * 1024, (g + 1) * 1024
    parts.append(run(sc.shard(lo, hi), a0[lo * 64:hi * 64], a1[lo * 64:hi * 64]))
np.save("gpurun_out/solo_parts.npy", np.concatenate(parts))
print("done")
