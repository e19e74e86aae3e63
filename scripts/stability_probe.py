"""Are the per-wave / per-SIMD costs of the step launch stable from one step to the next?  (-DT2D_TIMING build.)
Dumps the per-wave phase ticks of several consecutive steps to gpurun_out/stab_<k>.npy for offline analysis."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
sc = S.mixed(4096, 64, 3)
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
rng = np.random.default_rng(0)
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
os.makedirs("gpurun_out", exist_ok=True)
for k in range(40):
    a0, a1 = sc.sample_actions(rng); pool.set_actions(a0, a1); pool.step(100)
    if k >= 30:
        pool.sync()
        buf = np.zeros(4096 * 16, np.uint64)
        lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
        np.save(f"gpurun_out/stab_{k - 30}.npy", buf.reshape(4096, 16))
print("done")
