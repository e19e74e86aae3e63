import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
sc = S.mixed(4096, 64, 3)
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
rng = np.random.default_rng(0)
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
for k in range(300):
    a0, a1 = sc.sample_actions(rng) if k % 50 == 0 else (a0, a1); pool.set_actions(a0, a1) if k % 50 == 0 else None; pool.step(100)
pool.sync()
buf = np.zeros(2 * 65536, np.uint64)
lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
a = buf[65536:].reshape(4096, 16).astype(np.float64)
print("startup split (ticks per wave): kernargs %.0f | issue state loads %.0f | issue rest %.0f | wait all loads %.0f | LDS stores %.0f | barrier %.0f" % tuple(a[:, k].mean() for k in (5, 0, 1, 2, 3, 4)))
print("p10/p50/p90 of wait:", np.percentile(a[:, 2], [10, 50, 90]), " of first:", np.percentile(a[:, 0], [10, 50, 90]))
