"""Per-phase cycle breakdown of the lidar kernel (-DT2D_TIMING build)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
sc = S.parking(4096)
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.lidar_config(360, 20.0, False)
for _ in range(5): pool.lidar_scan()
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
n_waves = sc.n_env * 2
buf = np.zeros(n_waves * 16, np.uint64)
pool.sync(); lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
v = buf.reshape(n_waves, 16).astype(np.float64)
names = ["0 ego transform + barrier", "1 edges + spans + barrier", "2 mask clear + scatter + barriers", "3 compaction + evaluation", "4 output"]
tot = v[:, :5].sum(1)
print("mean ticks per wave", tot.mean(), "max", tot.max())
for k, n in enumerate(names): print(f"  {n:36s} {v[:, k].mean():9.1f}  {100 * v[:, k].mean() / tot.mean():5.1f}%")
