"""Where do the workgroups of the step launch land?  (-DT2D_TIMING build: every wave records HW_ID / XCC_ID.)
Prints, for several consecutive launches, whether the blockIdx -> (XCD, SE, CU) and wave -> SIMD assignment repeats, and
which workgroups share a CU."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
sc = S.mixed(4096, 64, 3)
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
rng = np.random.default_rng(0)
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
n_waves = 4096
keys = []
for k in range(6):
    a0, a1 = sc.sample_actions(rng); pool.set_actions(a0, a1)
    for _ in range(3 if k % 2 else 1): pool.step(100)      # odd rounds: three launches back to back, read the last
    buf = np.zeros(n_waves * 16, np.uint64)
    lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
    raw = buf.reshape(n_waves, 16)
    hw = raw[:, 14] & 0xffffffff; xcc = (raw[:, 14] >> 32) & 0xf
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    keys.append((cuid.copy(), simd.copy(), raw[:, 15].copy(), raw[:, :14].sum(1).astype(np.float64)))
base = keys[0]
for k in range(1, 6):
    print("launch", k, "same CU as launch 0:", (keys[k][0] == base[0]).mean(), " same SIMD:", (keys[k][1] == base[1]).mean())
cuid, simd = base[0], base[1]
wg_cu = cuid.reshape(1024, 4)
print("waves of a workgroup on one CU:", (wg_cu == wg_cu[:, :1]).all(), " wave -> simd patterns:", np.unique(simd.reshape(1024, 4), axis=0)[:6].tolist())
groups = {}
for b in range(1024): groups.setdefault(int(wg_cu[b, 0]), []).append(b)
sizes = np.bincount([len(v) for v in groups.values()])
print("CUs used", len(groups), "workgroups per CU histogram", sizes.tolist())
for c in list(groups)[:6]: print(" CU", c, "workgroups", groups[c], "differences", np.diff(groups[c]).tolist())
xs = np.array([b % 8 for b in range(1024)]); print("blockIdx % 8 == XCD:", (xs == (cuid.reshape(1024, 4)[:, 0] // 256)).mean())
# dump of the last launch for offline analysis (per wave: 14 phase tick counts, HW_ID word, start tick)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/placement_raw.npy", raw)
