"""Per-phase cycle breakdown of the collide kernel (profiling build with -DT2D_TIMING)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import _ffi, scenarios as S
from tactics2d_amd.pool import ParticipantPool
cfg = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = {"metric": lambda: S.mixed(4096, 64, 3), "cfg3": lambda: S.highway(1024, 64), "cfg2": lambda: S.parking(4096),
      "cfg5": lambda: S.mixed(1024, 64, 3), "cfg5nosplit": lambda: S.mixed(1024, 64, 3), "cfg3split": lambda: S.highway(1024, 64)}[cfg]()
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool)
if cfg == "cfg5nosplit":
    pool.set_split_step(False)
split = pool.step_form(1) == "step_split"
rng = np.random.default_rng(0)
lib = _ffi.lib(); lib.t2d_debug_read.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
log2A = int(np.ceil(np.log2(sc.A))); epb = 256 >> log2A
n_blocks = sc.n_env if split else (sc.n_env + epb - 1) // epb
n_waves = n_blocks * 4
buf = np.zeros(n_waves * 16, np.uint64)
for k in range(60):
    a0, a1 = sc.sample_actions(rng); pool.set_actions(a0, a1); pool.step(100); pool.restore(done_only=True)
lib.t2d_debug_read(pool._h, buf.ctypes.data_as(C.c_void_p), buf.size)
raw = buf.reshape(n_waves, 16)
v = raw.astype(np.float64)
names = ["0 load+stage+barrier(a)", "1 pose", "2 barrier(b)", "3 broad phase", "4 pair compaction+narrow", "5 static AABB pass",
         "6 static narrow", "7 lane AABB pass", "8 lane narrow", "9 (loop exit)", "10 barrier(c)", "11 reduce+barrier(d)", "12 epilogue",
         "13 fused integrate"]
tot = v[:, :14].sum(1)
print(cfg, "split" if split else "", "waves", n_waves, "mean ticks/wave", tot.mean(), "max", tot.max())
if split:   # per role (wave of the env's workgroup) and env kind
    for t in range(3):
        for role in range(4):
            sel = ((np.arange(n_waves) // 4) % 3 == t) & (np.arange(n_waves) % 4 == role)
            print(f" env kind {t} role {role}: total {tot[sel].mean():8.0f} | " + " ".join(f"{k}:{v[sel, k].mean():.0f}" for k in range(14)))
for t in range(3 if cfg == "metric" else 1):
    sel = (np.arange(n_waves) % 3 == t) if cfg == "metric" else np.ones(n_waves, bool)
    print(" env type", t, "mean total", tot[sel].mean())
    for k, n in enumerate(names): print(f"  {n:32s} {v[sel, k].mean():10.1f}  {100 * v[sel, k].mean() / tot[sel].mean():5.1f}%")
q = np.quantile(tot, [0.5, 0.9, 0.99, 1.0])
print("quantiles of wave total", q)
slow = tot >= np.quantile(tot, 0.97)
print("slowest 3% waves: env types", np.bincount(np.arange(n_waves)[slow] % 3, minlength=3), "mean total", tot[slow].mean())
for k, n in enumerate(names): print(f"  {n:32s} {v[slow, k].mean():10.1f}   (all waves {v[:, k].mean():10.1f})")

# placement: per-SIMD sums (HW_ID bits: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC_ID low bits)
hw = raw[:, 14] & 0xffffffff; xcc = (raw[:, 14] >> 32) & 0xf
simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
uk, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
per = np.bincount(inv, weights=tot)
start = raw[:, 15].astype(np.float64); end = start + tot
print("SIMDs used", len(uk), "waves/SIMD min/mean/max", cnt.min(), cnt.mean(), cnt.max())
print("XCCs", np.unique(xcc), "SEs", np.unique(se), "CUs", np.unique(cu), "SH", np.unique(sh))
for x in np.unique(xcc):
    m = xcc == x
    print(f" xcc {x}: waves {m.sum()}, span {end[m].max() - start[m].min():.0f} ticks, first start {start[m].min():.0f}, last start {start[m].max() - start[m].min():.0f} after, mean wave {tot[m].mean():.0f}")
span = np.array([end[inv == i].max() - start[inv == i].min() for i in range(len(uk))])
print("per-SIMD span (first start -> last end): mean", span.mean(), "max", span.max(), "min", span.min())
print("per-SIMD sum of wave times: mean", per.mean(), "max", per.max())
wg = np.arange(n_waves) // 4
print("first 16 WGs -> (xcc,se,sh,cu):", [(int(xcc[4*i]), int(se[4*i]), int(sh[4*i]), int(cu[4*i])) for i in range(16)])
print("WG 0 waves simd:", simd[:4], " WG 1:", simd[4:8])
