"""Which path would the kinematic lanes of the metric scene take?  Runs the scene for a while, downloads state + actions and
evaluates the step's path conditions (t2d_integrate_dev.h step_kinematics, fast variant) per lane and per wave in numpy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tactics2d_amd import layout as L, scenarios as S
from tactics2d_amd.pool import ParticipantPool
name = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = {"metric": lambda: S.mixed(4096, 64, seed=3), "cfg4": lambda: S.intersection(512, 32, seed=2), "cfg2": lambda: S.parking(4096)}[name]()
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
rng = np.random.default_rng(5)
for k in range(int(os.environ.get("STEPS", "150"))):
    a0, a1 = sc.sample_actions(rng); pool.set_actions(a0, a1); pool.step(sc.interval_ms)
a0, a1 = sc.sample_actions(rng)
v = pool.download(L.F_SPEED).astype(np.float64); ids = pool.download(L.F_IDS)
rows = sc.rows
model = ids & 0xff; tid = (ids >> 8) & 0xff; active = (ids >> 16) & 0xff
r = rows[tid]
kin = (model == 0) & (active != 0)
fl = r[:, L.P_RANGE_FLAGS].astype(int)
acc = np.where(fl & 4, np.clip(a0, r[:, L.P_ACCEL_LO], r[:, L.P_ACCEL_HI]), a0).astype(np.float64)
dl = np.where(fl & 1, np.clip(a1, r[:, L.P_STEER_LO], r[:, L.P_STEER_HI]), a1).astype(np.float64)
clip_v = (fl & 2) != 0
vlo, vhi = r[:, L.P_SPEED_LO], r[:, L.P_SPEED_HI]
dt = r[:, L.P_DELTA_T_MS] / 1000; n = (sc.interval_ms // r[:, L.P_DELTA_T_MS]).astype(int)
ah = acc * dt
t = r[:, L.P_LR] / r[:, L.P_WB] * np.tan(dl); cb = 1 / np.sqrt(1 + t * t); kh = np.tan(dl) / r[:, L.P_WB] * cb * dt
pinned = clip_v & (((v == vhi) & (ah >= 0)) | ((v == vlo) & (ah <= 0)))
ah_l = np.where(pinned, 0.0, ah)
v_end = v + n * ah_l
eps0, eps_end, dlt = v * kh, v_end * kh, ah_l * kh
linear = (~clip_v | ((v >= vlo) & (v <= vhi) & (v_end >= vlo) & (v_end <= vhi))) & (np.abs(eps0) <= 0.1) & (np.abs(eps_end) <= 0.1)
m, M = (n - 1) / 2, n / 2
a = (eps0 + dlt * (m - 0.5)) * M; b = dlt * M * M / 2
resum = linear & (np.abs(a) <= 0.5) & (np.abs(b) <= 5e-3)
A = sc.A
W = 64 // A if A < 64 else 1
def per_wave(mask):   # a wave = 64 consecutive slots
    k = kin.reshape(-1, 64); mm = (mask | ~kin).reshape(-1, 64)
    has = k.any(1)
    return has.sum(), (mm.all(1) & has).sum()
print(name, "kinematic lanes", kin.sum(), "linear", (linear & kin).sum(), "resum", (resum & kin).sum(), "pinned", (pinned & kin).sum())
hw, lw = per_wave(linear); _, rw = per_wave(resum)
print("waves with kinematic lanes", hw, "all linear", lw, "all resummable", rw)
bad = kin & ~linear
print("non-linear lanes by type:", {int(t_): int(((tid == t_) & bad).sum()) for t_ in np.unique(tid[bad])})
print("speed ranges of those types:", {int(t_): (rows[t_, L.P_SPEED_LO], rows[t_, L.P_SPEED_HI]) for t_ in np.unique(tid[bad])})
pool.close()
