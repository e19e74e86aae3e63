"""Randomised soak of the HBM grid tier (tactics2d_amd/csrc/t2d_mapgrid.hip) against the oracle: roads of random length, curvature,
lane count and sampling density, random clutter beside and on them, vehicles of random size scattered along -- flags and env flags
bit-exact, seed after seed.  (tests/test_gpu_mapgrid.py runs a fixed handful of such scenes.)
Usage on the GPU box: python tests/soak/soak_mapgrid.py [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_mapgrid as T
from oracle import oracle as O
from tactics2d_amd import layout as L, mapgeom as MG
from tactics2d_amd.pool import ParticipantPool
O.build(); O.set_threads(min(16, os.cpu_count() or 1))


n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t0 = time.time(); total = 0; bad_total = 0; grid = 0; seen = np.zeros(3, np.int64)
for seed in range(n_seeds):
    rng = np.random.default_rng(424242 + seed)
    n_env, A = int(rng.integers(2, 6)), int(rng.choice([16, 48, 64, 100]))
    n_pts = int(rng.integers(120, 320)); lanes_n = int(rng.integers(2, 6))
    rails = T._road(n_pts=n_pts, radius=500.0, arc=float(rng.uniform(0.8, 2.0)), lanes=lanes_n, width=float(rng.uniform(3.0, 4.0)))
    sc = T._scene(n_env, A, 1000 + seed, rails, n_static=int(rng.integers(0, 400)))
    if seed % 3 == 1:   # long vehicles: many cells per pose
        rows = sc["rows"].copy(); box = rows[:, L.P_SHAPE] == L.SHAPE_OBB
        rows[box, L.P_LENGTH] = rng.uniform(6.0, 18.0); sc["rows"] = rows
    if MG.geometry_budget(n_env, A, static=sc["statics"], lanes=[sc["lanes"]] * n_env)["tier"] != "hbm_grid":
        continue
    grid += 1
    pool = ParticipantPool(n_env, A)
    static, lanes = T._load(pool, sc, n_env, [sc["lanes"]] * n_env)
    for step in range(3):
        pool.set_actions(np.float32(rng.uniform(-2.0, 2.0, n_env * A)), np.float32(rng.uniform(-0.3, 0.3, n_env * A)))
        pool.step(100)
        x, y, h = (pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING))
        got, got_env = pool.download(L.F_FLAGS), pool.download(L.F_ENV_FLAGS)
        want, want_env = O.collide(sc["rows"], n_env, A, x, y, h, sc["tid"], sc["active"], static, None, None, lanes, 0)
        bad = int((got != want).sum()) + int((got_env != want_env).sum())
        total += got.size; bad_total += bad
        seen += [int((got & L.FLAG_OFF_LANE).astype(bool).sum()), int((got & L.FLAG_COLLISION_STATIC).astype(bool).sum()), got.size]
        if bad:
            i = np.nonzero(got != want)[0][:5]
            print("MISMATCH seed", seed, "step", step, "count", bad, "first", i, got[i], want[i])
    pool.close()
print(f"grid tier soak: {grid} scenes of {n_seeds} seeds on the grid tier, {total} participant checks ({seen[0]} off lane, {seen[1]} static hits), "
      f"{bad_total} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad_total or grid == 0 else 0)
