"""Randomised soak of the event kernels against the oracle: many seeds x pool shapes, flags bit-exact.
(The pytest suite runs a fixed handful of these; this is the long version for after kernel changes.)
Usage on the GPU box: python tests/soak/soak.py [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from oracle import oracle as O
O.build(); O.set_threads(min(16, os.cpu_count() or 1))
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
shapes = [(48, 64, (60.0, 16.0), dict(n_static=6, n_lanes=3)), (48, 64, (400.0, 15.0), dict(n_static=0, n_lanes=4)),
          (64, 32, (40.0, 40.0), dict(n_static=8, n_lanes=2)), (33, 48, (40.0, 30.0), dict(n_static=4, n_lanes=0)),
          (120, 8, (20.0, 12.0), dict(n_static=5, n_lanes=2)), (150, 4, (12.0, 8.0), dict(n_static=5, n_lanes=1)),
          (150, 2, (12.0, 8.0), dict(n_static=3, n_lanes=1)), (400, 1, (30.0, 20.0), dict(n_static=12, n_lanes=0)),
          (4, 200, (120.0, 60.0), dict(n_static=16, n_lanes=0)), (6, 100, (90.0, 40.0), dict(n_static=6, n_lanes=3)),
          (40, 64, (30.0, 10.0), dict(n_static=3, n_lanes=2, with_peds=False, inactive_frac=0.5)),
          (40, 64, (25.0, 8.0), dict(n_static=2, n_lanes=6)), (24, 16, (15.0, 6.0), dict(n_static=2, n_lanes=2))]
t0 = time.time(); total = 0; bad_total = 0
for seed in range(n_seeds):
    for (n_env, A, extent, kw) in shapes:
        rng = np.random.default_rng(100000 * seed + n_env * 1000 + A)
        sc = H.random_scene(rng, n_env, A, extent, **kw)
        wf, we = H.oracle_collide(O, sc)
        gf, ge = H.gpu_collide(sc)
        bad = int((gf != wf).sum()) + int((ge != we).sum())
        total += gf.size; bad_total += bad
        if bad:
            i = np.nonzero(gf != wf)[0][:5]
            print("MISMATCH seed", seed, "shape", n_env, A, "count", bad, "first", i, gf[i], wf[i])
# lane unions that abut / overlap / leave holes (off-lane = contains) and polygons of 3..8 vertices (fans of quads)
for seed in range(n_seeds):
    for (n_env, A) in ((48, 64), (64, 32), (100, 8), (5, 150)):
        for kind in ("structured", "polygons"):
            rng = np.random.default_rng(7000000 + 100000 * seed + n_env * 1000 + A)
            sc = H.structured_lane_scene(rng, n_env, A) if kind == "structured" else H.polygon_scene(rng, n_env, A)
            wf, we = H.oracle_collide(O, sc)
            gf, ge = H.gpu_collide(sc)
            bad = int((gf != wf).sum()) + int((ge != we).sum())
            total += gf.size; bad_total += bad
            if bad:
                i = np.nonzero(gf != wf)[0][:5]
                print("MISMATCH", kind, "seed", seed, "shape", n_env, A, "count", bad, "first", i, gf[i], wf[i])
# stepping soak: fused exact step vs oracle (integrate -> collide), teacher-forced on the pool's fp32 state
from tactics2d_amd import layout as L, scenarios as S
from tactics2d_amd.pool import ParticipantPool
step_bad = 0; step_total = 0
for seed in range(max(2, n_seeds // 10)):
    for n_env, A in ((24, 64), (24, 32), (64, 3), (2, 200)):
        sc = S.mixed(n_env, A, seed=1000 + seed)
        pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_integrator_variant("exact")
        rng = np.random.default_rng(seed)
        x, y, h, v = sc.x.copy(), sc.y.copy(), sc.heading.copy(), sc.speed.copy()
        vx = np.float32(v * np.cos(np.float64(h))); vy = np.float32(v * np.sin(np.float64(h)))
        for t in range(8):
            a0, a1 = sc.sample_actions(rng)
            pool.set_actions(a0, a1); pool.step(100)
            g = [pool.download(f) for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY, L.F_FLAGS)]
            O.set_trig(1)
            o = O.integrate(sc.rows, x, y, h, v, vx, vy, a0, a1, sc.type_id, sc.active, 100)
            O.set_trig(0)
            wf, _ = O.collide(sc.rows, sc.n_env, sc.A, g[0], g[1], g[2], sc.type_id, sc.active, sc.static, sc.boundary,
                              sc.boundary_valid, sc.lanes, 1)
            b = sum(int((np.float32(o[:, k]) != g[k]).sum()) for k in range(4)) + int((wf != g[6]).sum())
            step_bad += b; step_total += g[0].size
            x, y, h, v = g[0], g[1], g[2], g[3]
            is_dyn = sc.rows[sc.type_id, 0] == 1
            vx = np.where(is_dyn, vx, g[4]); vy = np.where(is_dyn, vy, g[5])
        pool.close()
print(f"stepping soak: {step_total} participant-steps, {step_bad} mismatches")
bad_total += step_bad
print(f"soak: {n_seeds} seeds x {len(shapes)} shapes, {total} participants, {bad_total} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad_total else 0)
