"""Randomised soak of the lidar kernel against the oracle (bit-exact): random scenes, beam counts 90 / 360 / 1024,
participants on and off, plus scenes with obstacle vertices very close to the sensor (large angular error of the
fp32 span estimate).  Usage on the GPU box: python tests/soak/soak_lidar.py [n_seeds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from oracle import oracle as O
from tactics2d_amd import layout as L
from tactics2d_amd.pool import ParticipantPool
O.build(); O.set_threads(min(16, os.cpu_count() or 1))
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
t0 = time.time(); total = bad_total = 0; hits = 0
for seed in range(n_seeds):
    for (n_env, A, extent, kw, beams, rng_max, part) in [
            (40, 1, (30.0, 20.0), dict(n_static=12, n_lanes=0), 360, 20.0, False),
            (24, 16, (40.0, 30.0), dict(n_static=8, n_lanes=0), 360, 20.0, True),
            (12, 64, (60.0, 16.0), dict(n_static=6, n_lanes=0), 1024, 35.0, True),
            (30, 4, (12.0, 8.0), dict(n_static=5, n_lanes=0), 90, 8.0, True),
            (30, 1, (3.0, 3.0), dict(n_static=10, n_lanes=0), 720, 20.0, False),   # cramped: vertices centimetres away
            # <= 32 edges, static only: the scan drops back edges behind a front edge of their ring (occlusion culling)
            (48, 1, (24.0, 16.0), dict(n_static=8, n_lanes=0), 360, 20.0, False),
            (36, 1, (5.0, 4.0), dict(n_static=7, n_lanes=0), 360, 20.0, False),
            (36, 1, (2.0, 2.0), dict(n_static=6, n_lanes=0), 1024, 12.0, False),
            # 33 .. 48 edges (a generated parking lot's 9 .. 12 quads): culled since round 4, edge bits 0..47 + ring bits 48..63
            (36, 1, (24.0, 16.0), dict(n_static=11, n_lanes=0), 360, 20.0, False),
            (36, 1, (4.0, 3.0), dict(n_static=12, n_lanes=0), 1024, 12.0, False),
            (36, 1, (8.0, 6.0), dict(n_static=9, n_lanes=0), 720, 20.0, False)]:
        rng = np.random.default_rng(7000 * seed + n_env * 100 + A)
        sc = H.random_scene(rng, n_env, A, extent, **kw)
        pool = ParticipantPool(n_env, A)
        pool.set_param_table(sc["rows"])
        pool.set_static_geometry(sc["static"], sc.get("boundary"), sc.get("boundary_valid"))
        pool.reset(sc["x"], sc["y"], sc["heading"], np.zeros(n_env * A, np.float32), sc["type_id"], active=sc["active"])
        pool.lidar_config(beams, rng_max, part)
        pool.lidar_scan()
        got = pool.download(L.F_LIDAR)
        pool.close()
        want = O.lidar(sc["rows"], n_env, A, 0, sc["x"], sc["y"], sc["heading"], sc["type_id"], sc["active"], sc["static"],
                       int(part), beams, rng_max, trig=0)
        bad = int((got.view(np.uint32) != want.view(np.uint32)).sum())
        total += got.size; bad_total += bad; hits += int(np.isfinite(want).sum())
        if bad:
            e, k = np.nonzero(got.view(np.uint32) != want.view(np.uint32))
            print("MISMATCH seed", seed, n_env, A, beams, "count", bad, "first", e[:3], k[:3], got[e[:3], k[:3]], want[e[:3], k[:3]])
print(f"lidar soak: {n_seeds} seeds, {total} beams ({100 * hits / total:.0f} % with a return), {bad_total} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad_total else 0)
