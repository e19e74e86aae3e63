"""What a step costs on the HBM grid tier (maps beyond the 32 KiB LDS record: DESIGN.md 4.3c).

    python scripts/mapgrid_timing.py [n_env]

The scene of tests/test_gpu_mapgrid.py -- a 1-km four-lane curved road of 220-point rails, 876 lane pieces per env, 24 boxes beside
it, 64 participants per env scattered along it -- in n_env (default 1024) envs.  Timed: t2d_step (= integrate -> map events -> events
+ status, three launches) per step over 200 steps, and each kernel by itself (t2d_integrate / the map-events launch inside
t2d_collide / t2d_collide), wall time of back-to-back launches; beside it the same participants on a short road that fits the LDS
record (one fused launch)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import test_gpu_mapgrid as T  # noqa: E402
from tactics2d_amd import layout as L, mapgeom as MG  # noqa: E402
from tactics2d_amd.pool import ParticipantPool  # noqa: E402

n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = 64
dev = torch.device("cuda", 0)


def loop(fn, n=200, warm=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t) / n


res = {}
cases = (("grid_tier_876_pieces", T._road(n_pts=220)), ("grid_tier_876_pieces_one_map_for_all_envs", T._road(n_pts=220)),
         ("lds_record_44_pieces", T._road(n_pts=12, arc=0.2)))
if os.environ.get("T2D_MG_ONLY"):   # (profiling runs: the grid tier alone)
    cases = cases[:2]
for name, rails in cases:
    sc = T._scene(n_env, A, 17, rails, n_static=24 if name.startswith("grid") else 6)
    if name.endswith("one_map_for_all_envs"):   # every env on the SAME map, obstacles included (what a pool on one reference map is):
        sc["statics"] = [sc["statics"][0]] * n_env   # the library then keeps one grid and one set of registrations for all of them
    if not name.startswith("grid"):   # participants on the short road only
        keep = np.abs(np.arctan2(sc["x"], 500.0)) < 0.09
        sc["active"] = (sc["active"].astype(bool) & keep).astype(np.uint8)
    t0 = time.perf_counter()
    pool = ParticipantPool(n_env, A)
    T._load(pool, sc, n_env, [sc["lanes"]] * n_env)
    pool.snapshot()
    pool.set_auto_reset(True)
    load_s = time.perf_counter() - t0
    rng = np.random.default_rng(3)
    pool.set_actions(np.float32(rng.uniform(-2, 2, n_env * A)), np.float32(rng.uniform(-0.3, 0.3, n_env * A)))
    st = torch.cuda.Stream(device=dev)
    r = dict(form=pool.step_form(1), load_seconds=round(load_s, 2), pieces_per_env=len(sc["lanes"]),
             budget=MG.geometry_budget(n_env, A, lanes=[sc["lanes"]] * n_env)["tier"] if hasattr(MG, "geometry_budget") else None)
    r["step_us"] = round(loop(lambda: pool.step(100, st.cuda_stream)), 2)
    r["integrate_us"] = round(loop(lambda: pool.integrate(100, st.cuda_stream)), 2)
    r["collide_us"] = round(loop(lambda: pool.collide(st.cuda_stream)), 2)       # (grid tier: map events + events, two launches)
    r["check_status_us"] = round(loop(lambda: pool.check_status(100, st.cuda_stream)), 2)
    flags = pool.download(L.F_FLAGS)
    r["off_lane_frac"] = float(((flags & L.FLAG_OFF_LANE) != 0).mean())
    res[name] = r
    print(name, r, flush=True)
    pool.close()
print("MAPGRID_TIMING", res)
