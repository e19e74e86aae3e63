"""Soak of the ParkingLotGenerator kernel against its CPU restatement: many (seed, proportion, vehicle size, first_env)
combinations, every output array compared bit for bit.   python tests/soak/soak_generate.py [n_configs] [scenes_per_config]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from tactics2d_amd.generator import ParkingLotGenerator

O.build()
O.set_threads(min(16, os.cpu_count() or 1))
n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 60
per = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
rng = np.random.default_rng(123)
t0 = time.time(); bad = 0; total = 0; flagged = 0; attempts_max = 0
for c in range(n_cfg):
    seed = int(rng.integers(0, 2**63)); prop = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
    size = (float(rng.uniform(3.2, 6.0)),); size = (size[0], float(rng.uniform(1.5, min(2.6, size[0]))))
    first = int(rng.integers(0, 2**40))
    got = ParkingLotGenerator(size, prop).generate(per, seed, first_env=first)
    want = O.generate_parking(seed, per, prop, size, first_env=first, trig=1)
    same = (np.array_equal(got.info, want["info"]) and np.array_equal(got.n_quads, want["n_quads"]) and
            np.array_equal(got.quad_id, want["quad_id"]) and np.array_equal(got.start, want["start"]) and
            np.array_equal(got.target_heading, want["target_heading"]) and
            np.array_equal(got.quads.view(np.uint32), want["quads"].view(np.uint32)) and
            np.array_equal(got.target.view(np.uint32), want["target"].view(np.uint32)) and
            np.array_equal(got.boundary, want["boundary"]))
    bad += 0 if same else 1; total += per
    flagged += int(((got.info & 0x1e) != 0).sum()); attempts_max = max(attempts_max, int(((got.info >> 8) & 255).max()))
print(f"generator soak: {n_cfg} configurations x {per} scenes = {total} scenes, {bad} configurations with a mismatch, "
      f"{flagged} flagged scenes, most obstacle attempts {attempts_max}, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
