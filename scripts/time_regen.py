"""Latency of the scene-regeneration launch (kernel id 6) as a function of how many envs finish in the step:
pools of n envs with max_step = 1 (every env finishes every step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tactics2d_amd.generator import ParkingLotGenerator
from tactics2d_amd.pool import ParticipantPool
size = (4.284, 1.81)
for n in (1, 64, 1024, 4096, 65536):
    scene = ParkingLotGenerator(size, 0.5).generate(n, 1).scene(max_step=1)
    p = ParticipantPool(n, 1); p.set_param_table(scene.rows); p.set_status_config(**scene.status)
    p.parking_scenes(1, 0.5, size, regenerate=True)
    z = np.zeros(n, np.float32); p.set_actions(z, z)
    for _ in range(5): p.step(100)
    p.profile_enable(True)
    for _ in range(50): p.step(100)
    ms, l = p.profile_read(6); ms2, l2 = p.profile_read(2)
    ep = p.get_parking_scenes().episode
    print(f"{n:6d} envs all finishing: regeneration launch {1e3 * ms / l:8.1f} us, step kernel {1e3 * ms2 / l2:6.1f} us (episodes now {ep.min()}..{ep.max()})")
    p.close()
