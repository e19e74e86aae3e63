"""ParkingLotGenerator on the device (row f4).
 (1) t2d_generate_parking: launch + D2H of the scene arrays, next to the oracle on one host core;
 (2) scenes into a pool: host boundary (generate -> pack -> upload -> reset -> snapshot) vs t2d_parking_scenes
     (generate + install in one launch, nothing crosses PCIe);
 (3) per-step cost of regenerating finished episodes: step alone, step + regeneration launch (kernel ids 2 and 6)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from tactics2d_amd.generator import ParkingLotGenerator
from tactics2d_amd.pool import ParticipantPool

O.build()
size = (4.284, 1.81)
gen = ParkingLotGenerator(size, 0.5)
gen.generate(64, 0)
for n in (4096, 65536, 524288):
    t = time.perf_counter(); sc = gen.generate(n, 1); t_gen = time.perf_counter() - t
    t = time.perf_counter(); w = O.generate_parking(1, n, 0.5, size); t_cpu = time.perf_counter() - t
    assert np.array_equal(sc.start, w["start"])
    print(f"{n:7d} scenes: device generate + D2H {1e3 * t_gen:8.2f} ms ({n / t_gen:.3e} scenes/s)   oracle 1 core {1e3 * t_cpu:8.2f} ms ({n / t_cpu:.3e}/s)")
for n in (4096, 65536):
    scene = gen.generate(n, 1).scene(max_step=40)
    host = ParticipantPool(n, 1)
    t = time.perf_counter(); scene.load(host); t_host = time.perf_counter() - t
    dev = ParticipantPool(n, 1); dev.set_param_table(scene.rows); dev.set_status_config(**scene.status)
    dev.parking_scenes(1, 0.5, size)                      # first call allocates
    t = time.perf_counter(); dev.parking_scenes(1, 0.5, size, regenerate=True); t_dev = time.perf_counter() - t
    print(f"{n:7d} scenes into a pool: host boundary {1e3 * t_host:8.2f} ms   t2d_parking_scenes {1e3 * t_dev:8.2f} ms")
    rng = np.random.default_rng(0)
    for p in (host, dev):
        p.set_auto_reset(True)
    for label, p in (("snapshot auto-reset", host), ("regenerated scenes ", dev)):
        for _ in range(10):
            a0, a1 = scene.sample_actions(rng); p.set_actions(a0, a1); p.step(100)
        p.download(0)
        t = time.perf_counter()
        for _ in range(200):
            p.step(100)
        p.download(0)
        wall = (time.perf_counter() - t) / 200
        p.profile_enable(True)
        for _ in range(200):
            p.step(100)
        ms2, l2 = p.profile_read(2)
        line = f"        {label}: {1e6 * wall:7.1f} us/step wall, step kernel {1e3 * ms2 / max(l2, 1):6.1f} us"
        if p is dev:
            ms6, l6 = p.profile_read(6)
            line += f", regeneration launch {1e3 * ms6 / max(l6, 1):6.1f} us (episodes of {scene.status['max_step']} steps at most)"
        print(line)
    host.close(); dev.close()
