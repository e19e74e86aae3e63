import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import scenarios as S
from tactics2d_amd.pool import ParticipantPool
for name, sc, part in (("cfg2 parking 4096x1", S.parking(4096), False), ("metric mixed 4096x64", S.mixed(4096, 64), True)):
    pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.lidar_config(360, 20.0, part)
    for _ in range(1500): pool.lidar_scan()   # ~40 ms: the GPU has left its idle clocks (bench.py: clock_warm)
    pool.profile_enable(True)
    for _ in range(200): pool.lidar_scan()
    ms, n = pool.profile_read(3)
    out_bytes = sc.n_env * 360 * 4
    print(name, "lidar avg us", 1e3 * ms / n, "output GB/s", out_bytes / (ms / n * 1e-3) / 1e9)
    pool.close()
