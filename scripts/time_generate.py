"""ParkingLotGenerator kernel (row f4): wall time of t2d_generate_parking (launch + D2H of the scene arrays) and of
the whole reset (generate + host packing + upload) per batch size, next to the oracle on the host cores."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle as O
from tactics2d_amd.generator import ParkingLotGenerator
from tactics2d_amd.pool import ParticipantPool

O.build()
gen = ParkingLotGenerator((4.284, 1.81), 0.5)
gen.generate(64, 0)
for n in (4096, 65536, 524288):
    t = time.perf_counter(); sc = gen.generate(n, 1); t_gen = time.perf_counter() - t
    t = time.perf_counter(); w = O.generate_parking(1, n, 0.5, (4.284, 1.81)); t_cpu = time.perf_counter() - t
    assert np.array_equal(sc.start, w["start"])
    line = f"{n:7d} scenes: device generate {1e3 * t_gen:8.2f} ms ({n / t_gen:.3e} scenes/s)   oracle 1 core {1e3 * t_cpu:8.2f} ms ({n / t_cpu:.3e}/s)"
    if n <= 65536:
        pool = ParticipantPool(n, 1)
        t = time.perf_counter(); sc.scene().load(pool); pool.sync() if hasattr(pool, "sync") else None
        line += f"   install (pack + upload + reset) {1e3 * (time.perf_counter() - t):8.2f} ms"
        pool.close()
    print(line)
