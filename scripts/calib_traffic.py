"""Known-byte calibration workload for the rocprofv3 FETCH_SIZE / WRITE_SIZE counters: t2d_restore
mode 0 streams 7 arrays in and 8 arrays out with the same 4-B-per-lane coalesced pattern as the
integrator (MI355X_MICROARCH.md: these counters are only calibrated for 16-B/lane reads)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tactics2d_amd import scenarios as S
from tactics2d_amd.pool import ParticipantPool
sc = S.mixed(4096, 64, seed=3)
pool = ParticipantPool(sc.n_env, sc.A)
sc.load(pool)
for _ in range(40):
    pool.restore(done_only=False)
pool.sync()
print("N", sc.n, "read_bytes_per_launch", 7 * 4 * sc.n, "write_bytes_per_launch", 8 * 4 * sc.n)
