import os, sys, time
sys.path.insert(0,'/root/repo')
import numpy as np, torch
from tactics2d_amd import scenarios as S, mapgeom as MG
from tactics2d_amd.pool import ParticipantPool
sc = S.mixed(4096, 64, seed=3)
print("segs", os.environ.get("T2D_ROUNDABOUT_SEGS"), "lanes per env (first 3):", [int(sc.lanes[0][e+1]-sc.lanes[0][e]) for e in range(3)])
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
print("occupancy (wgs/CU, LDS bytes):", pool.step_occupancy(), "geometry bytes/launch", pool.geometry_bytes_per_launch())
dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
sets = [sc.sample_actions(rng) for _ in range(4)]
a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous(); a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
st = torch.cuda.Stream(device=dev)
pool.bind_actions(a0.data_ptr(), a1.data_ptr())
for _ in range(40): pool.step_n(20, 100, 0, st.cuda_stream)
torch.cuda.synchronize()
for rep in range(3):
    t=time.perf_counter()
    for _ in range(100): pool.step_n(20, 100, 0, st.cuda_stream)
    torch.cuda.synchronize(); print("chained us/step %.2f" % (1e6*(time.perf_counter()-t)/2000))
pool.close()
