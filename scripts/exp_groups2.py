import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd import scenarios as S
from tactics2d_amd.pipeline import EnvGroups
dev = torch.device("cuda", 0)
sc = S.mixed(4096, 64, seed=3)
rng = np.random.default_rng(0)
ring = []
for _ in range(4):
    a0, a1 = sc.sample_actions(rng)
    ring.append((torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)))
for G in (1, 2, 4):
    for mode in ("fixed", "ring", "python-loop"):
        eg = EnvGroups(sc, G)
        eg.configure(lambda p: p.set_auto_reset(True))
        eg.bind_actions(*ring[0])
        torch.cuda.synchronize()
        def step(k):
            if mode == "ring": eg.bind_actions(*ring[k & 3])
            if mode == "python-loop":
                for p, s in zip(eg.pools, eg.streams): p.step(100, s.cuda_stream)
            else:
                eg.step(100)
        if mode == "python-loop":
            for (lo, hi), p in zip(eg.bounds, eg.pools):
                p.bind_actions(ring[0][0].data_ptr() + 4 * lo * 64, ring[0][1].data_ptr() + 4 * lo * 64)
        for k in range(50): step(k)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(600): step(k)
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        # host-only enqueue cost
        t = time.perf_counter()
        for k in range(200): step(k)
        host = (time.perf_counter() - t) / 200
        torch.cuda.synchronize()
        print(f"G={G} {mode:12s}: {1e6 * el / 600:.2f} us/step; host enqueue {1e6 * host:.2f} us/step")
        eg.close()
