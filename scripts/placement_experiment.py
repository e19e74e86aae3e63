"""Does a cost-aware placement of the step launch pay?  Builds placement maps for the metric scene from per-env cost
estimates (gpurun_out/solo_parts.npy of scripts/solo_cost_probe.py when present: measured solo durations; else the static
proxy) and times 2000 steps with each; checks that flags / status are identical to the identity placement."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tactics2d_amd import scenarios as S, layout as L
from tactics2d_amd import debug as D

sc = S.mixed(4096, 64, 3)
rng = np.random.default_rng(0)
acts = [sc.sample_actions(rng) for _ in range(4)]
XCD_START_US = np.array([0.1, 0.1, 1.5, 2.0, 2.55, 2.55, 2.55, 2.55])   # measured (s_memrealtime at wave start)


def build_map(cost_env, skew=True, rotate=True, a=0.343, cycles_per_us=2200.0):
    C = cost_env.reshape(1024, 4)
    tot = C.sum(1)
    order = np.argsort(-tot)                                  # most expensive workgroups first
    groups = [list(order[4 * i:4 * i + 4]) for i in range(256)]   # four workgroups of similar cost share a CU
    off = XCD_START_US[np.arange(256) % 8] * cycles_per_us if skew else np.zeros(256)
    cus = np.argsort(off, kind="stable")                      # CUs that start early get the expensive groups
    if not skew:
        # no skew model: snake the groups over the CUs instead (balance CU totals)
        groups = [[] for _ in range(256)]
        fold = np.concatenate([np.arange(256), np.arange(255, -1, -1), np.arange(256), np.arange(255, -1, -1)])
        for r, g in enumerate(order): groups[fold[r]].append(g)
        cus = np.arange(256)
    m = np.zeros(1024, np.uint32)
    for rank, c in enumerate(cus):
        g = groups[rank]
        best = None
        for rots in (itertools.product(range(4), repeat=3) if rotate else [(0, 0, 0)]):
            rr = (0,) + tuple(rots)
            s = np.zeros(4)
            for k in range(4):
                for q in range(4):
                    # wave w of the k-th workgroup of a CU runs on SIMD (w - k) mod 4; rotation r: physical wave w steps env (w + r) & 3
                    s[q] += C[g[k]][((q + k) % 4 + rr[k]) % 4]
            if best is None or s.max() < best[0]: best = (s.max(), rr)
        for k in range(4):
            m[c + 256 * k] = g[k] | (best[1][k] << 16)
    return m


def run(wgmap, n=2000):
    pool = D.pool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
    if wgmap is not None: D.set_step_placement(pool, wgmap)
    import torch
    dev = torch.device("cuda", 0)
    ring = [(torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)) for a0, a1 in acts]
    for k in range(600):
        pool.bind_actions(ring[k & 3][0].data_ptr(), ring[k & 3][1].data_ptr()); pool.step(100)
    pool.sync(); t = time.perf_counter()
    for k in range(n):
        pool.bind_actions(ring[k & 3][0].data_ptr(), ring[k & 3][1].data_ptr()); pool.step(100)
    pool.sync(); el = time.perf_counter() - t
    out = (pool.download(L.F_FLAGS).copy(), pool.download(L.F_STATUS).copy(), pool.download(L.F_X).copy())
    pool.close()
    return 1e6 * el / n, out

solo = "scripts/_solo_parts.npy" if os.path.exists("scripts/_solo_parts.npy") else "gpurun_out/solo_parts.npy"
if os.path.exists(solo):
    p = np.load(solo)[:, :14].astype(np.float64); cost = p.sum(1) - p[:, 0]
    print("cost = measured solo durations")
else:
    kind = np.arange(4096) % 3; cost = np.array([22000.0, 24400.0, 22900.0])[kind]
    print("cost = static proxy by env kind")
base_t, base_o = run(None)
print(f"identity placement: {base_t:.2f} us per step")
for name, kw in (("snake + rotations (no skew model)", dict(skew=False)), ("skew-aware + rotations", dict(skew=True)),
                 ("skew-aware, no rotations", dict(skew=True, rotate=False))):
    m = build_map(cost, **kw)
    t, o = run(m)
    same = all(np.array_equal(a, b) for a, b in zip(base_o, o))
    print(f"{name}: {t:.2f} us per step, results identical to identity: {same}")
t2, _ = run(None)
print(f"identity placement again: {t2:.2f} us per step")
