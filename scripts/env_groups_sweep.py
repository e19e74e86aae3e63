"""Experiment: one pool of 4096 envs on one stream vs G env groups on G streams (software pipelining of
independent env groups: one group's start-up latency and tail overlap the other's body)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd import scenarios as S
from tactics2d_amd.pool import ParticipantPool
dev = torch.device("cuda", 0)
def run(groups, n_env_total=4096, A=64, steps=1000):
    n = n_env_total // groups
    pools, acts, streams = [], [], []
    for g in range(groups):
        sc = S.mixed(n, A, seed=3 + g)
        p = ParticipantPool(n, A, 0); sc.load(p); p.set_auto_reset(True)
        a0, a1 = sc.sample_actions(np.random.default_rng(g))
        t0, t1 = torch.from_numpy(a0).to(dev), torch.from_numpy(a1).to(dev)
        p.bind_actions(t0.data_ptr(), t1.data_ptr())
        pools.append(p); acts.append((t0, t1)); streams.append(torch.cuda.Stream(device=dev))
    torch.cuda.synchronize()
    for k in range(50):
        for p, s in zip(pools, streams): p.step(100, s.cuda_stream)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps):
        for p, s in zip(pools, streams): p.step(100, s.cuda_stream)
    torch.cuda.synchronize()
    el = time.perf_counter() - t
    for p in pools: p.close()
    return 1e6 * el / steps, n_env_total * A * steps / el
for total in (4096, 8192, 16384):
    for g in (1, 2, 4, 8):
        us, rate = run(g, total)
        print(f"{total} envs, {g} group(s): {us:.2f} us per step, {rate:.3e} participant-steps/s")
