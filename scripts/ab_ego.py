"""A/B of library builds on the single-ego path (BASELINE config 2, 4096 parking envs): one launch per step (ego_step_kernel),
t2d_step_n fragments, the vector env's device-resident step (ego step + lidar) and the host path, one process per build.
    python scripts/ab_ego.py libA.so libB.so ..."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import hashlib
    import numpy as np, torch
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.envs import VecParkingEnv
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    out = {}
    sc = S.parking(4096)
    rng = np.random.default_rng(5)
    sets = [sc.sample_actions(rng) for _ in range(4)]
    a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
    a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
    pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
    st = torch.cuda.Stream(device=dev)

    def single(n):
        for k in range(n):
            pool.bind_actions(a0.data_ptr() + 4 * sc.n * (k & 3), a1.data_ptr() + 4 * sc.n * (k & 3))
            pool.step(sc.interval_ms, st.cuda_stream)

    def chained(n):
        pool.bind_actions(a0.data_ptr(), a1.data_ptr())
        for _ in range(n // 20):
            pool.step_n(20, sc.interval_ms, 0, st.cuda_stream)

    def timed(fn, steps, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(); t = time.perf_counter(); fn(steps); torch.cuda.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t) / steps)
        return best
    pool.set_integrator_variant("exact"); single(64)
    h = hashlib.sha256()
    for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_STATUS, L.F_REWARD, L.F_IOU, L.F_CNT_STEP):
        h.update(pool.download(f).tobytes())
    out["checksum"] = h.hexdigest()[:16]
    pool.restore(False); pool.set_integrator_variant("fast")
    single(600); torch.cuda.synchronize()
    out["single"] = timed(single, 2000)
    out["chain20"] = timed(chained, 2000)
    pool.close()
    env = VecParkingEnv(4096, max_step=200, auto_reset=True, seed=1, info_lidar=False); env.reset()
    lo = torch.tensor([-0.524, -2.0], device=dev); hi = torch.tensor([0.524, 2.0], device=dev)
    acts = [lo + (hi - lo) * torch.rand((4096, 2), device=dev) for _ in range(8)]
    k = [0]
    def vec(n):
        for _ in range(n):
            env.step_torch(acts[k[0] & 7]); k[0] += 1
    vec(100); torch.cuda.synchronize()
    out["vec_step_torch"] = timed(vec, 1000)
    hacts = [env.action_space.sample(rng, 4096) for _ in range(8)]
    def host(n):
        for _ in range(n):
            env.step(hacts[k[0] & 7]); k[0] += 1
    host(50)
    out["vec_step_host_nolidar"] = timed(host, 500)
    env.close()
    print("AB_RESULT", os.environ.get("T2D_LIB_NAME"), out, flush=True)


if __name__ == "__main__":
    if os.environ.get("T2D_AB_CHILD"):
        child()
    else:
        for lib in sys.argv[1:]:
            env = dict(os.environ, T2D_LIB_NAME=lib, T2D_AB_CHILD="1", GPU_MAX_HW_QUEUES="8", T2D_ALLOW_MISSING_SYMBOLS="1")
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=600)
