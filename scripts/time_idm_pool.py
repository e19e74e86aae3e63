"""A pool whose participants (all but the ego of every env) are IDM-controlled: idm_kernel + step launch per step (chaining off),
one step launch with the controllers in its front, and t2d_step_n fragments (PIPE form: the integrator waves run them).  python scripts/time_idm_pool.py [cfg3|cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd import scenarios as S, layout as L
from tactics2d_amd.pool import ParticipantPool
from tactics2d_amd.controller import IDMController, install
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["cfg3", "cfg5"]):
    sc = {"cfg3": lambda: S.highway(1024, 64, seed=1), "cfg5": lambda: S.mixed(1024, 64, seed=3), "cfg4": lambda: S.intersection(512, 32, seed=2), "metric": lambda: S.mixed(4096, 64, seed=3)}[name]()
    pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool); pool.set_auto_reset(True)
    veh = (sc.rows[sc.type_id, L.P_MODEL] != L.MODEL_POINTMASS).reshape(sc.n_env, sc.A)
    cid = np.full((sc.n_env, sc.A), L.IDM_NONE, np.uint8)
    cid[:, 1:] = np.where(veh[:, 1:], 0, L.IDM_NONE)
    install(pool, [IDMController(desired_speed=25.0, horizon=120.0)], cid.reshape(-1))
    rng = np.random.default_rng(0)
    a0, a1 = sc.sample_actions(rng)
    A0 = torch.from_numpy(a0).to(dev); A1 = torch.from_numpy(a1).to(dev)
    pool.bind_actions(A0.data_ptr(), A1.data_ptr())
    st = torch.cuda.Stream(device=dev)
    def single(n):
        for _ in range(n): pool.step(sc.interval_ms, st.cuda_stream)
    def frag(n):
        for _ in range(n // 20): pool.step_n(20, sc.interval_ms, 0, st.cuda_stream)
    res = {}
    for key, fn, chaining in (("separate", single, 0), ("fused", single, 1), ("fragments", frag, 1)):
        pool.set_step_chaining(chaining)
        form = pool.step_form(20 if key == "fragments" else 1)
        fn(600); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); fn(2000); torch.cuda.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t) / 2000)
        res[key] = (best, form)
    print(f"{name} with IDM agents: {res['separate'][0]:.2f} us per step as idm_kernel + step launch ({res['separate'][1]}), "
          f"{res['fused'][0]:.2f} as one launch per step ({res['fused'][1]}), "
          f"{res['fragments'][0]:.2f} as t2d_step_n fragments of 20 ({res['fragments'][1]})")
    pool.close()
