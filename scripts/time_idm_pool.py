"""A pool whose participants (all but the ego of every env) are IDM-controlled: one t2d_step per step (idm_kernel + step launch)
against t2d_step_n fragments (the PIPE form's integrator waves run the controllers).  python scripts/time_idm_pool.py [cfg3|cfg5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tactics2d_amd import scenarios as S, layout as L
from tactics2d_amd.pool import ParticipantPool
from tactics2d_amd.controller import IDMController, install
dev = torch.device("cuda", 0)
for name in (sys.argv[1:] or ["cfg3", "cfg5"]):
    sc = {"cfg3": lambda: S.highway(1024, 64, seed=1), "cfg5": lambda: S.mixed(1024, 64, seed=3), "cfg4": lambda: S.intersection(512, 32, seed=2)}[name]()
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
    for key, fn in (("separate", single), ("fragments", frag)):
        fn(600); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); fn(2000); torch.cuda.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t) / 2000)
        res[key] = best
    print(f"{name} with IDM agents: {res['separate']:.2f} us per step as idm + step launches ({pool.step_form(1)}), "
          f"{res['fragments']:.2f} as t2d_step_n fragments of 20 ({pool.step_form(20)})")
    pool.close()
