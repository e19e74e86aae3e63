"""A/B of step-kernel builds and step modes on one box, one process per library build.

    python scripts/ab_step.py libt2d_hip.so libt2d_base.so ...        (names inside tactics2d_amd/)

Per build (T2D_LIB_NAME): the metric scene (4096 x 64, auto-reset on, device-resident action ring) as
  * one launch per step (t2d_step), outputs all / state only (t2d_set_outputs),
  * chained launches (t2d_step_n) of 20 and 100 steps, both wave-priority rules,
and cfg3 / cfg4 / cfg5 (per-GPU shards) per step and chained.  Every figure: wall time of >= 2000 steps after a 600-step
clock ramp, best of 3 repetitions.  A checksum of the final state / flags / record ring (exact integrator, 64 steps from the
snapshot) is printed per build: equal checksums = bit-identical results.
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    from tactics2d_amd import layout as L, scenarios as S
    from tactics2d_amd.pool import ParticipantPool
    dev = torch.device("cuda", 0)
    out = {}

    def scene(name):
        return {"metric": lambda: S.mixed(4096, 64, seed=3), "cfg3": lambda: S.highway(1024, 64, seed=1),
                "cfg4": lambda: S.intersection(512, 32, seed=2), "cfg5": lambda: S.mixed(1024, 64, seed=3)}[name]()

    def timed(fn, steps, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fn(steps)
            torch.cuda.synchronize()
            best = min(best, 1e6 * (time.perf_counter() - t) / steps)
        return best

    only = os.environ.get("T2D_AB_ONLY")   # e.g. "metric": one scene
    for name in (only,) if only else ("cfg3", "cfg4", "cfg5") if os.environ.get("T2D_AB_SMALL") else ("metric", "cfg3", "cfg4", "cfg5"):
        sc = scene(name)
        rng = np.random.default_rng(5)
        K = 4
        sets = [sc.sample_actions(rng) for _ in range(K)]
        a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
        a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
        pool = ParticipantPool(sc.n_env, sc.A)
        sc.load(pool)
        pool.set_auto_reset(True)
        st = torch.cuda.Stream(device=dev)

        def single(n):
            for k in range(n):
                pool.bind_actions(a0.data_ptr() + 4 * sc.n * (k & 3), a1.data_ptr() + 4 * sc.n * (k & 3))
                pool.step(sc.interval_ms, st.cuda_stream)

        def chained(frag):
            def run(n):
                pool.bind_actions(a0.data_ptr(), a1.data_ptr())
                for _ in range(n // frag):
                    pool.step_n(frag, sc.interval_ms, 0, st.cuda_stream)   # (one action set repeated: the ring has 4)
            return run

        if name == "metric":   # bit-identity checksum: exact integrator, 64 steps
            pool.set_integrator_variant("exact")
            single(64)
            h = hashlib.sha256()
            for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_STATUS, L.F_REWARD, L.F_RECORD, L.F_CNT_STEP):
                h.update(pool.download(f).tobytes())
            out["checksum"] = h.hexdigest()[:16]
            pool.restore(False)
            pool.set_integrator_variant("fast")
        single(600)
        torch.cuda.synchronize()
        steps = 2000
        r = {}
        r["single"] = timed(single, steps)
        if name == "metric":
            pool.set_outputs(velocity=False, applied=False)
            r["single_state_only"] = timed(single, steps)
            pool.set_outputs()
        if hasattr(pool, "step_n") and os.environ.get("T2D_AB_CHAIN", "1") == "1":
            for rule in (1, 0):
                pool.set_step_chaining(True, rule)
                for frag in (20, 100) if name == "metric" else (20,):
                    r[f"chain{frag}_rule{rule}"] = timed(chained(frag), steps)
                if name == "metric" and rule == 1:   # what the driver's `--steps 20` sees: ONE fragment between two synchronisations
                    def one_by_one(n, run=chained(20)):
                        for _ in range(n // 20):
                            run(20)
                            torch.cuda.synchronize()
                    r["chain20_synced_each"] = timed(one_by_one, steps)
                if name != "metric" and os.environ.get("T2D_AB_SMALL"):   # the looping forms one by one
                    for mode, key in ((4, "pipe1"), (3, "loop")):
                        pool.set_step_chaining(mode, rule)
                        r[f"{key}_rule{rule}"] = timed(chained(20), steps)
        out[name] = r
        pool.close()
    print("AB_RESULT", os.environ.get("T2D_LIB_NAME"), out, flush=True)


if __name__ == "__main__":
    if os.environ.get("T2D_AB_CHILD"):
        child()
    else:
        for lib in sys.argv[1:]:
            env = dict(os.environ, T2D_LIB_NAME=lib, T2D_AB_CHILD="1", GPU_MAX_HW_QUEUES="8", T2D_ALLOW_MISSING_SYMBOLS="1")
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, timeout=600)
