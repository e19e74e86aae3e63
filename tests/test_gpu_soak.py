"""Soak of the step kernel's wave-level LDS synchronisation.  wave_sync() (t2d_collide.hip) issues no s_waitcnt: it relies on
the LDS executing one wave's operations in the order they were issued.  Here the product build and a build in which every
such sync first waits for all of the wave's LDS operations (libt2d_hip_waitcnt.so, -DT2D_WAVE_SYNC_WAITCNT, built by
__graft_entry__.build()) step the metric scene through thousands of steps -- exact integrator, auto-reset on, ~25 sync
points per wave and step, 4096 waves per step -- and every bit of state, flags, statuses and rewards must agree; so must the
chained launches (t2d_step_n) of the product build.  (The reference loop this stands for: envs/parking.py:219-256.)"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from tactics2d_amd import layout as L, scenarios as S
from tactics2d_amd.pool import ParticipantPool
steps, chained = int(sys.argv[1]), sys.argv[2] == "chain"
dev = torch.device("cuda", 0)
sc = S.parking(int(sys.argv[3]), seed0=3) if len(sys.argv) > 4 and sys.argv[4] == "parking" else S.mixed(int(sys.argv[3]), 64, seed=3)
if sc.name.startswith("parking"):
    sc.status.update(max_step=40, no_action_max_step=5)   # short episodes: time limits and NoAction verdicts every few steps
rng = np.random.default_rng(11)
sets = [sc.sample_actions(rng) for _ in range(32)]
a0 = torch.from_numpy(np.stack([s[0] for s in sets])).to(dev).contiguous()
a1 = torch.from_numpy(np.stack([s[1] for s in sets])).to(dev).contiguous()
pool = ParticipantPool(sc.n_env, sc.A); sc.load(pool)
pool.set_integrator_variant("exact"); pool.set_auto_reset(True)
form = pool.step_form(32 if chained else 1)
h = hashlib.sha256()
done = 0
while done < steps:
    if chained:
        pool.bind_actions(a0.data_ptr(), a1.data_ptr())
        pool.step_n(32, 100, sc.n)
    else:
        for k in range(32):
            pool.bind_actions(a0.data_ptr() + 4 * sc.n * k, a1.data_ptr() + 4 * sc.n * k)
            pool.step(100)
    done += 32
    if done %% 64 == 0:   # the record ring holds the last 64 steps' rewards / statuses of every env
        for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_FLAGS, L.F_RECORD, L.F_CNT_STEP):
            h.update(pool.download(f).tobytes())
flags = pool.download(L.F_FLAGS)
print(json.dumps(dict(sha=h.hexdigest(), steps=done, flagged=float((flags != 0).mean()), form=form)))
"""


def _run(lib, steps, mode, n_env=4096, scene="mixed"):
    env = dict(os.environ, T2D_LIB_NAME=lib)
    out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), str(steps), mode, str(n_env), scene], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_waitcnt_free_wave_sync_and_chained_launches_over_thousands_of_steps():
    check = os.path.join(ROOT, "tactics2d_amd", "libt2d_hip_waitcnt.so")
    if not os.path.exists(check):   # normally built by __graft_entry__.build(); hipcc is in the image
        from tactics2d_amd import build as B
        B.build_check_lib()
    steps = 2048
    ref = _run("libt2d_hip.so", steps, "step")
    assert ref["steps"] == steps and ref["flagged"] > 0.05
    assert _run("libt2d_hip_waitcnt.so", steps, "step")["sha"] == ref["sha"], "the waitcnt-free wave_sync changed a result"
    assert _run("libt2d_hip.so", steps, "chain")["sha"] == ref["sha"], "chained launches differ from separate launches"


def test_integrator_waves_a_step_ahead_over_thousands_of_steps():
    """The PIPE form of t2d_step_n (a pool of at most one workgroup per CU: integrator waves, event waves and lane waves hand
    each other the state and the verdicts through LDS words, t2d_collide.hip) against separate launches, and against the
    conservative build, in which every hand-shake first waits for the wave's LDS operations: 8192 steps of 768 mixed envs =
    1.6 M hand-overs, 0.5 M of them after a wrong speculation (an episode ends in about a third of the env-steps).  One hash."""
    steps, n_env = 8192, 768
    ref = _run("libt2d_hip.so", steps, "step", n_env)
    assert ref["steps"] == steps and ref["flagged"] > 0.05 and ref["form"] in ("step", "step_split")
    got = _run("libt2d_hip.so", steps, "chain", n_env)
    assert got["form"] == "loop_pipe"
    assert got["sha"] == ref["sha"], "the PIPE form differs from separate launches"
    chk = _run("libt2d_hip_waitcnt.so", steps, "chain", n_env)
    assert chk["form"] == "loop_pipe" and chk["sha"] == ref["sha"], "the conservative build's PIPE form differs"


def test_single_ego_integrator_waves_over_thousands_of_steps():
    """The same for the single-ego kernel's PIPE form (t2d_ego.hip): 2048 parking envs, 4096 steps, IoU events on, episodes of
    at most 40 steps -- separate launches, fragments, and the fragments of the conservative build: one hash."""
    steps, n_env = 4096, 2048
    ref = _run("libt2d_hip.so", steps, "step", n_env, "parking")
    assert ref["steps"] == steps and ref["form"] == "ego"
    got = _run("libt2d_hip.so", steps, "chain", n_env, "parking")
    assert got["form"] == "ego_loop_pipe" and got["sha"] == ref["sha"], "the single-ego PIPE form differs from separate launches"
    chk = _run("libt2d_hip_waitcnt.so", steps, "chain", n_env, "parking")
    assert chk["sha"] == ref["sha"], "the conservative build's single-ego PIPE form differs"


def test_host_frames_through_mapped_memory_over_thousands_of_steps():
    """The Gym-API host path's mapped-memory mode (T2D_FRAME_ZEROCOPY: the step kernel reads the actions from, the pack and
    lidar kernels write the frame to, pinned host memory -- no copy commands) relies on one thing: what a kernel wrote to
    mapped host memory is visible to the host once the stream has been synchronised.  3000 steps of 512 regenerating parking
    envs (episodes end and restart all the time), frame by frame against a second env driven through copy commands: every
    section of every frame bit-identical, lidar included."""
    import numpy as np
    from tactics2d_amd.envs import VecParkingEnv
    n = 512
    a = VecParkingEnv(n, max_step=40, seed=11, auto_reset=True, scene_source="generator", zero_copy=True, copy=False)
    b = VecParkingEnv(n, max_step=40, seed=11, auto_reset=True, scene_source="generator", zero_copy=False, copy=False)
    oa, ia = a.reset(); ob, ib = b.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(0)
    acts = [a.action_space.sample(rng, n) for _ in range(16)]
    ended = 0
    for t in range(3000):
        x = acts[t & 15]
        if t % 97 == 0:
            x = np.zeros_like(x)
        ra, rb = a.step(x), b.step(x)
        for u, v in zip(ra[:4], rb[:4]):
            assert np.array_equal(np.ascontiguousarray(u).view(np.uint8), np.ascontiguousarray(v).view(np.uint8)), t
        for k in ("iou", "lidar", "episode", "target_area", "target_heading", "diff_position", "diff_angle", "diff_heading"):
            assert np.array_equal(np.ascontiguousarray(ra[4][k]).view(np.uint8), np.ascontiguousarray(rb[4][k]).view(np.uint8)), (t, k)
        ended += int((ra[2] | ra[3]).sum())
    assert ended > 20 * n   # every env went through dozens of episodes
    a.close(); b.close()
