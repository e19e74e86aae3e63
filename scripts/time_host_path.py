"""The Gym-API host path, timed: VecParkingEnv.step (numpy actions in, numpy 5-tuple out) and the single-env ParkingEnv.step,
against the per-field path it replaced (round 4: two blocking uploads, the step, ~12 blocking t2d_download calls and a second
lidar launch per step) on the same box.  Prints one JSON object; bench.py's next_rows carries the same figures."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tactics2d_amd import layout as L
from tactics2d_amd.envs import ParkingEnv, VecParkingEnv


def per_field_step(env, a):
    """What VecParkingEnv.step cost before the packed frame: one blocking copy per field."""
    m = env.scenario_manager
    m.step(a[:, 1], a[:, 0])
    pool = m.pool
    status = pool.download(L.F_STATUS); reward = pool.download(L.F_REWARD)
    obs = m.get_observation()
    frame = pool.download(L.F_FRAME_MS); iou = pool.download(L.F_IOU)
    pool.lidar_scan(); lidar = pool.download(L.F_LIDAR)
    return obs, reward, status, frame, iou, lidar


def time_vec(n, steps, **kw):
    env = VecParkingEnv(n, max_step=200, auto_reset=True, seed=1, **kw); env.reset()
    rng = np.random.default_rng(0)
    acts = [env.action_space.sample(rng, n) for _ in range(8)]
    for k in range(30): env.step(acts[k & 7])
    t = time.perf_counter()
    for k in range(steps): env.step(acts[k & 7])
    el = (time.perf_counter() - t) / steps
    env.close()
    return 1e6 * el


def time_per_field(n, steps):
    env = VecParkingEnv(n, max_step=200, auto_reset=True, seed=1); env.reset()
    rng = np.random.default_rng(0)
    acts = [env.action_space.sample(rng, n) for _ in range(8)]
    for k in range(10): per_field_step(env, acts[k & 7])
    t = time.perf_counter()
    for k in range(steps): per_field_step(env, acts[k & 7])
    el = (time.perf_counter() - t) / steps
    env.close()
    return 1e6 * el


def time_single(steps, **kw):
    env = ParkingEnv(max_step=int(2e4), seed=0, **kw); env.reset()
    rng = np.random.default_rng(0)
    acts = [env.action_space.sample(rng) * 0.2 for _ in range(64)]
    for k in range(200): env.step(acts[k & 63])
    n_reset = 0
    t = time.perf_counter()
    for k in range(steps):
        o, r, te, tr, info = env.step(acts[k & 63])
        if te or tr:
            env.reset(); n_reset += 1
    el = (time.perf_counter() - t) / steps
    env.close()
    return 1e6 * el, n_reset


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    out = {"n_envs": n}
    out["vec_step_us_lidar_copy"] = time_vec(n, steps)
    out["vec_step_us_lidar_views"] = time_vec(n, steps, copy=False)
    out["vec_step_us_nolidar_copy"] = time_vec(n, steps, info_lidar=False)
    out["vec_step_us_nolidar_views"] = time_vec(n, steps, info_lidar=False, copy=False)
    out["vec_step_us_nolidar_views_zero_copy"] = time_vec(n, steps, info_lidar=False, copy=False, zero_copy=True)
    out["vec_step_us_lidar_views_zero_copy"] = time_vec(n, steps, copy=False, zero_copy=True)
    out["vec_step_us_generator_nolidar_views"] = time_vec(n, steps, info_lidar=False, copy=False, scene_source="generator")
    out["vec_step_us_per_field_round4"] = time_per_field(n, max(steps // 4, 20))
    us, nr = time_single(3000)
    out["parking_env_single_step_us"] = us
    out["parking_env_single_steps_per_s"] = 1e6 / us
    out["parking_env_single_resets_in_run"] = nr
    out["parking_env_single_step_us_copy_commands"] = time_single(1500, zero_copy=False)[0]
    out["parking_env_single_step_us_no_lidar"] = time_single(1500, info_lidar=False)[0]
    print(json.dumps(out))
