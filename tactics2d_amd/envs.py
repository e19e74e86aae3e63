"""Gym-style environments on top of the batched hot path.

Mirrors (tactics2d v0.1.9rc3) `ParkingEnv` -- envs/parking.py:44-444 -- for the part that is on
the accelerated path: `step()` = physics update of the ego (`_ParkingScenarioManager.update`, :352-359)
+ ordered status checks (`check_status`, :361-392) + terminated / truncated / reward (:243-250,
:148-166).  `Arrival` (IoU >= 0.95 with the target bay -> COMPLETED, +5, terminated), `NoAction` (IoU with the
previous pose > 0.999 on more than 100 checks, including the reference's quirk of reporting it in
`traffic_status`) and the IoU / distance reward shaping are evaluated in the same launch.  `info["lidar"]` is the 360-beam / 20 m
SingleLineLidar scan of the same poses (t2d_lidar_scan).  The rendered camera image is not on the
accelerated path: the observation returned here is the ego state vector.

gymnasium is not a dependency: `Box` below is the minimal stand-in for `spaces.Box`.
"""
import ctypes as C

import numpy as np

from . import layout as L, scenarios
from ._ffi import ERR_ACTION, T2DError
from .traffic import BatchedScenarioManager, ScenarioStatus, TrafficStatus

_SCENARIO = {int(v): v for v in ScenarioStatus}
_TRAFFIC = {int(v): v for v in TrafficStatus}
MAX_STEER = 0.524  # envs/parking.py:30
MAX_ACCEL = 2.0    # envs/parking.py:31


class InvalidAction(Exception):
    """Same role as gymnasium.error.InvalidAction in envs/parking.py:235-236."""


class Box:
    def __init__(self, low, high, dtype=np.float32):
        self.low = np.asarray(low, dtype)
        self.high = np.asarray(high, dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)

    def contains(self, a):
        a = np.asarray(a)
        return a.shape[-len(self.shape):] == self.shape and bool(np.all(a >= self.low) and np.all(a <= self.high))

    def sample(self, rng, n=None):
        size = self.shape if n is None else (n,) + self.shape
        return rng.uniform(self.low, self.high, size).astype(self.dtype)


class VecParkingEnv:
    """n_envs independent ParkingEnv scenes stepped by one t2d_step per call.

    step(actions[n_envs, 2]) -> (obs[n_envs, 6], reward[n_envs], terminated[n_envs], truncated[n_envs], infos)
    The action layout is the reference's: (steering, accel) (envs/parking.py:239)."""

    _max_steer = MAX_STEER
    _max_accel = MAX_ACCEL
    _discrete_actions = {1: (0, 0), 2: (-0.5, 0), 3: (0.5, 0), 4: (0, 1), 5: (0, -1)}  # parking.py:95

    def __init__(self, n_envs, max_step=int(2e4), continuous=True, auto_reset=False, seed=0, device_id=0,
                 scene_source="layout", type_proportion=0.5, info_lidar=True, copy=True, zero_copy=None, lidar_beams=360):
        """scene_source: "generator" = the device-side ParkingLotGenerator (tactics2d_amd.generator; bay and
        parallel scenes with the reference's rejection sampler, `type_proportion` as in envs/parking.py:331-333),
        "layout" = the fixed bay layout of scenarios.parking (BASELINE config 2).
        The host path (`step`) is ONE library call per step (t2d_step_host): actions up, step + 360-beam scan + pack, one
        frame back (pool.HostFrame).  info_lidar=False leaves the scan out of the host path (info["lidar"] is None: at 4096
        envs the 360 floats per env are 5.9 MB per step over PCIe).  The arrays handed out are views of pinned host frames:
        copy=True (default) fills a frame nobody holds a view of any more -- as good as fresh arrays, without a memcpy (a
        caller that keeps many steps' results by reference gets real copies once the four frames are held); copy=False
        takes the frames in turn (valid for three further steps); copy="always" hands out arrays that OWN their memory (one
        memcpy of the frame per step: nothing the caller holds is ever touched again, whatever it keeps and however).
        lidar_beams: beams of the scan in info["lidar"] / step_torch()["lidar"], a divisor of the reference's 360
        (envs/parking.py:303-304): the scan is then exactly `full_scan[:, ::360 // lidar_beams]` -- the tutorial policy keeps
        every third beam (docs/tutorial/train_parking_demo.ipynb), i.e. lidar_beams=120 moves a third of the bytes.  zero_copy: the kernels read the actions from / write
        the frame to mapped host memory instead of copy commands (None = for pools of at most 16384 envs; beyond, the
        8 B per env of the actions would cross PCIe inside the step kernel)."""
        if scene_source not in ("layout", "generator"):
            raise ValueError(f"unknown scene_source {scene_source!r}")
        self.scene_source = scene_source
        self.type_proportion = type_proportion
        self.device_id = device_id
        self.n_envs = int(n_envs)
        self.max_step = max_step
        self.continuous = continuous
        self.auto_reset = auto_reset
        self.info_lidar = bool(info_lidar)
        if copy not in (True, False, "always"):
            raise ValueError('copy must be True, False or "always"')
        self.copy = copy is True
        self.copy_always = copy == "always"
        self.lidar_beams = int(lidar_beams)
        if self.lidar_beams < 1 or 360 % self.lidar_beams:
            raise ValueError("lidar_beams must divide 360 (a regular subset of the reference's scan)")
        self.zero_copy = self.n_envs <= 16384 if zero_copy is None else bool(zero_copy)
        self.observation_space = Box(np.full(6, -np.inf), np.full(6, np.inf))
        self.action_space = Box([-self._max_steer, -self._max_accel], [self._max_steer, self._max_accel])
        lo, hi = self.action_space.low, self.action_space.high
        self._action_box = np.float32([lo[0], hi[0], lo[1], hi[1]]) if continuous else None
        # ScenarioManager(max_step, step_size=100, ...)  envs/parking.py:144-146
        self.scenario_manager = BatchedScenarioManager(self.n_envs, 1, max_step, 100, device_id=device_id)
        self._seed = seed
        self._scene = None

    # ------------------------------------------------------------------ reset
    def reset(self, seed=None, options=None):
        """New scenes for every env (envs/parking.py:397-405: map_.reset(), map_generator.generate(map_),
        agent.reset(start_state))."""
        if seed is not None:
            self._seed = int(seed)
        m = self.scenario_manager
        if self.scene_source == "generator":
            # generate + install in one launch on the device; with auto_reset every finished episode continues in a
            # NEW scene (the reference's reset() per episode), not in a copy of the first one
            from .participant import VEHICLE_TEMPLATE, vehicle_model
            size = VEHICLE_TEMPLATE["medium_car"][:2]
            ego = vehicle_model("medium_car", "kinematics", speed_range=(-0.5, 0.5), accel_range=(-2.0, 2.0),
                                steer_range=(-0.524, 0.524))
            rows = ego.param_row(L.SHAPE_OBB, *size)[None]
            m.configure(rows, check_dynamic=False, check_off_lane=False, check_arrival=1, check_no_action=1,
                        no_action_max_step=100, shaped_reward=1)
            m.pool.parking_scenes(self._seed, self.type_proportion, size, regenerate=self.auto_reset)
            self._generated = m.pool.get_parking_scenes()
            bad = self._generated.info & 0x1e
            if bad.any():
                raise RuntimeError(f"{int((bad != 0).sum())} generated scenes are flagged; use another seed")
            self._scene = self._generated.scene(max_step=self.max_step)
        else:
            sc = scenarios.parking(self.n_envs, seed0=self._seed * self.n_envs)
            self._scene = sc
            m.pool.set_target_areas(sc.target)
            m.pool.set_target_headings(sc.target_heading)
            m.configure(sc.rows, check_dynamic=False, check_off_lane=False, check_arrival=1, check_no_action=1,
                        no_action_max_step=100, shaped_reward=1)
            m.status_checklist["collision"].reset(_csr_to_lists(sc.static))
            m.status_checklist["out_bound"].reset(sc.boundary)
            m.reset(sc.x, sc.y, sc.heading, sc.speed, sc.type_id, sc.active)
        m.pool.bind_actions(None, None)   # a step_torch binding does not outlive the episode set-up
        m.pool.set_auto_reset(self.auto_reset)
        # SingleLineLidar(perception_range=20, freq_detect=360 * 10)  envs/parking.py:303-304,422-431
        m.pool.lidar_config(self.lidar_beams, 20.0, include_participants=False, subsample_of=360)
        # the target area changes under the caller only when scenes are regenerated on the device: the frame carries it then
        self._moving_targets = self.scene_source == "generator" and self.auto_reset
        m.pool.frame_config(lidar=self.info_lidar, target=self._moving_targets, zero_copy=self.zero_copy)
        self._target_area, self._target_heading = self._scene.target, self._scene.target_heading
        fr = m.pool.frame_fetch(fresh=self.copy)
        fr = self._last = fr.copy() if self.copy_always else fr
        return fr.obs, self._infos(fr)

    @property
    def generated(self):
        """The generated scenes the envs are in NOW (a generator.ParkingScenes + .episode): fetched from the device on demand
        when scenes are regenerated there (nothing on the step path reads it: the frame carries the target areas)."""
        if getattr(self, "_moving_targets", False):
            self._generated = self.scenario_manager.pool.get_parking_scenes()
        return self._generated

    # ------------------------------------------------------------------ step
    def _to_continuous(self, actions):
        if self.continuous:
            # (`action_space.contains` runs inside t2d_step_host, in the pass that stages the actions: 50 us of numpy at 4096 envs)
            try:
                return np.ascontiguousarray(actions, np.float32).reshape(self.n_envs, 2)
            except (ValueError, TypeError):
                raise InvalidAction(f"Action {actions} is not in the action space.") from None
        idx = np.asarray(actions).reshape(self.n_envs)
        if not np.all(np.isin(idx, list(self._discrete_actions))):
            raise InvalidAction(f"Action {actions} is not in the action space.")
        return np.array([self._discrete_actions[int(i)] for i in idx], np.float32)

    def step(self, actions):
        """(obs[n, 6], reward[n], terminated[n], truncated[n], infos) -- envs/parking.py:219-256 for every env: one
        t2d_step_host call (physics_model.step(state, accel, steering) parking.py:355 + check_status + scan + pack)."""
        if self._scene is None:
            raise RuntimeError("call reset() first")
        a = self._to_continuous(actions)
        try:
            fr = self.scenario_manager.pool.step_host(a, 100, action_box=self._action_box, fresh=self.copy)
            fr = self._last = fr.copy() if self.copy_always else fr
        except T2DError as exc:
            if exc.code == ERR_ACTION:
                raise InvalidAction(f"Action {actions} is not in the action space.") from None
            raise
        self.scenario_manager._flags_cache = None
        return fr.obs, fr.reward, fr.terminated, fr.truncated, self._infos(fr)

    def step_torch(self, actions, stream=None):
        """The device-resident step: `actions` is a float32 CUDA tensor [n_envs, 2] in the reference's layout (steering,
        accel); nothing is copied to the host and nothing synchronises.  Returns a dict of torch tensors that are
        ZERO-COPY VIEWS of the pool (valid until the next step): state [6 x n_envs] columns (vx, vy as written by the ego's
        SingleTrackKinematics -- a dynamics / drift ego leaves those two fields alone, include/t2d.h), reward, status (u8 [n, 4]:
        scenario, traffic, terminated, truncated), iou, and `lidar` [n_envs, lidar_beams] written by the scan kernel straight
        into a tensor owned by this env -- the observation buffer handed back to the policy.  Out-of-range actions are
        the caller's responsibility here (the numpy `step` raises InvalidAction like the reference)."""
        import torch
        if self._scene is None:
            raise RuntimeError("call reset() first")
        pool = self.scenario_manager.pool
        dev = actions.device
        cur = torch.cuda.current_stream(dev)
        st = stream if stream is not None else cur
        if st != cur:
            st.wait_stream(cur)   # `actions` was produced on the caller's current stream; the step reads it in place
        with torch.cuda.stream(st):
            # the [n, 2] (steering, accel) tensor is read in place: accel = column 1, steering = column 0, stride 2
            # (two copy kernels per step otherwise: 4.7 us of a 36 us vector step); kept alive until the next step
            if actions.dtype != torch.float32 or tuple(actions.shape) != (self.n_envs, 2):
                raise ValueError(f"actions must be float32 [{self.n_envs}, 2]")
            self._act = actions if actions.is_contiguous() else actions.contiguous()
            base = self._act.data_ptr()
            pool.bind_actions(base + 4, base, stride=2)
            pool.step(100, st.cuda_stream)
            if getattr(self, "_t_lidar", None) is None or self._t_lidar.device != dev:
                self._t_lidar = torch.empty((self.n_envs, self.lidar_beams), dtype=torch.float32, device=dev)
                view = lambda f: torch.as_tensor(pool.device_array(f), device=dev)
                self._t_views = dict(x=view(L.F_X), y=view(L.F_Y), heading=view(L.F_HEADING), speed=view(L.F_SPEED),
                                     vx=view(L.F_VX), vy=view(L.F_VY), reward=view(L.F_REWARD), status=view(L.F_STATUS),
                                     iou=view(L.F_IOU))
            pool.lidar_scan(self._t_lidar.data_ptr(), st.cuda_stream)
        out = dict(self._t_views)
        out["lidar"] = self._t_lidar
        return out

    def _infos(self, fr):
        """_get_infos (envs/parking.py:203-217) for every env, as views of the frame."""
        obs, st = fr.obs, fr.status
        ta, th = (fr.target, fr.target_heading) if self._moving_targets else (self._target_area, self._target_heading)
        return dict(state=dict(x=obs[:, 0], y=obs[:, 1], heading=obs[:, 2], speed=obs[:, 3], vx=obs[:, 4], vy=obs[:, 5],
                               frame=fr.frame_ms),
                    scenario_status=st[:, 0], traffic_status=st[:, 1],
                    target_area=ta, target_heading=th,
                    diff_position=fr.rel[:, 0], diff_angle=fr.rel[:, 1], diff_heading=fr.rel[:, 2],
                    iou=fr.iou, lidar=fr.lidar, episode=fr.episode)

    def _targets(self):
        """Target areas / headings of the scenes the envs are in NOW (as of the last step / reset)."""
        if self._moving_targets:   # (the env keeps the HostFrame object, not views of it: the frame stays free to be refilled)
            return self._last.target.copy(), self._last.target_heading.copy()
        return self._target_area, self._target_heading

    def render(self):
        raise NotImplementedError("rendering is outside the accelerated path")

    def close(self):
        """Frees the pool -- and with it the pinned frames: arrays handed out by reset() / step() are views of that memory
        (DESIGN.md 5a), so copy what has to outlive the env before closing it."""
        self.scenario_manager.close()


class ParkingEnv:
    """Single-scene adapter with the reference's 5-tuple (envs/parking.py:256)."""

    def __init__(self, type_proportion=0.5, render_mode="rgb_array", render_fps=60, max_step=int(2e4),
                 continuous=True, seed=0, scene_source="layout", info_lidar=True, zero_copy=True):
        if render_mode not in ("human", "rgb_array"):
            raise NotImplementedError(f"Render mode {render_mode} is not supported.")  # parking.py:119-120
        self.max_step = max_step
        self.continuous = continuous
        self._vec = VecParkingEnv(1, max_step, continuous, seed=seed, scene_source=scene_source,
                                  type_proportion=type_proportion, info_lidar=info_lidar, copy=False, zero_copy=zero_copy)
        self.observation_space = self._vec.observation_space
        self.action_space = self._vec.action_space
        self.scenario_manager = self._vec.scenario_manager

    def reset(self, seed=None, options=None):
        obs, infos = self._vec.reset(seed, options)
        self._abuf = np.zeros((1, 2), np.float32)
        infos = _first(infos)
        infos["scenario_status"] = ScenarioStatus(int(infos["scenario_status"]))
        infos["traffic_status"] = TrafficStatus(int(infos["traffic_status"]))
        # (what reset hands out must not alias the pinned frame the steps fill: copies, like the step's own results)
        obs = obs.copy()
        infos = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in infos.items()}
        # the step's fast path: everything a step needs of the library and of frame 0, looked up once (one env: the results
        # are copied out of the frame anyway, so every step fills the same pinned frame)
        pool = self.scenario_manager.pool
        fr = pool._frames[0]                  # (built by the reset's frame_fetch: the first frame handed out)
        self._vec._last = fr
        self._fast = (pool._lib.t2d_step_host, pool._h, self._abuf.ctypes.data,
                      None if self._vec._action_box is None else self._vec._action_box.ctypes.data, C.byref(pool._frame_ptr), pool,
                      fr, fr.obs[0], fr.status[0], fr.rel[0], None if fr.lidar is None else fr.lidar[0])
        return obs[0], infos

    def step(self, action):
        """envs/parking.py:219-256: (observation, reward, terminated, truncated, infos) as host values -- one library call
        (t2d_step_host on a zero-copy frame: the kernels read the action from and write the results to mapped host memory).
        `infos["state"]` holds Python floats (the stored fp32 values, exactly), the statuses are the reference's enums."""
        v = self._vec
        if v._scene is None:
            raise RuntimeError("call reset() first")
        a = self._abuf
        if self.continuous:
            try:
                a[0, 0], a[0, 1] = action   # (exactly two scalars: Box.contains checks the shape as well)
            except (ValueError, TypeError):
                raise InvalidAction(f"Action {action} is not in the action space.") from None
        else:
            try:
                a[0] = v._discrete_actions[int(action)]
            except (KeyError, ValueError, TypeError):
                raise InvalidAction(f"Action {action} is not in the action space.") from None
        step_host, h, a_ptr, box_ptr, frame_ref, pool, fr, obs_row, st_row, rel_row, lidar_row = self._fast
        # (`action_space.contains`: checked by the library while it stages the action -- a NaN is outside, like Box.contains)
        rc = step_host(h, a_ptr, box_ptr, 100, None, 0, frame_ref)
        if rc:
            if rc == ERR_ACTION:
                raise InvalidAction(f"Action {action} is not in the action space.")
            pool._ck(rc)
        v.scenario_manager._flags_cache = None
        o = obs_row.copy()
        x, y, heading, speed, vx, vy = o.tolist()
        s0, s1, s2, s3 = st_row.tolist()
        d0, d1, d2 = rel_row.tolist()
        ta, th = (fr.target[0].copy(), float(fr.target_heading[0])) if v._moving_targets else (v._target_area[0], v._target_heading[0])
        infos = {"state": {"x": x, "y": y, "heading": heading, "speed": speed, "vx": vx, "vy": vy, "frame": int(fr.frame_ms[0])},
                 "scenario_status": _SCENARIO[s0], "traffic_status": _TRAFFIC[s1], "target_area": ta, "target_heading": th,
                 "diff_position": d0, "diff_angle": d1, "diff_heading": d2, "iou": float(fr.iou[0]),
                 "lidar": None if lidar_row is None else lidar_row.copy(), "episode": int(fr.episode[0])}
        return o, float(fr.reward[0]), s2 != 0, s3 != 0, infos

    def close(self):
        self._vec.close()


def _first(infos):
    out = {}
    for k, v in infos.items():
        if isinstance(v, dict):
            out[k] = {kk: (vv[0] if vv is not None else None) for kk, vv in v.items()}
        else:
            out[k] = v[0] if isinstance(v, np.ndarray) else v
    return out


def _csr_to_lists(csr):
    eo, vo, xy = csr
    return [[xy[vo[p]:vo[p + 1]] for p in range(eo[e], eo[e + 1])] for e in range(len(eo) - 1)]
