"""Batched counterparts of tactics2d.traffic: status enums, event detectors, ScenarioManager.

Mirrors (tactics2d v0.1.9rc3):
    ScenarioStatus / TrafficStatus          traffic/status.py:10-61
    EventBase.update / reset                traffic/event_detection/event_base.py:10-19
    StaticCollision / DynamicCollision      traffic/event_detection/collision.py:12-46
    OutBound                                traffic/event_detection/out_bound.py:12-65
    OffLane                                 traffic/event_detection/off_lane.py:10-20
    TimeExceed                              traffic/event_detection/time_exceed.py:10-36
    ScenarioManager                         traffic/scenario_manager.py:13-98
    _ParkingScenarioManager.update/check_status/reset   envs/parking.py:352-441

One `BatchedScenarioManager` owns one ParticipantPool = n_env independent scenes.  `update()` is one
t2d_integrate launch for every participant of every scene, `check_status()` one t2d_collide launch
(flags + the ordered status logic); `step()` runs both (t2d_step).  The detectors are views on the
flags of the last launch -- there is no per-participant Python loop and no CPU path.
"""
from enum import IntEnum

import numpy as np

from . import layout as L
from .pool import ParticipantPool


class ScenarioStatus(IntEnum):
    """traffic/status.py:10-30"""
    NORMAL = 1
    COMPLETED = 2
    TIME_EXCEEDED = 3
    OUT_BOUND = 4
    NO_ACTION = 5
    FAILED = 6


class TrafficStatus(IntEnum):
    """traffic/status.py:33-61"""
    NORMAL = 1
    UNKNOWN = 2
    COLLISION_STATIC = 3
    COLLISION_DYNAMIC = 4
    OFF_ROUTE = 5
    OFF_LANE = 6
    VIOLATION_RETROGRADE = 7
    VIOLATION_NON_DRIVABLE = 8
    VIOLATION_TRAFFIC_LIGHT = 9
    VIOLATION_TRAFFIC_SIGN = 10


def polygons_to_csr(per_env_polygons):
    """[[poly(n,2), ...] per env] -> (env_offsets, vertex_offsets, verts_xy) for the C ABI."""
    eo, vo, xy = [0], [0], []
    for polys in per_env_polygons:
        for q in polys:
            q = np.asarray(q, np.float32).reshape(-1, 2)
            xy.append(q); vo.append(vo[-1] + len(q))
        eo.append(eo[-1] + len(polys))
    xy = np.concatenate(xy) if xy else np.zeros((0, 2), np.float32)
    return np.array(eo, np.int32), np.array(vo, np.int32), xy


class _FlagDetector:
    """EventBase over one bit of the pool's event flags (evaluated by the collide kernel)."""
    bit = 0

    def __init__(self, manager=None):
        self._m = manager

    def bind(self, manager):
        self._m = manager

    def update(self, ego_only=True):
        """Reference: update(agent_pose) -> bool.  Batched: bool[n_env] for the ego of every scene
        (ego_only) or bool[n_env, max_agents] for every participant."""
        f = self._m.flags()
        hit = (f & self.bit) != 0
        return hit[:, self._m.ego_index] if ego_only else hit


class StaticCollision(_FlagDetector):
    bit = L.FLAG_COLLISION_STATIC

    def __init__(self, static_objects=None, manager=None):
        super().__init__(manager)
        self.static_objects = static_objects

    def reset(self, static_objects=None):
        """static_objects: per-env lists of convex polygons (array (n, 2), 3..8 vertices)."""
        self.static_objects = static_objects
        if self._m is not None:
            self._m._static = None if static_objects is None else polygons_to_csr(static_objects)
            self._m._push_geometry()


class DynamicCollision(_FlagDetector):
    bit = L.FLAG_COLLISION_DYNAMIC

    def reset(self):
        return


class OutBound(_FlagDetector):
    bit = L.FLAG_OUT_BOUND

    def __init__(self, boundary=None, manager=None):
        super().__init__(manager)
        self.map_boundary = boundary

    def reset(self, boundary=None):
        """boundary: (n_env, 4) xmin, xmax, ymin, ymax (out_bound.py:20-35) or None (= never out)."""
        self.map_boundary = boundary
        if self._m is not None:
            self._m._boundary = None if boundary is None else np.asarray(boundary, np.float32).reshape(-1, 4)
            self._m._push_geometry()


class OffLane(_FlagDetector):
    """The reference detector is a stub that returns False (off_lane.py:16-17).  Here it becomes real once lanes
    are supplied: `not union(lane polygons).contains(pose)` -- the predicate of OutBound (out_bound.py:37-48) applied
    to the lanes (build-defined; lanes that abut must share their vertices exactly or overlap)."""
    bit = L.FLAG_OFF_LANE

    def __init__(self, manager=None):
        super().__init__(manager)
        self.lanes = None

    def reset(self, lanes):
        self.lanes = lanes
        if self._m is not None:
            self._m._lanes = None if lanes is None else polygons_to_csr(lanes)
            self._m.pool.set_lane_geometry(self._m._lanes)


class TimeExceed:
    """time_exceed.py:10-36; the counter lives in the pool (cnt_step per scene)."""

    def __init__(self, max_step, manager=None):
        self.max_step = max_step
        self._m = manager

    def bind(self, manager):
        self._m = manager

    def update(self):
        return self._m.pool.download(L.F_CNT_STEP) > self.max_step

    def reset(self):
        return


class BatchedScenarioManager:
    """ScenarioManager (traffic/scenario_manager.py:13-98) for n_env scenes at once."""

    def __init__(self, n_env, max_agents=1, max_step=None, step_size=None, render_fps=60, off_screen=True,
                 device_id=0):
        self.render_fps = render_fps
        self.off_screen = off_screen
        self.max_step = max_step
        self.step_size = int(step_size) if step_size is not None else int(1000 / render_fps)
        self.n_env, self.max_agents = n_env, max_agents
        self.ego_index = 0
        self.pool = ParticipantPool(n_env, max_agents, device_id)
        self.render_manager = None
        self._static = self._lanes = self._boundary = None
        self.status_checklist = {
            "time_exceed": TimeExceed(max_step, self),
            "out_bound": OutBound(manager=self),
            "collision": StaticCollision(manager=self),
            "dynamic_collision": DynamicCollision(self),
            "off_lane": OffLane(self),
        }
        self._flags_cache = None

    # -- state views ----------------------------------------------------------------------------
    @property
    def cnt_step(self):
        return self.pool.download(L.F_CNT_STEP)

    @property
    def scenario_status(self):
        return self.pool.download(L.F_STATUS)[:, 0]

    @property
    def traffic_status(self):
        return self.pool.download(L.F_STATUS)[:, 1]

    def flags(self):
        if self._flags_cache is None:
            self._flags_cache = self.pool.download(L.F_FLAGS).reshape(self.n_env, self.max_agents)
        return self._flags_cache

    def _push_geometry(self):
        self.pool.set_static_geometry(self._static, self._boundary)

    # -- ScenarioManager interface -----------------------------------------------------------------
    def configure(self, rows, check_dynamic=False, check_off_lane=False, **options):
        self.pool.set_param_table(rows)
        self.pool.set_status_config(max_step=self.max_step if self.max_step is not None else 0,
                                    ego_index=self.ego_index, check_dynamic=int(check_dynamic),
                                    check_off_lane=int(check_off_lane), **options)

    def reset(self, x, y, heading, speed, type_id, active=None, env_mask=None):
        """Load the start states (Trajectory reset + detector resets, envs/parking.py:397-441)."""
        self.pool.reset(x, y, heading, speed, type_id, active, env_mask=env_mask)
        self.pool.snapshot()
        self._flags_cache = None
        self._ego_velocity_derived = None   # (an env's ego model is fixed until the next reset: see get_observation)

    def update(self, act0, act1, stream=None):
        """Physics step of every participant (parking.py:352-359: `physics_model.step` + add_state)."""
        self.pool.set_actions(act0, act1)
        self.pool.integrate(self.step_size, stream)
        self._flags_cache = None

    def check_status(self, stream=None):
        """Ordered event checks (parking.py:361-392) -> (scenario_status[E], traffic_status[E])."""
        self.pool.check_status(self.step_size, stream)
        self._flags_cache = None
        st = self.pool.download(L.F_STATUS)
        return st[:, 0], st[:, 1]

    def step(self, act0=None, act1=None, stream=None):
        """update + check_status in one t2d_step (two launches, stream ordered)."""
        if act0 is not None:
            self.pool.set_actions(act0, act1)
        self.pool.step(self.step_size, stream)
        self._flags_cache = None

    def step_host(self, actions, lidar=False, fresh=True):
        """update + check_status for every scene, host to host, in ONE library call (t2d_step_host): `actions` float32
        [n_env * max_agents, 2] in the reference's action layout (steering, accel) -- (ay, ax) for a point mass -- in, a
        pool.HostFrame out: the ego's state, reward, status bytes, IoU, frame, counters (and the 360-beam scan with lidar=True,
        after lidar_config) of every scene as numpy views of one pinned frame, instead of one blocking copy per field."""
        want = (bool(lidar),)
        if getattr(self, "_frame_cfg", None) != want:
            self.pool.frame_config(lidar=bool(lidar))
            self._frame_cfg = want
        a = np.ascontiguousarray(actions, np.float32).reshape(self.n_env * self.max_agents, 2)
        self._flags_cache = None
        return self.pool.step_host(a, self.step_size, fresh=fresh)

    def render(self):
        raise NotImplementedError("rendering is outside the accelerated path (DESIGN.md section 9)")

    def get_active_participants(self, frame=None):
        """Per scene, the indices of the active participants (scenario_manager.py:83-94)."""
        ids = self.pool.download(L.F_IDS).reshape(self.n_env, self.max_agents)
        act = (ids >> 16) & 0xff
        return [np.nonzero(a)[0].tolist() for a in act]

    def get_observation(self):
        """State observation of the ego of every scene: float32 [n_env, 6] = x, y, heading, speed, vx, vy
        (the reference returns a rendered camera image, scenario_manager.py:96-98: not on this path)."""
        cols = [self.pool.download(f).reshape(self.n_env, self.max_agents)[:, self.ego_index]
                for f in (L.F_X, L.F_Y, L.F_HEADING, L.F_SPEED, L.F_VX, L.F_VY)]
        # SingleTrackDynamics / SingleTrackDrift return a State without vx, vy (single_track_dynamics.py:220-227) and
        # the pool's VX / VY fields are not written by those models (include/t2d.h): State.velocity then DERIVES
        # (speed cos(heading), speed sin(heading)) (state.py:152-169) -- done here the same way, lazily
        # (the mask is read once per reset, not per call: an ego's model does not change in between)
        if getattr(self, "_ego_velocity_derived", None) is None:
            model = self.pool.download(L.F_IDS).reshape(self.n_env, self.max_agents)[:, self.ego_index] & 0xff
            self._ego_velocity_derived = (model == L.MODEL_DYNAMICS) | (model == L.MODEL_DRIFT)
        derived = self._ego_velocity_derived
        if derived.any():
            h, v = cols[2].astype(np.float64), cols[3].astype(np.float64)
            cols[4] = np.where(derived, (v * np.cos(h)).astype(np.float32), cols[4])
            cols[5] = np.where(derived, (v * np.sin(h)).astype(np.float32), cols[5])
        return np.stack(cols, 1)

    def close(self):
        self.pool.close()
