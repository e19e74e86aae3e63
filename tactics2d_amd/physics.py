"""Batched counterparts of tactics2d.physics with the reference's constructor semantics.

Mirrors (tactics2d v0.1.9rc3):
    SingleTrackKinematics   physics/single_track_kinematics.py:62-124 (ctor), :178-198 (step)
    SingleTrackDynamics     physics/single_track_dynamics.py:58-138 (ctor), :231-251 (step)
    PointMass               physics/point_mass.py:33-81 (ctor), :209-232 (step)

The constructors normalise ranges / delta_t exactly like the reference (including its quirks: an
`int` range means "unbounded" for the vehicle models, PointMass clamps negative bounds to 0) and
produce one row of the kernels' parameter table.  `step()` takes a BatchedState and arrays of
actions and runs ONE HIP launch for all of them (t2d_integrate); there is no per-participant
Python loop and no CPU fallback.
"""
import logging

import numpy as np

from . import layout as L

_DELTA_T = 5       # PhysicsModelBase._DELTA_T   physics/physics_model_base.py:23
_MIN_DELTA_T = 1   # PhysicsModelBase._MIN_DELTA_T


def _vehicle_range(r):
    """single_track_kinematics.py:87-115: float r -> [-r, r] (None if r < 0); 2-sequence kept if
    lo < hi; anything else (including an int) -> None."""
    if isinstance(r, float):
        return None if r < 0 else [-r, r]
    if hasattr(r, "__len__") and len(r) == 2:
        return None if r[0] >= r[1] else r
    return None


def _pointmass_range(r):
    """point_mass.py:50-66: float r -> [0, r]; 2-sequence -> [max(0, lo), max(0, hi)], None if lo >= hi."""
    if isinstance(r, float):
        return None if r < 0 else [0, r]
    if hasattr(r, "__len__") and len(r) == 2:
        out = [max(0, r[0]), max(0, r[1])]
        return None if out[0] >= out[1] else out
    return None


def _resolve_delta_t(delta_t, interval):
    """single_track_kinematics.py:119-124."""
    if delta_t is None:
        return _DELTA_T
    dt = max(delta_t, _MIN_DELTA_T)
    if interval is not None:
        dt = min(dt, interval)
    return dt


class BatchedState:
    """SoA stand-in for a batch of `State` objects (participant/trajectory/state.py:12-223): the same typed record, the
    same lazily derived fields, one float32 column per field (what the pool stores) and one `frame` (ms) for the batch.

    * typed `__setattr__` (:108-126): `frame` is coerced to int, every other annotated field to a float column (a
      scalar fills the batch); None stays None; a value that cannot be converted raises ValueError with the
      reference's message;
    * `speed` (:135-150) = the scalar that was set, else ||(vx, vy)||, else None;
    * `velocity` (:152-169) = (vx, vy) when set, else (speed cos(heading), speed sin(heading)), else None;
    * `accel` (:171-186) = ||(ax, ay)|| when set, else the NORM of `acceleration` -- for a state that only carries the
      scalar (what the physics models return) that is ||accel (cos h, sin h)|| = |accel| up to rounding
      (0.9999999999999999 for accel = 1.0, heading = 0.3), not the signed scalar itself, which stays in `_accel`;
    * `acceleration` (:188-204) = (ax, ay) when set, else (accel cos h, accel sin h), else None.
    Derived values are fp64 arrays computed from the stored fp32 columns with the reference's expressions.
    """

    __annotations__ = {"frame": int, "x": float, "y": float, "heading": float, "vx": float, "vy": float, "_speed": float,
                       "ax": float, "ay": float, "_accel": float}

    def __init__(self, frame, x=0, y=0, heading=0, vx=None, vy=None, speed=None, ax=None, ay=None, accel=None):
        object.__setattr__(self, "_n", max(np.size(x), np.size(y), 1))
        self.frame = frame
        self.x = x
        self.y = y
        self.heading = heading
        self.vx = vx
        self.vy = vy
        self._speed = speed
        self.ax = ax
        self.ay = ay
        self._accel = accel

    def __setattr__(self, name, value):
        kind = self.__annotations__.get(name)
        if kind is None or value is None:
            object.__setattr__(self, name, value)
            return
        try:
            if kind is int:
                value = value if isinstance(value, int) else int(value)
            else:
                col = np.asarray(value, dtype=np.float32)
                if col.ndim > 1 or col.size not in (1, self._n):
                    raise TypeError("not a column of the batch")
                value = np.ascontiguousarray(np.broadcast_to(col.reshape(-1), (self._n,)))
        except Exception:
            raise ValueError(f"Failed to convert {value} to the expected type of {name}: ({kind}).") from None
        object.__setattr__(self, name, value)

    def __len__(self):
        return self._n

    def __str__(self):
        return (f"{self.__class__.__name__}(frame={self.frame}, x={self.x}, y={self.y}, heading={self.heading}, "
                f"vx={self.vx}, vy={self.vy}, speed={self.speed}, ax={self.ax}, ay={self.ay}, accel={self.accel})")

    @property
    def location(self):
        return self.x, self.y

    @property
    def speed(self):
        if self._speed is not None:
            return self._speed
        if self.vx is not None and self.vy is not None:
            vx, vy = self.vx.astype(np.float64), self.vy.astype(np.float64)
            return np.sqrt(vx * vx + vy * vy)                   # np.linalg.norm([vx, vy])
        return None

    @property
    def velocity(self):
        if self.vx is not None and self.vy is not None:
            return self.vx, self.vy
        v = self.speed
        if v is not None and self.heading is not None:
            h, v = self.heading.astype(np.float64), np.asarray(v, np.float64)
            return v * np.cos(h), v * np.sin(h)
        return None

    @property
    def acceleration(self):
        if self.ax is not None and self.ay is not None:
            return self.ax, self.ay
        if self._accel is not None and self.heading is not None:
            h, a = self.heading.astype(np.float64), self._accel.astype(np.float64)
            return a * np.cos(h), a * np.sin(h)
        return None

    @property
    def accel(self):
        acc = self.acceleration
        if acc is None:
            return None
        ax, ay = np.asarray(acc[0], np.float64), np.asarray(acc[1], np.float64)
        return np.sqrt(ax * ax + ay * ay)                       # np.linalg.norm(...)

    def set_heading(self, heading):
        self.heading = heading

    def set_velocity(self, vx, vy):
        self.vx = vx
        self.vy = vy

    def set_speed(self, speed):
        self._speed = speed

    def set_accel(self, ax, ay):
        """sets ax, ay and the scalar ||(ax, ay)|| (state.py:218-223)"""
        self.ax = ax
        self.ay = ay
        a, b = self.ax.astype(np.float64), self.ay.astype(np.float64)
        self._accel = np.sqrt(a * a + b * b)


class _BatchedModel:
    model_id = None

    def _pool(self, n):
        from .pool import ParticipantPool
        pool = getattr(self, "_cached_pool", None)
        if pool is None or pool.n_env != n:
            if pool is not None:
                pool.close()
            pool = ParticipantPool(n, 1)
            pool.set_param_table(self.param_row()[None])
            self._cached_pool = pool
        return pool

    def param_row(self, shape=L.SHAPE_OBB, length=0.0, width=0.0):
        raise NotImplementedError

    def verify_state(self, state, last_state, interval=None):
        """Batched `verify_state(state, last_state, interval)` -> bool[n] (single_track_kinematics.py:200-250,
        single_track_dynamics.py:253-306, point_mass.py:234-259), evaluated on the device."""
        interval = state.frame - last_state.frame if interval is None else interval
        n = len(state)
        pool = self._pool(n)
        z = np.zeros(n, np.float32)
        if self.model_id == L.MODEL_POINTMASS:
            vx, vy = last_state.velocity
            pool.reset(last_state.x, last_state.y, z, z, np.zeros(n, np.uint8), vx=vx, vy=vy)
            return pool.verify_state(state.x, state.y, z, z, interval)
        pool.reset(last_state.x, last_state.y, last_state.heading, last_state.speed, np.zeros(n, np.uint8))
        return pool.verify_state(state.x, state.y, state.heading, state.speed, interval)

    def close(self):
        pool = getattr(self, "_cached_pool", None)
        if pool is not None:
            pool.close()
            self._cached_pool = None


class SingleTrackKinematics(_BatchedModel):
    model_id = L.MODEL_KINEMATICS

    def __init__(self, lf, lr, steer_range=None, speed_range=None, accel_range=None, interval=100,
                 delta_t=None):
        self.lf = lf
        self.lr = lr
        self.wheel_base = lf + lr
        self.steer_range = _vehicle_range(steer_range)
        self.speed_range = _vehicle_range(speed_range)
        self.accel_range = _vehicle_range(accel_range)
        self.interval = interval
        self.delta_t = _resolve_delta_t(delta_t, interval)

    def param_row(self, shape=L.SHAPE_OBB, length=0.0, width=0.0):
        r = np.zeros(L.PARAM_COLS)
        r[L.P_MODEL] = self.model_id
        r[L.P_LF], r[L.P_LR], r[L.P_WB] = self.lf, self.lr, self.wheel_base
        flags = 0
        if self.steer_range is not None:
            r[L.P_STEER_LO], r[L.P_STEER_HI] = self.steer_range
            flags |= L.RANGE_STEER
        if self.speed_range is not None:
            r[L.P_SPEED_LO], r[L.P_SPEED_HI] = self.speed_range
            flags |= L.RANGE_SPEED
        if self.accel_range is not None:
            r[L.P_ACCEL_LO], r[L.P_ACCEL_HI] = self.accel_range
            flags |= L.RANGE_ACCEL
        r[L.P_RANGE_FLAGS] = flags
        r[L.P_DELTA_T_MS] = self.delta_t
        r[L.P_SHAPE], r[L.P_LENGTH], r[L.P_WIDTH] = shape, length, width
        return r

    def step(self, state, accel, delta, interval=None):
        """Batched `step`: returns (next_state, applied_accel, applied_delta) like the reference."""
        interval = interval if interval is not None else self.interval
        n = len(state)
        pool = self._pool(n)
        z = np.zeros(n, np.uint8)
        pool.reset(state.x, state.y, state.heading, state.speed, z)
        pool.set_actions(np.broadcast_to(np.asarray(accel, np.float32), (n,)),
                         np.broadcast_to(np.asarray(delta, np.float32), (n,)))
        pool.integrate(interval)
        d = pool.download
        app0, app1 = d(L.F_APPLIED0), d(L.F_APPLIED1)
        has_v = self.model_id != L.MODEL_DYNAMICS  # dynamics State has vx = vy = None
        nxt = BatchedState(state.frame + interval, d(L.F_X), d(L.F_Y), d(L.F_HEADING),
                           d(L.F_VX) if has_v else None, d(L.F_VY) if has_v else None,
                           speed=d(L.F_SPEED), accel=app0)
        return nxt, app0, app1


class SingleTrackDynamics(SingleTrackKinematics):
    model_id = L.MODEL_DYNAMICS

    def __init__(self, lf, lr, mass, mass_height, mu=0.7, I_z=1500, cf=20.89, cr=20.89,
                 steer_range=None, speed_range=None, accel_range=None, interval=100, delta_t=None):
        super().__init__(lf, lr, steer_range, speed_range, accel_range, interval, delta_t)
        self.mass, self.mass_height = mass, mass_height
        self.mu, self.I_z, self.cf, self.cr = mu, I_z, cf, cr

    def param_row(self, shape=L.SHAPE_OBB, length=0.0, width=0.0):
        r = super().param_row(shape, length, width)
        r[L.P_MASS], r[L.P_MASS_HEIGHT] = self.mass, self.mass_height
        r[L.P_MU], r[L.P_IZ], r[L.P_CF], r[L.P_CR] = self.mu, self.I_z, self.cf, self.cr
        return r


class SingleTrackDrift(SingleTrackKinematics):
    """Batched `tactics2d.physics.SingleTrackDrift` (single_track_drift.py:52-503): same constructor
    (the built-in `Tire` constants only) and the five-tuple `step` return."""
    model_id = L.MODEL_DRIFT

    def __init__(self, lf, lr, mass, mass_height, radius=0.344, T_sb=0.76, T_se=1, tire=None, I_z=1500, I_yw=1.7,
                 steer_range=None, speed_range=None, accel_range=None, interval=100, delta_t=None):
        if tire is not None:
            raise NotImplementedError("only the reference's built-in Tire constants are implemented on the device")
        super().__init__(lf, lr, steer_range, speed_range, accel_range, interval, delta_t)
        self.mass, self.mass_height, self.radius = mass, mass_height, radius
        self.T_sb, self.T_se, self.I_z, self.I_yw = T_sb, T_se, I_z, I_yw

    def param_row(self, shape=L.SHAPE_OBB, length=0.0, width=0.0):
        r = super().param_row(shape, length, width)
        r[L.P_MASS], r[L.P_MASS_HEIGHT], r[L.P_IZ] = self.mass, self.mass_height, self.I_z
        r[L.P_DRIFT_TSB], r[L.P_DRIFT_TSE], r[L.P_DRIFT_RADIUS], r[L.P_DRIFT_IYW] = self.T_sb, self.T_se, self.radius, self.I_yw
        return r

    def step(self, state, omega_wf, omega_wr, accel, delta, interval=None):
        """-> (next_state, next_omega_wf, next_omega_wr, applied_accel, applied_delta)  (:467-503)"""
        interval = interval if interval is not None else self.interval
        n = len(state)
        pool = self._pool(n)
        bc = lambda a: np.broadcast_to(np.asarray(a, np.float32), (n,))
        pool.reset(state.x, state.y, state.heading, state.speed, np.zeros(n, np.uint8))
        pool.upload(L.F_OMEGA_F, bc(omega_wf)); pool.upload(L.F_OMEGA_R, bc(omega_wr))
        pool.set_actions(bc(accel), bc(delta))
        pool.integrate(interval)
        d = pool.download
        app0, app1 = d(L.F_APPLIED0), d(L.F_APPLIED1)
        nxt = BatchedState(state.frame + interval, d(L.F_X), d(L.F_Y), d(L.F_HEADING), speed=d(L.F_SPEED), accel=app0)
        return nxt, d(L.F_OMEGA_F), d(L.F_OMEGA_R), app0, app1


class PointMass(_BatchedModel):
    model_id = L.MODEL_POINTMASS
    backends = ["newton", "euler"]

    def __init__(self, speed_range=None, accel_range=None, interval=100, delta_t=None, backend="newton"):
        self.speed_range = _pointmass_range(speed_range)
        self.accel_range = _pointmass_range(accel_range)
        self.interval = interval
        self.delta_t = _resolve_delta_t(delta_t, interval)
        if backend not in self.backends:   # point_mass.py:77-81
            logging.warning(f"Unsupported backend {backend}. Using `newton` instead.")
            backend = "newton"
        self.backend = backend
        # (the euler back-end -- point_mass.py:177-207 -- is T2D_MODEL_POINTMASS_EULER on the device: integrated by the side
        # kernel that also takes the drift model, a launch ahead of the step launch)
        self.model_id = L.MODEL_POINTMASS_EULER if backend == "euler" else L.MODEL_POINTMASS

    def param_row(self, shape=L.SHAPE_CIRCLE, length=0.0, width=0.0):
        r = np.zeros(L.PARAM_COLS)
        r[L.P_MODEL] = self.model_id
        flags = 0
        if self.speed_range is not None:
            r[L.P_SPEED_LO], r[L.P_SPEED_HI] = self.speed_range
            flags |= L.RANGE_SPEED
        if self.accel_range is not None:
            r[L.P_ACCEL_LO], r[L.P_ACCEL_HI] = self.accel_range
            flags |= L.RANGE_ACCEL
        r[L.P_RANGE_FLAGS] = flags
        r[L.P_DELTA_T_MS] = self.delta_t
        r[L.P_SHAPE], r[L.P_LENGTH], r[L.P_WIDTH] = shape, length, width
        return r

    def step(self, state, accel, interval=None):
        """Batched `step(state, (ax, ay), interval)` -> next_state (point_mass.py:209-232)."""
        interval = interval if interval is not None else self.interval
        n = len(state)
        pool = self._pool(n)
        vx, vy = state.velocity
        z = np.zeros(n, np.float32)
        # (the euler back-end re-projects a clipped speed onto state.heading, point_mass.py:195-197: heading is state there)
        h = np.asarray(state.heading, np.float32) if self.backend == "euler" else z
        pool.reset(state.x, state.y, h, z, np.zeros(n, np.uint8), vx=vx, vy=vy)
        ax, ay = accel
        pool.set_actions(np.broadcast_to(np.asarray(ax, np.float32), (n,)),
                         np.broadcast_to(np.asarray(ay, np.float32), (n,)))
        pool.integrate(interval)
        d = pool.download
        return BatchedState(state.frame + interval, d(L.F_X), d(L.F_Y), d(L.F_HEADING), d(L.F_VX), d(L.F_VY))
