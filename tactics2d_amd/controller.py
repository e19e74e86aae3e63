"""Host-side mirror of the reference's IDM controller for the batched device path (scope row f3).

Reference: `tactics2d.controller.IDMController` (controller/idm_controller.py:15-157): same
constructor arguments and defaults, `configure(**kwargs)` with the same AttributeError, and `step`
returning `(steering, acceleration)` with steering always 0.0.  Here one controller object is one
parameter set; `install(pool, controllers, ctrl_id)` hands the sets to the device, after which
`pool.step()` computes the controlled participants' accelerations on the GPU before integrating.
There is no CPU path: `step` runs the HIP kernel on a scratch pool.
"""
import numpy as np

from . import layout as L

_PARAMS = ("desired_speed", "time_headway", "min_spacing", "max_acceleration", "comfortable_deceleration", "delta")


class IDMController:
    def __init__(self, desired_speed=10.0, time_headway=1.5, min_spacing=2.0, max_acceleration=1.0,
                 comfortable_deceleration=3.0, delta=4.0, lane_half_width=1.875, horizon=np.inf):
        # reference ctor: idm_controller.py:33-57.  lane_half_width / horizon belong to the
        # build-defined leader rule (include/t2d.h, t2d_set_idm); the reference has no such rule.
        self.desired_speed = desired_speed
        self.time_headway = time_headway
        self.min_spacing = min_spacing
        self.max_acceleration = max_acceleration
        self.comfortable_deceleration = comfortable_deceleration
        self.delta = delta
        self.lane_half_width = lane_half_width
        self.horizon = horizon

    def configure(self, **kwargs):
        """idm_controller.py:143-157; re-install on the pool afterwards."""
        for key, value in kwargs.items():
            if hasattr(self, key):
                setattr(self, key, value)
            else:
                raise AttributeError(f"IDMController has no parameter '{key}'")

    def row(self):
        r = np.zeros(L.IDM_COLS)
        for k, name in enumerate(_PARAMS):
            r[k] = float(getattr(self, name))
        r[L.IDM_LANE_HALF_WIDTH] = float(self.lane_half_width)
        r[L.IDM_HORIZON] = float(self.horizon)
        return r

    def step(self, ego_state, leading_state=None, **kwargs):
        """Batched `IDMController.step(ego_state, leading_state)`: states are `physics.BatchedState`s of
        equal length (leading_state=None: free flow).  Returns (steering[n] zeros, acceleration[n])."""
        from .pool import ParticipantPool
        n = len(np.atleast_1d(ego_state.x))
        two = leading_state is not None
        A = 2 if two else 1
        z = np.zeros(n, np.float32)

        def col(name):
            e = np.asarray(getattr(ego_state, name), np.float32).reshape(n)
            if not two:
                return e
            l = np.asarray(getattr(leading_state, name), np.float32).reshape(n)
            return np.stack([e, l], 1).reshape(-1)

        pool = ParticipantPool(n, A)
        try:
            row = np.zeros((1, L.PARAM_COLS)); row[0, L.P_DELTA_T_MS] = 5.0  # a kinematic dummy type: never integrated
            row[0, L.P_LF] = row[0, L.P_LR] = 1.0; row[0, L.P_WB] = 2.0; row[0, L.P_LENGTH] = 4.0; row[0, L.P_WIDTH] = 2.0
            pool.set_param_table(row)
            hd = col("heading") if getattr(ego_state, "heading", None) is not None else np.zeros(n * A, np.float32)
            pool.reset(col("x"), col("y"), hd, col("speed"), np.zeros(n * A, np.uint8))
            cid = np.full(n * A, L.IDM_NONE, np.uint8); cid[::A] = 0
            pool.set_idm(self.row()[None], cid)
            forced = np.full(n * A, L.IDM_LEADER_FREE, np.int32)
            if two:
                forced[::A] = 1
            pool.upload(L.F_LEADER, forced)           # reuse the field as the device-side leader list
            ptr, _ = pool.field_ptr(L.F_LEADER)
            pool.idm_actions(ptr)
            acc = pool.download(L.F_ACT0)[::A].astype(np.float64)
        finally:
            pool.close()
        return z.astype(np.float64), acc


def install(pool, controllers, ctrl_id):
    """controllers: sequence of IDMController; ctrl_id[n]: index into it or layout.IDM_NONE."""
    rows = np.stack([c.row() for c in controllers]) if len(controllers) else None
    pool.set_idm(rows, ctrl_id)
