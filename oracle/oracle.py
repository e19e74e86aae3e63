"""ctypes front-end of oracle/libt2d_oracle.so (see t2d_oracle.c for the parity status).

TEST INFRASTRUCTURE ONLY -- the checker, never the thing measured or shipped.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libt2d_oracle.so")
_lib = None

NCOL = 24

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


class StatusConfig(C.Structure):
    """Mirror of t2d_status_config (include/t2d.h)."""
    _fields_ = [("max_step", C.c_int32), ("ego_index", C.c_int32), ("check_dynamic", C.c_int32),
                ("check_off_lane", C.c_int32), ("reward_collision", C.c_float),
                ("reward_time_exceed", C.c_float), ("reward_out_bound", C.c_float),
                ("reward_completed", C.c_float), ("time_penalty_scale", C.c_float),
                ("check_arrival", C.c_int32), ("check_no_action", C.c_int32),
                ("no_action_max_step", C.c_int32), ("shaped_reward", C.c_int32),
                ("arrival_threshold", C.c_float), ("no_action_iou", C.c_float),
                ("dist_reward_scale", C.c_float)]


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "t2d_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libt2d_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.t2do_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.t2do_integrate.argtypes = [_f64p, C.c_int, C.c_int] + [C.c_void_p] * 8 + \
            [_u8p, C.c_void_p, C.c_int, _f64p]
        _lib.t2do_pose_obb.argtypes = [C.c_double] * 5 + [C.c_int, _f64p]
        _lib.t2do_convex_intersects.argtypes = [_f64p, C.c_int, _f64p, C.c_int]
        _lib.t2do_point_in_convex.argtypes = [_f64p, C.c_int, _f64p]
        _lib.t2do_circle_convex_intersects.argtypes = [_f64p, C.c_double, _f64p, C.c_int]
        _lib.t2do_circle_circle_intersects.argtypes = [_f64p, C.c_double, _f64p, C.c_double]
        _lib.t2do_polygon_is_convex.argtypes = [_f32p, C.c_int]
        _lib.t2do_pose_in_lane_union.argtypes = [_f64p, _f64p, _i32p, _f32p, C.c_int, C.c_int]
        _lib.t2do_circle_in_lane_union.argtypes = [_f64p, C.c_double, _i32p, _f32p, C.c_int, C.c_int]
        _lib.t2do_lane_boundary.argtypes = [_i32p, _f32p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib.t2do_collide.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _u8p,
                                      _u8p] + [C.c_void_p] * 8 + [C.c_int, _u32p, _u32p]
        _lib.t2do_status.argtypes = [C.POINTER(StatusConfig), C.c_int, C.c_int, _u32p, C.c_int,
                                     _i32p, _i32p, _u8p, _f32p]
        _lib.t2do_pointmass_euler.argtypes = [_f64p] + [C.POINTER(C.c_double)] * 5 + \
            [C.c_double, C.c_double, C.c_int]
    return _lib


def _ptr(a, dtype):
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(C.c_void_p)


def sincos(x):
    s, c = C.c_double(), C.c_double()
    lib().t2do_sincos(float(x), C.byref(s), C.byref(c))
    return s.value, c.value


def integrate(rows, x, y, heading, speed, vx, vy, act0, act1, type_id, active, interval_ms):
    """fp64 results (n, 8): x, y, heading, speed, vx, vy, applied0, applied1."""
    rows = np.ascontiguousarray(rows, np.float64)
    n = len(x)
    keep, ptrs = [], []
    for a in (x, y, heading, speed, vx, vy, act0, act1):
        if a is None:
            a = np.zeros(n, np.float32)
        arr, p = _ptr(a, np.float32)
        keep.append(arr); ptrs.append(p)
    tid = np.ascontiguousarray(type_id, np.uint8)
    act, actp = _ptr(active, np.uint8)
    out = np.empty((n, 8), np.float64)
    lib().t2do_integrate(rows, rows.shape[1], n, *ptrs, tid, actp, int(interval_ms), out)
    return out


def pose_obb(x, y, h, L, W, trig=0):
    v = np.empty(8, np.float64)
    lib().t2do_pose_obb(float(x), float(y), float(h), float(L), float(W), int(trig), v)
    return v.reshape(4, 2)


def convex_intersects(A, B):
    A = np.ascontiguousarray(A, np.float64); B = np.ascontiguousarray(B, np.float64)
    return bool(lib().t2do_convex_intersects(A, len(A), B, len(B)))


def point_in_convex(P, pt):
    P = np.ascontiguousarray(P, np.float64)
    return bool(lib().t2do_point_in_convex(P, len(P), np.ascontiguousarray(pt, np.float64)))


def circle_convex_intersects(c, R, P):
    P = np.ascontiguousarray(P, np.float64)
    return bool(lib().t2do_circle_convex_intersects(np.ascontiguousarray(c, np.float64), float(R), P, len(P)))


def circle_circle_intersects(c1, R1, c2, R2):
    return bool(lib().t2do_circle_circle_intersects(np.ascontiguousarray(c1, np.float64), float(R1),
                                                    np.ascontiguousarray(c2, np.float64), float(R2)))


def polygon_is_convex(verts):
    v = np.ascontiguousarray(verts, np.float32)
    return bool(lib().t2do_polygon_is_convex(v, len(v)))


def _lane_csr(lanes):
    vo = np.zeros(len(lanes) + 1, np.int32)
    vo[1:] = np.cumsum([len(q) for q in lanes])
    xy = np.ascontiguousarray(np.concatenate([np.asarray(q, np.float32).reshape(-1, 2) for q in lanes]), np.float32)
    return vo, xy


def lane_boundary(lanes):
    """Boundary pieces of the union of the convex lane polygons (t2do_lane_boundary): (pieces[k, 4] = Ax, Ay, Bx, By
    in fp64, owner[k] = index of the lane polygon the piece is a part of an edge of)."""
    vo, xy = _lane_csr(lanes)
    n = lib().t2do_lane_boundary(vo, xy, 0, len(lanes), None, None, 0)
    pieces = np.zeros((max(n, 1), 4), np.float64); owner = np.zeros(max(n, 1), np.int32)
    lib().t2do_lane_boundary(vo, xy, 0, len(lanes), pieces.ctypes.data_as(C.c_void_p), owner.ctypes.data_as(C.c_void_p), n)
    return pieces[:n], owner[:n]


def pose_in_lane_union(pose, centre, lanes):
    """`union(lanes).contains(pose)` (build-defined off-lane, t2d_oracle.c): pose = 4 x (x, y) in the vertex order
    of pose_obb, centre = (x, y), lanes = list of convex polygons (fp32 vertices, either winding)."""
    vo, xy = _lane_csr(lanes)
    return bool(lib().t2do_pose_in_lane_union(np.ascontiguousarray(pose, np.float64).reshape(8),
                                              np.ascontiguousarray(centre, np.float64), vo, xy, 0, len(lanes)))


def circle_in_lane_union(c, R, lanes):
    vo, xy = _lane_csr(lanes)
    return bool(lib().t2do_circle_in_lane_union(np.ascontiguousarray(c, np.float64), float(R), vo, xy, 0, len(lanes)))


def collide(rows, n_env, A, x, y, heading, type_id, active, static=None, boundary=None,
            boundary_valid=None, lanes=None, trig=0):
    """static / lanes: (env_off[E+1], vert_off[P+1], verts_xy[V,2]) CSR tuples or None.
    Returns (flags[N], env_flags[E])."""
    rows = np.ascontiguousarray(rows, np.float64)
    keep = []

    def csr(t):
        if t is None:
            return [None, None, None]
        eo, vo, xy = t
        out = []
        for a, dt in ((eo, np.int32), (vo, np.int32), (xy, np.float32)):
            arr, p = _ptr(a, dt); keep.append(arr); out.append(p)
        return out

    sp = csr(static)
    b, bp = _ptr(boundary, np.float32)
    bv, bvp = _ptr(boundary_valid, np.uint8)
    lp = csr(lanes)
    N = n_env * A
    flags = np.zeros(N, np.uint32); env_flags = np.zeros(n_env, np.uint32)
    lib().t2do_collide(rows, rows.shape[1], n_env, A, np.ascontiguousarray(x, np.float32),
                       np.ascontiguousarray(y, np.float32), np.ascontiguousarray(heading, np.float32),
                       np.ascontiguousarray(type_id, np.uint8), np.ascontiguousarray(active, np.uint8),
                       sp[0], sp[1], sp[2], bp, bvp, lp[0], lp[1], lp[2], int(trig), flags, env_flags)
    return flags, env_flags


def status(cfg, n_env, A, flags, interval_ms, cnt_step, frame_ms):
    """Advance cnt_step / frame_ms in place; returns (status[E,4] u8, reward[E] f32)."""
    st = np.zeros((n_env, 4), np.uint8); rw = np.zeros(n_env, np.float32)
    lib().t2do_status(C.byref(cfg), n_env, A, np.ascontiguousarray(flags, np.uint32),
                      int(interval_ms), cnt_step, frame_ms, st.reshape(-1), rw)
    return st, rw


def pointmass_euler(row, x, y, heading, vx, vy, ax, ay, interval):
    vals = [C.c_double(v) for v in (x, y, heading, vx, vy)]
    lib().t2do_pointmass_euler(np.ascontiguousarray(row, np.float64), *[C.byref(v) for v in vals],
                               float(ax), float(ay), int(interval))
    return [v.value for v in vals]


def set_trig(mode):
    """0 = libm (reference-faithful, default), 1 = deterministic t2d trig (bit-reproducible)."""
    lib().t2do_set_trig(int(mode))


def atan_det(x):
    f = lib().t2do_atan
    f.restype = C.c_double; f.argtypes = [C.c_double]
    return f(float(x))


def set_threads(n):
    """Host threads for the batch loops (integrate / collide / status).  Results are independent of it."""
    lib().t2do_set_threads(int(n))


def has_openmp():
    return bool(lib().t2do_has_openmp())


def make_config(**kw):
    """t2d_status_config with the ParkingEnv defaults; keyword overrides."""
    cfg = StatusConfig(20000, 0, 0, 0, -5.0, -1.0, -5.0, 5.0, 0.001, 0, 0, 100, 0, 0.95, 0.999, 0.1)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError(f"unknown status option {k}")
        setattr(cfg, k, v)
    return cfg


def quad_iou(A, B):
    f = lib().t2do_quad_iou
    f.restype = C.c_double
    f.argtypes = [_f64p, _f64p]
    return f(np.ascontiguousarray(A, np.float64).reshape(-1), np.ascontiguousarray(B, np.float64).reshape(-1))


def ccw(q):
    q = np.asarray(q, np.float64).reshape(4, 2)
    a2 = sum(q[i, 0] * q[(i + 1) % 4, 1] - q[(i + 1) % 4, 0] * q[i, 1] for i in range(4))
    return q if a2 > 0 else q[::-1].copy()


class EpisodeState:
    """Per-env detector / shaping state of the status epilogue (NoAction, _max_iou, _min_dist_to_target)."""

    def __init__(self, n_env, target=None, centroid=None, start_xy=None):
        self.n_env = n_env
        self.target = None
        self.target_c = None
        if target is not None:
            self.target = np.stack([ccw(np.float32(t)) for t in target]).reshape(n_env, 8)
            if centroid is None:
                c = []
                for t in self.target.reshape(n_env, 4, 2):
                    w = t[:, 0] * np.roll(t[:, 1], -1) - np.roll(t[:, 0], -1) * t[:, 1]
                    a2 = w.sum()
                    c.append([((t[:, 0] + np.roll(t[:, 0], -1)) * w).sum() / (3 * a2),
                              ((t[:, 1] + np.roll(t[:, 1], -1)) * w).sum() / (3 * a2)])
                centroid = np.array(c)
            self.target_c = np.ascontiguousarray(centroid, np.float64)
        self.last_pose = np.zeros((n_env, 8)); self.last_valid = np.zeros(n_env, np.uint8)
        self.cnt_na = np.zeros(n_env, np.int32); self.max_iou = np.full(n_env, -np.inf)
        self.min_dist = np.full(n_env, np.inf)
        if self.target is not None and start_xy is not None:
            self.reset_envs(np.ones(n_env, bool), start_xy)
        self.start_min_dist = self.min_dist.copy()

    def reset_envs(self, mask, start_xy=None):
        self.last_valid[mask] = 0; self.cnt_na[mask] = 0; self.max_iou[mask] = -np.inf
        if start_xy is not None and self.target is not None:
            d = np.float64(np.float32(start_xy)) - self.target_c
            self.min_dist[mask] = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2)[mask]
        elif hasattr(self, "start_min_dist"):
            self.min_dist[mask] = self.start_min_dist[mask]


def status_ex(cfg, A, flags, interval_ms, cnt_step, frame_ms, rows, x, y, heading, type_id, ep):
    """Extended status step for the ego of every env (IoU events + shaped reward); mutates cnt_step,
    frame_ms and `ep` (EpisodeState).  Returns (status[E,4], reward[E], iou[E])."""
    n_env = ep.n_env
    ego = cfg.ego_index
    rows = np.ascontiguousarray(rows, np.float64)
    pose = np.zeros((n_env, 8)); is_obb = np.zeros(n_env, np.uint8); xy = np.zeros((n_env, 2))
    g = lib().t2do_ego_poses
    g.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _u8p, C.c_int, _f64p, _f64p, _u8p]
    g(rows, rows.shape[1], n_env, A, int(ego), np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32),
      np.ascontiguousarray(heading, np.float32), np.ascontiguousarray(type_id, np.uint8), 0,
      pose.reshape(-1), xy.reshape(-1), is_obb)
    st = np.zeros((n_env, 4), np.uint8); rw = np.zeros(n_env, np.float32); iou = np.zeros(n_env, np.float32)
    f = lib().t2do_status_ex
    f.argtypes = [C.POINTER(StatusConfig), C.c_int, C.c_int, _u32p, C.c_int, _i32p, _i32p, _u8p, _f32p,
                  _f64p, _f64p, _u8p, C.c_void_p, C.c_void_p, C.c_int, _f64p, _u8p, _i32p, _f64p, _f64p, _f32p]
    tgt = ep.target.ctypes.data_as(C.c_void_p) if ep.target is not None else None
    tc = ep.target_c.ctypes.data_as(C.c_void_p) if ep.target is not None else None
    f(C.byref(cfg), n_env, A, np.ascontiguousarray(flags, np.uint32), int(interval_ms), cnt_step, frame_ms,
      st.reshape(-1), rw, pose.reshape(-1), xy.reshape(-1), is_obb, tgt, tc, int(ep.target is not None),
      ep.last_pose.reshape(-1), ep.last_valid, ep.cnt_na, ep.max_iou, ep.min_dist, iou)
    return st, rw, iou


def idm_accel(params, v, has_lead, dx=0.0, dy=0.0, v_lead=0.0, trig=0):
    """One IDMController.step -> clipped acceleration.  trig=0: libm pow/hypot (pins vs the golden
    vectors); trig=1: the deterministic spec shared with the GPU."""
    f = lib().t2do_idm_accel
    f.restype = C.c_double
    f.argtypes = [_f64p, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double]
    lib().t2do_set_trig(trig)
    try:
        return f(np.ascontiguousarray(params, np.float64), float(v), int(has_lead), float(dx), float(dy), float(v_lead))
    finally:
        lib().t2do_set_trig(0)


def det_pow(x, y):
    f = lib().t2do_pow; f.restype = C.c_double; f.argtypes = [C.c_double, C.c_double]
    return f(float(x), float(y))


def det_exp(x):
    f = lib().t2do_exp; f.restype = C.c_double; f.argtypes = [C.c_double]
    return f(float(x))


def det_log(x):
    f = lib().t2do_log; f.restype = C.c_double; f.argtypes = [C.c_double]
    return f(float(x))


def idm(ctrl_rows, ctrl_id, n_env, A, x, y, heading, speed, active, act0, act1, forced_leader=None, trig=1):
    """Batched IDM with the build-defined leader rule.  Returns (act0, act1, leader) -- act0/act1 are
    copies of the inputs with the controlled participants' entries replaced."""
    rows = np.ascontiguousarray(ctrl_rows, np.float64)
    a0 = np.array(act0, np.float32).reshape(-1).copy(); a1 = np.array(act1, np.float32).reshape(-1).copy()
    lead = np.full(n_env * A, -1, np.int32)
    fl = None if forced_leader is None else np.ascontiguousarray(forced_leader, np.int32)
    f = lib().t2do_idm
    f.restype = None
    f.argtypes = [_f64p, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, _u8p, C.c_void_p,
                  _f32p, _f32p, _i32p]
    lib().t2do_set_trig(trig)
    try:
        f(rows, rows.shape[1], rows.shape[0], np.ascontiguousarray(ctrl_id, np.uint8), n_env, A,
          np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32),
          np.ascontiguousarray(heading, np.float32), np.ascontiguousarray(speed, np.float32),
          np.ascontiguousarray(active, np.uint8), None if fl is None else fl.ctypes.data_as(C.c_void_p), a0, a1, lead)
    finally:
        lib().t2do_set_trig(0)
    return a0, a1, lead


def verify_state(rows, type_id, last, cand, interval_ms, trig=0):
    """Batched verify_state: last (n, 6) = x, y, heading, speed, vx, vy; cand (n, 4) = x, y, heading, speed
    (both rounded to fp32, as the pool stores them).  Returns bool[n]."""
    rows = np.ascontiguousarray(rows, np.float64)
    n = len(type_id)
    lc = [np.ascontiguousarray(np.asarray(last)[:, k], np.float32) for k in range(6)]
    cc = [np.ascontiguousarray(np.asarray(cand)[:, k], np.float32) for k in range(4)]
    out = np.zeros(n, np.uint8)
    f = lib().t2do_verify_batch
    f.restype = None
    f.argtypes = [_f64p, C.c_int, C.c_int, _u8p] + [_f32p] * 10 + [C.c_int, _u8p]
    lib().t2do_set_trig(trig)
    try:
        f(rows, rows.shape[1], n, np.ascontiguousarray(type_id, np.uint8), *lc, *cc, int(interval_ms), out)
    finally:
        lib().t2do_set_trig(0)
    return out.astype(bool)


def drift(rows, type_id, state, action, interval_ms, active=None, trig=0):
    """Batched SingleTrackDrift.step: state (n, 6) = x, y, heading, speed, omega_wf, omega_wr (fp32-rounded),
    action (n, 2).  Returns fp64 (n, 8): x, y, heading, speed, omega_wf, omega_wr, applied accel, applied steer."""
    rows = np.ascontiguousarray(rows, np.float64)
    n = len(type_id)
    st = [np.ascontiguousarray(np.asarray(state)[:, k], np.float32) for k in range(6)]
    ac = [np.ascontiguousarray(np.asarray(action)[:, k], np.float32) for k in range(2)]
    out = np.zeros((n, 8))
    f = lib().t2do_drift_batch
    f.restype = None
    f.argtypes = [_f64p, C.c_int, C.c_int] + [_f32p] * 8 + [_u8p, C.c_void_p, C.c_int, _f64p]
    act = None if active is None else np.ascontiguousarray(active, np.uint8)
    lib().t2do_set_trig(trig)
    try:
        f(rows, rows.shape[1], n, *st, *ac, np.ascontiguousarray(type_id, np.uint8),
          None if act is None else act.ctypes.data_as(C.c_void_p), int(interval_ms), out.reshape(-1))
    finally:
        lib().t2do_set_trig(0)
    return out


def beam_tables(n_beams):
    th = np.linspace(0, 2 * np.pi, n_beams, endpoint=False)
    return np.ascontiguousarray(np.sin(th)), np.ascontiguousarray(np.cos(th))


def lidar(rows, n_env, A, ego_index, x, y, heading, type_id, active, static, include_participants,
          n_beams, max_range, trig=0):
    """SingleLineLidar scan of every env's ego -> float32 [n_env, n_beams] (+inf = no return)."""
    rows = np.ascontiguousarray(rows, np.float64)
    keep = []
    if static is None:
        sp = [None, None, None]
    else:
        sp = []
        for a, dt in zip(static, (np.int32, np.int32, np.float32)):
            arr, p = _ptr(a, dt); keep.append(arr); sp.append(p)
    bs, bc = beam_tables(n_beams)
    out = np.empty((n_env, n_beams), np.float32)
    f = lib().t2do_lidar
    f.argtypes = [_f64p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, _f32p, _f32p, _u8p, _u8p, C.c_void_p,
                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, _f64p, _f64p, C.c_int, _f32p]
    f(rows, rows.shape[1], n_env, A, ego_index, np.ascontiguousarray(x, np.float32),
      np.ascontiguousarray(y, np.float32), np.ascontiguousarray(heading, np.float32),
      np.ascontiguousarray(type_id, np.uint8), np.ascontiguousarray(active, np.uint8), sp[0], sp[1], sp[2],
      int(include_participants), int(n_beams), float(max_range), bs, bc, int(trig), out.reshape(-1))
    return out


GEN_MAX_QUADS = 12
GEN_BAY, GEN_UNVERIFIED, GEN_START_UNVERIFIED, GEN_NONCONVEX, GEN_OVERFLOW = 1, 2, 4, 8, 16
GEN_START_FLIPPED, GEN_TARGET_FLIPPED = 32, 64


def generate_parking(seed, n_env, type_proportion=0.5, vehicle_size=(5.3, 2.5), first_env=0, trig=1):
    """ParkingLotGenerator.generate restated for n_env scenes (row f4; pinned by replay, see t2d_oracle.c).
    Returns a dict of arrays: quads (n, 12, 4, 2) f32, quad_id (n, 12), n_quads (n,), start (n, 3) f64,
    target (n, 4, 2) f32, target_heading (n,) f64, boundary (n, 4) f32, info (n,) u32."""
    out = dict(quads=np.zeros((n_env, GEN_MAX_QUADS, 4, 2), np.float32), quad_id=np.zeros((n_env, GEN_MAX_QUADS), np.int32),
               n_quads=np.zeros(n_env, np.int32), start=np.zeros((n_env, 3)), target=np.zeros((n_env, 4, 2), np.float32),
               target_heading=np.zeros(n_env), boundary=np.zeros((n_env, 4), np.float32), info=np.zeros(n_env, np.uint32))
    f = lib().t2do_generate_parking
    f.restype = None
    f.argtypes = [C.c_uint64, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, _f32p,
                  np.ctypeslib.ndpointer(np.int32, flags="C"), np.ctypeslib.ndpointer(np.int32, flags="C"), _f64p, _f32p,
                  _f64p, _f32p, np.ctypeslib.ndpointer(np.uint32, flags="C")]
    lib().t2do_set_trig(trig)
    try:
        f(int(seed), int(first_env), int(n_env), float(type_proportion), float(vehicle_size[0]), float(vehicle_size[1]),
          out["quads"], out["quad_id"], out["n_quads"], out["start"], out["target"], out["target_heading"],
          out["boundary"], out["info"])
    finally:
        lib().t2do_set_trig(0)
    return out


def generate_parking_replay(tape_kind, tape_val, type_proportion=0.5, vehicle_size=(5.3, 2.5), trig=1):
    """ONE scene of the restated generator on a tape of recorded draws (oracle/gen_golden_generator.py: the values numpy handed
    the reference's own generate()).  Returns (scene dict as generate_parking, draws consumed, index of the first draw whose kind
    did not match the tape's -- or that ran past its end --, -1 = none)."""
    out = dict(quads=np.zeros((1, GEN_MAX_QUADS, 4, 2), np.float32), quad_id=np.zeros((1, GEN_MAX_QUADS), np.int32),
               n_quads=np.zeros(1, np.int32), start=np.zeros((1, 3)), target=np.zeros((1, 4, 2), np.float32),
               target_heading=np.zeros(1), boundary=np.zeros((1, 4), np.float32), info=np.zeros(1, np.uint32))
    kind = np.ascontiguousarray(tape_kind, np.int32); val = np.ascontiguousarray(tape_val, np.float64)
    desync = np.zeros(1, np.int32)
    f = lib().t2do_generate_parking_replay
    f.restype = C.c_int
    f.argtypes = [_f64p, np.ctypeslib.ndpointer(np.int32, flags="C"), C.c_int, C.c_double, C.c_double, C.c_double, _f32p,
                  np.ctypeslib.ndpointer(np.int32, flags="C"), np.ctypeslib.ndpointer(np.int32, flags="C"), _f64p, _f32p,
                  _f64p, _f32p, np.ctypeslib.ndpointer(np.uint32, flags="C"), np.ctypeslib.ndpointer(np.int32, flags="C")]
    lib().t2do_set_trig(trig)
    try:
        used = f(val, kind, int(kind.size), float(type_proportion), float(vehicle_size[0]), float(vehicle_size[1]),
                 out["quads"], out["quad_id"], out["n_quads"], out["start"], out["target"], out["target_heading"],
                 out["boundary"], out["info"], desync)
    finally:
        lib().t2do_set_trig(0)
    return out, int(used), int(desync[0])
