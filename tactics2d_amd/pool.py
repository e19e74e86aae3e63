"""ParticipantPool -- thin, typed Python face of one t2d_pool handle (include/t2d.h).

Host logic only: argument marshalling and error translation.  All compute happens in the
HIP kernels behind libt2d_hip.so.
"""
import ctypes as C
import sys

import numpy as np

from . import _ffi, layout as L


def _arr(a, dtype, n=None, name="array"):
    if a is None:
        return None
    out = np.ascontiguousarray(a, dtype=dtype)
    if n is not None and out.size != n:
        raise ValueError(f"{name}: expected {n} elements, got {out.size}")
    return out


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _DevArray:
    """Zero-copy view of a pool field for `torch.as_tensor(..., device='cuda')`."""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}
        self._owner = owner


class HostFrame:
    """numpy views of one host frame of t2d_step_host (include/t2d.h): what ParkingEnv.step hands its caller, for the ego of
    every env.  `copy()` returns a frame that owns its memory; `in_use()` tells whether anything outside this object still
    holds a view of the frame (so that the pinned frame can be handed out again without a copy)."""
    _SECTIONS = (("rel", "off_rel", np.float64, 3), ("obs", "off_obs", np.float32, 6), ("reward", "off_reward", np.float32, 0),
                 ("status", "off_status", np.uint8, 4), ("iou", "off_iou", np.float32, 0),
                 ("frame_ms", "off_frame_ms", np.int32, 0), ("cnt_step", "off_cnt_step", np.int32, 0),
                 ("episode", "off_episode", np.int32, 0), ("target_heading", "off_target_heading", np.float64, 0),
                 ("target", "off_target", np.float32, 8), ("lidar", "off_lidar", np.float32, -1))

    def __init__(self, base, lay):
        """base: uint8 array holding the frame (or its front part up to the lidar section)."""
        self.base, self.lay = base, lay
        n = lay.n_env
        self.header = base[:64].view(np.uint32)
        for name, off_name, dt, cols in self._SECTIONS:
            off = getattr(lay, off_name)
            cols = lay.n_beams if cols < 0 else cols
            nb = n * max(cols, 1) * np.dtype(dt).itemsize
            if off < 0 or off + nb > base.size:
                setattr(self, name, None)
                continue
            v = base[off:off + nb].view(dt)
            setattr(self, name, v.reshape(n, 4, 2) if name == "target" else v.reshape(n, cols) if cols else v)
        st = self.status
        self.terminated, self.truncated = st[:, 2].view(np.bool_), st[:, 3].view(np.bool_)
        del st, v
        # everything a caller can get hold of is one of these objects or a view whose .base is one of them (numpy collapses
        # the base chain of a view to the array that exposes the memory): their reference counts tell whether the frame is held
        self._tracked = [self.base] + [v for v in self.__dict__.values() if isinstance(v, np.ndarray) and v is not self.base]
        self._idle_refs = None   # set by calibrate(); until then the frame counts as in use (the safe answer)

    def calibrate(self):
        """Record the reference counts of the frame's arrays in the IDLE state -- call once, right after construction, when
        nothing outside this object holds a view yet (ParticipantPool._frame does).  Measured, not derived: whatever
        temporaries the interpreter keeps while counting are the same then and later."""
        self._idle_refs = self._refs()
        return self

    def _refs(self):
        # the object itself counts too: a caller that keeps the HostFrame (`frames.append(mgr.step_host(a))`) holds no array
        # reference, and must pin the frame all the same.  (calibrate() and in_use() are both called through one local name
        # beside whatever container owns the frame -- ParticipantPool._frame / _pick_frame keep it that way.)
        return [sys.getrefcount(self)] + [sys.getrefcount(v) for v in self._tracked]

    def in_use(self):
        return self._idle_refs is None or self._refs() != self._idle_refs

    def copy(self, lidar=True):
        """A frame that owns its memory (one memcpy); lidar=False leaves the lidar section out (its views become None)."""
        end = self.lay.bytes if lidar or self.lay.off_lidar < 0 else self.lay.off_lidar
        return HostFrame(self.base[:end].copy(), self.lay)


class ParticipantPool:
    """n_env environments x max_agents participants resident on one MI355X."""

    def __init__(self, n_env, max_agents=1, device_id=0, library=None):
        """library: the loaded C library the pool lives in -- libt2d_hip.so unless a test / measurement asks for the hooks of
        libt2d_hip_debug.so (tactics2d_amd.debug.pool)."""
        self._lib = library if library is not None else _ffi.lib()
        self._h = C.c_void_p()
        self.n_env, self.max_agents = int(n_env), int(max_agents)
        self.n = self.n_env * self.max_agents
        self.device_id = int(device_id)
        _ffi.check(self._lib.t2d_create(self.n_env, self.max_agents, self.device_id, C.byref(self._h)), None, self._lib)

    # ---------------------------------------------------------------- lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.t2d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        _ffi.check(rc, self._h, self._lib)

    # ---------------------------------------------------------------- configuration
    def set_param_table(self, rows):
        rows = np.ascontiguousarray(rows, np.float64)
        if rows.ndim != 2 or rows.shape[1] < L.PARAM_COLS:
            raise ValueError("rows must be (n_types, >=24) float64")
        self._ck(self._lib.t2d_set_param_table(self._h, _p(rows), rows.shape[0], rows.shape[1]))
        self.n_types = rows.shape[0]

    @staticmethod
    def _csr(t, n_env):
        if t is None:
            return None, None, None
        eo, vo, xy = t
        eo = _arr(eo, np.int32, n_env + 1, "env offsets")
        vo = _arr(vo, np.int32, None, "vertex offsets")
        xy = _arr(xy, np.float32, None, "verts_xy")
        if vo.size != eo[-1] + 1:
            raise ValueError("vertex offsets must have n_poly + 1 entries")
        if xy.size != 2 * (vo[-1] if vo.size else 0):
            raise ValueError("verts_xy must have 2 * n_vert entries")
        return eo, vo, xy

    def set_static_geometry(self, static=None, boundary=None, boundary_valid=None):
        """static: (env_poly_offsets[E+1], poly_vert_offsets[P+1], verts_xy[V,2]) or None;
        boundary: (E,4) xmin,xmax,ymin,ymax or None."""
        eo, vo, xy = self._csr(static, self.n_env)
        b = _arr(boundary, np.float32, 4 * self.n_env, "boundary")
        bv = _arr(boundary_valid, np.uint8, self.n_env, "boundary_valid")
        self._ck(self._lib.t2d_set_static_geometry(self._h, _p(eo), _p(vo), _p(xy), _p(b), _p(bv)))

    def set_lane_geometry(self, lanes=None):
        eo, vo, xy = self._csr(lanes, self.n_env)
        self._ck(self._lib.t2d_set_lane_geometry(self._h, _p(eo), _p(vo), _p(xy)))

    def set_status_config(self, **kw):
        cfg = _ffi.StatusConfig(20000, 0, 0, 0, -5.0, -1.0, -5.0, 5.0, 0.001, 0, 0, 100, 0, 0.95, 0.999, 0.1)
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown status option {k}")
            setattr(cfg, k, v)
        self._ck(self._lib.t2d_set_status_config(self._h, C.byref(cfg)))
        self.status_config = cfg

    def set_target_areas(self, target_xy=None, centroid=None):
        """target_xy: (n_env, 4, 2) quads or None; centroid: (n_env, 2) or None (computed)."""
        t = _arr(target_xy, np.float32, 8 * self.n_env, "target_xy")
        c = _arr(centroid, np.float32, 2 * self.n_env, "centroid")
        self._ck(self._lib.t2d_set_target_areas(self._h, _p(t), _p(c)))

    def lidar_config(self, n_beams=360, max_range=20.0, include_participants=False, subsample_of=None):
        """SingleLineLidar of every ego; beam tables come from numpy like the reference's linspace/sin/cos.
        subsample_of = N: the n_beams beams are every (N / n_beams)-th beam of the N-beam scan -- the SAME angles, so the scan
        equals `full_scan[:, ::N // n_beams]` bit for bit (what the tutorial policy keeps of ParkingEnv's 360 beams)."""
        if subsample_of is None:
            th = np.linspace(0, 2 * np.pi, int(n_beams), endpoint=False)
        else:
            if int(subsample_of) % int(n_beams):
                raise ValueError(f"{n_beams} beams are not a regular subset of {subsample_of}")
            th = np.linspace(0, 2 * np.pi, int(subsample_of), endpoint=False)[::int(subsample_of) // int(n_beams)]
        bs, bc = np.ascontiguousarray(np.sin(th)), np.ascontiguousarray(np.cos(th))
        self._ck(self._lib.t2d_lidar_config(self._h, int(n_beams), float(max_range), int(bool(include_participants)),
                                            _p(bs), _p(bc)))
        self.n_beams = int(n_beams)

    def lidar_scan(self, out_ptr=None, stream=None):
        self._ck(self._lib.t2d_lidar_scan(self._h, out_ptr, stream))

    def set_idm(self, ctrl_rows, ctrl_id):
        """Install IDM controllers: ctrl_rows [n_ctrl, 8] (layout.IDM_*), ctrl_id [n] uint8 (IDM_NONE =
        action supplied by the caller).  ctrl_rows=None uninstalls."""
        if ctrl_rows is None:
            self._ck(self._lib.t2d_set_idm(self._h, None, 0, 0, None))
            return
        rows = np.ascontiguousarray(ctrl_rows, np.float64)
        if rows.ndim != 2:
            raise ValueError("ctrl_rows must be 2-D [n_ctrl, >= 8]")
        cid = _arr(ctrl_id, np.uint8, self.n, "ctrl_id")
        self._ck(self._lib.t2d_set_idm(self._h, _p(rows), rows.shape[0], rows.shape[1], _p(cid)))

    def idm_actions(self, forced_leader_ptr=None, stream=None):
        """IDMController.step for every controlled participant (also runs inside step()/integrate())."""
        self._ck(self._lib.t2d_idm_actions(self._h, forced_leader_ptr, stream))

    def verify_state_ptr(self, x_ptr, y_ptr, heading_ptr, speed_ptr, interval_ms, valid_ptr, stream=None):
        """verify_state of device-resident candidate columns against the pool's current state."""
        self._ck(self._lib.t2d_verify_state(self._h, x_ptr, y_ptr, heading_ptr, speed_ptr, int(interval_ms),
                                            valid_ptr, stream))

    def verify_state(self, x, y, heading, speed, interval_ms):
        """Host-array convenience: candidate columns are staged in this pool's action / applied-action
        fields (scratch here: the next integrate overwrites them anyway) and the verdict bytes in FLAGS."""
        n = self.n
        saved = [self.download(f) for f in (L.F_ACT0, L.F_ACT1, L.F_APPLIED0, L.F_APPLIED1, L.F_FLAGS)]
        for f, v in zip((L.F_ACT0, L.F_ACT1, L.F_APPLIED0, L.F_APPLIED1), (x, y, heading, speed)):
            self.upload(f, _arr(v, np.float32, n, "candidate"))
        ptr = lambda f: self.field_ptr(f)[0]
        self.verify_state_ptr(ptr(L.F_ACT0), ptr(L.F_ACT1), ptr(L.F_APPLIED0), ptr(L.F_APPLIED1), interval_ms,
                              ptr(L.F_FLAGS))
        out = self.download(L.F_FLAGS).view(np.uint8)[:n].astype(bool)
        for f, v in zip((L.F_ACT0, L.F_ACT1, L.F_APPLIED0, L.F_APPLIED1, L.F_FLAGS), saved):
            self.upload(f, v)
        return out

    def parking_scenes(self, seed, type_proportion=0.5, vehicle_size=(5.3, 2.5), regenerate=False, first_env=0,
                       env_stride=None):
        """Device-side ParkingLotGenerator writing straight into this pool (one participant per env): obstacles,
        boundary, target, start pose, snapshot, IoU state -- nothing crosses PCIe.  regenerate=True: after every
        step, envs whose episode ended get the scene of their next episode (stream first_env + e + k * env_stride) --
        from a ring of 16 lots per env staged ahead on a stream of the pool's own, copied in by the step launch itself
        (single-ego pools) or by one small launch behind it.  An env cannot outrun its ring (an episode lasts at least
        two steps, the ring is topped up every 8); should a slot ever be found unstaged the env keeps its lot and the
        next sync() / download() raises T2DError(ERR_STATE) once (include/t2d.h, t2d_parking_scenes).
        regenerate="inline" generates them on the step's stream instead of staging them ahead (C ABI value 2)."""
        stride = self.n_env if env_stride is None else int(env_stride)
        self._ck(self._lib.t2d_parking_scenes(self._h, int(seed) & (2**64 - 1), int(first_env), stride,
                                              float(type_proportion), float(vehicle_size[0]), float(vehicle_size[1]),
                                              2 if regenerate == "inline" else int(bool(regenerate))))

    def get_parking_scenes(self):
        """The scenes currently installed by parking_scenes(): a generator.ParkingScenes plus `.episode`."""
        from .generator import MAX_QUADS, ParkingScenes
        n = self.n_env
        out = ParkingScenes(np.zeros((n, MAX_QUADS, 4, 2), np.float32), np.zeros((n, MAX_QUADS), np.int32),
                            np.zeros(n, np.int32), np.zeros((n, 3)), np.zeros((n, 4, 2), np.float32), np.zeros(n),
                            np.zeros((n, 4), np.float32), np.zeros(n, np.uint32), None)
        episode = np.zeros(n, np.int32)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        self._ck(self._lib.t2d_get_parking_scenes(self._h, ptr(out.quads), ptr(out.quad_id), ptr(out.n_quads),
                                                  ptr(out.start), ptr(out.target), ptr(out.target_heading),
                                                  ptr(out.boundary), ptr(out.info), ptr(episode)))
        out.episode = episode
        return out

    # ---------------------------------------------------------------- the Gym-API host path
    def frame_config(self, lidar=False, target=False, zero_copy=False, n_frames=4):
        """Sections of the host frame t2d_step_host fills (t2d_frame_config) and the number of pinned host frames; returns
        the layout."""
        lay = _ffi.FrameLayout()
        mask = (L.FRAME_LIDAR if lidar else 0) | (L.FRAME_TARGET if target else 0) | (L.FRAME_ZEROCOPY if zero_copy else 0)
        self._ck(self._lib.t2d_frame_config(self._h, mask, int(n_frames), C.byref(lay)))
        key = (mask, int(n_frames), bytes(lay))
        if getattr(self, "_frame_key", None) == key:
            # the configuration already in place (every env reset asks): the library kept the pinned frames, and so do we --
            # HostFrame objects, their views and what callers still hold of them stay valid and tracked
            return self.frame_layout
        self._frame_key = key
        self.frame_layout = lay
        self.n_frames = int(n_frames)
        self._frames = [None] * self.n_frames
        self._frame_ptr = C.c_void_p()
        self._frame_turn = 0
        return lay

    def host_action_buffer(self):
        """float32 [n, 2] numpy view of the pool's own pinned action staging buffer (t2d_host_action_buffer): actions written
        there and passed to step_host as this very array are not copied again (valid until the next frame_config)."""
        ptr = C.c_void_p()
        self._ck(self._lib.t2d_host_action_buffer(self._h, C.byref(ptr)))
        buf = (C.c_float * (2 * self.n)).from_address(ptr.value)
        return np.frombuffer(buf, np.float32).reshape(self.n, 2)

    def set_target_headings(self, heading):
        h = _arr(heading, np.float64, self.n_env, "target_heading")
        self._ck(self._lib.t2d_set_target_headings(self._h, _p(h)))

    def _pick_frame(self, fresh):
        """Index of the pinned frame the next call fills and whether its contents must be copied out.  fresh=False: the frames
        in turn (views valid until n_frames - 1 further calls).  fresh=True: a frame nobody holds a view of any more -- what the
        caller gets is then as good as a new array, without a copy; should every frame but the last still be held (a caller
        that keeps the results of many steps by reference), the last one is filled and copied out."""
        if not fresh:
            k = self._frame_turn
            self._frame_turn = (k + 1) % self.n_frames
            return k, False
        last = self.n_frames - 1
        for k in range(last):
            fr = self._frames[k]
            if fr is None or not fr.in_use():
                return k, False
        return last, True

    def _frame(self, k):
        fr = self._frames[k]
        if fr is None:   # (the views of a pinned frame are built once)
            buf = (C.c_uint8 * self.frame_layout.bytes).from_address(self._frame_ptr.value)
            arr = np.frombuffer(buf, np.uint8)
            fr = HostFrame(arr, self.frame_layout)
            del arr, buf
            self._frames[k] = fr   # (first: the idle reference counts include the list's reference and the local name's,
            fr.calibrate()         #  exactly what _pick_frame's `fr.in_use()` sees when nobody else holds the frame)
        return fr

    def step_host(self, actions, interval_ms=100, stream=None, action_box=None, fresh=False):
        """ParkingEnv.step for every env, host to host (t2d_step_host): actions float32 [n, 2] (steering, accel) C-contiguous or
        None (the actions already in the pool); returns the HostFrame -- views of pinned memory (see _pick_frame for how long
        they stay valid).  action_box: float32 [4] (steering lo, hi, accel lo, hi) = `action_space.contains` for every row,
        checked by the library while it stages the actions (T2DError with code ERR_ACTION, nothing stepped)."""
        if actions is not None:
            if actions.dtype != np.float32 or actions.size != 2 * self.n or not actions.flags.c_contiguous:
                raise ValueError(f"actions must be C-contiguous float32 [{self.n}, 2]")
            actions = actions.ctypes.data
        if action_box is not None:
            action_box = action_box.ctypes.data
        k, must_copy = self._pick_frame(fresh)
        rc = self._lib.t2d_step_host(self._h, actions, action_box, interval_ms, stream, k, C.byref(self._frame_ptr))
        if rc:
            self._ck(rc)
        fr = self._frame(k)
        return fr.copy() if must_copy else fr

    def frame_fetch(self, stream=None, fresh=False):
        """The frame of the current state without stepping (t2d_frame_fetch)."""
        k, must_copy = self._pick_frame(fresh)
        self._ck(self._lib.t2d_frame_fetch(self._h, stream, k, C.byref(self._frame_ptr)))
        fr = self._frame(k)
        return fr.copy() if must_copy else fr

    def set_integrator_variant(self, variant):
        v = {"exact": 0, "fast": 1, "fast_iterated": 2, "fast_resummed": 3}.get(variant, variant)
        self._ck(self._lib.t2d_set_integrator_variant(self._h, int(v)))

    def set_outputs(self, velocity=True, applied=True):
        """Which pure output columns the integrators store (t2d_set_outputs): vx / vy of the single-track models
        and the applied action.  Both on = the reference's State; a point mass's velocity is state and always stored."""
        self._ck(self._lib.t2d_set_outputs(self._h, (L.OUT_VELOCITY if velocity else 0) | (L.OUT_APPLIED if applied else 0)))

    # ---------------------------------------------------------------- state
    def reset(self, x, y, heading, speed, type_id, active=None, vx=None, vy=None, env_mask=None):
        n = self.n
        a = [_arr(v, np.float32, n, k) for k, v in (("x", x), ("y", y), ("heading", heading), ("speed", speed))]
        vx_ = _arr(vx, np.float32, n, "vx"); vy_ = _arr(vy, np.float32, n, "vy")
        tid = _arr(type_id, np.uint8, n, "type_id")
        act = _arr(np.ones(n, np.uint8) if active is None else active, np.uint8, n, "active")
        mask = _arr(env_mask, np.uint8, self.n_env, "env_mask")
        self._ck(self._lib.t2d_reset(self._h, _p(mask), _p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]),
                                     _p(vx_), _p(vy_), _p(tid), _p(act)))

    def upload(self, field, values):
        dt = np.dtype(L.FIELD_DTYPES[field])
        v = np.ascontiguousarray(values, dt)
        self._ck(self._lib.t2d_upload(self._h, field, _p(v), v.nbytes))

    def download(self, field):
        dt = np.dtype(L.FIELD_DTYPES[field])
        n = self.n_env if field in L.PER_ENV_FIELDS else self.n
        if field == L.F_LIDAR:
            ptr, nb = self.field_ptr(field)
            out = np.empty((n, nb // (4 * n)), dt)
        elif field == L.F_STATUS:
            out = np.empty((n, 4), dt)
        elif field == L.F_RECORD:
            out = np.empty((L.RECORD_RING, n, 2), dt)
        else:
            out = np.empty(n, dt)
        self._ck(self._lib.t2d_download(self._h, field, _p(out), out.nbytes))
        return out

    def set_actions(self, act0, act1):
        self.upload(L.F_ACT0, act0)
        self.upload(L.F_ACT1, act1)

    def bind_actions(self, act0_ptr=None, act1_ptr=None, stride=1, extent=None):
        """Zero-copy actions from caller-owned device memory (raw pointers, e.g. tensor.data_ptr()); participant i reads
        element i * stride of each (a policy's [N, 2] (steering, accel) tensor: act0 = ptr + 4, act1 = ptr, stride = 2).
        extent = elements readable behind each pointer (an action ring of K sets: K * N * stride): `step_n` then refuses a
        fragment that would read past it instead of faulting on the device (t2d_set_action_extent)."""
        if stride == 1:
            self._ck(self._lib.t2d_bind_actions(self._h, act0_ptr, act1_ptr))
        else:
            self._ck(self._lib.t2d_bind_actions_strided(self._h, act0_ptr, act1_ptr, int(stride)))
        if extent is not None and act0_ptr is not None:
            self._ck(self._lib.t2d_set_action_extent(self._h, int(extent)))

    def field_ptr(self, field):
        ptr, nb = C.c_void_p(), C.c_size_t()
        self._ck(self._lib.t2d_get_field(self._h, field, C.byref(ptr), C.byref(nb)))
        return ptr.value, nb.value

    def device_array(self, field):
        """Object exposing __cuda_array_interface__ (zero-copy) for torch.as_tensor."""
        ptr, nb = self.field_ptr(field)
        dt = np.dtype(L.FIELD_DTYPES[field])
        shape = (nb // 4, 4) if field == L.F_STATUS else (L.RECORD_RING, nb // (8 * L.RECORD_RING), 2) if field == L.F_RECORD else \
            (self.n_env, nb // (4 * self.n_env)) if field == L.F_LIDAR else (nb // dt.itemsize,)
        return _DevArray(ptr, shape, dt.str, self)

    # ---------------------------------------------------------------- the hot path
    def integrate(self, interval_ms=100, stream=None):
        self._ck(self._lib.t2d_integrate(self._h, int(interval_ms), stream))

    def collide(self, stream=None):
        self._ck(self._lib.t2d_collide(self._h, stream))

    def check_status(self, interval_ms=100, stream=None):
        self._ck(self._lib.t2d_check_status(self._h, int(interval_ms), stream))

    def step(self, interval_ms=100, stream=None):
        self._ck(self._lib.t2d_step(self._h, int(interval_ms), stream))

    def step_n(self, n_steps, interval_ms=100, act_step_stride=0, stream=None):
        """n_steps consecutive steps enqueued by one call (t2d_step_n): step k reads the action of participant i at
        act[i * stride + k * act_step_stride]; 0 repeats one action set.  Same results as n_steps `step` calls."""
        self._ck(self._lib.t2d_step_n(self._h, int(interval_ms), int(n_steps), int(act_step_stride), stream))

    def set_split_step(self, on=True):
        """Small pools of 33..64-agent envs: one env per workgroup, its event stages on four waves (t2d_set_split_step)."""
        self._ck(self._lib.t2d_set_split_step(self._h, int(bool(on))))

    STEP_FORMS = ("unfused", "step", "step_split", "ego", "ego_loop", "chain", "chain_split", "loop", "loop_pipe", "ego_loop_pipe")

    def step_form(self, n_steps=1):
        """Name of the step-kernel form a call of n_steps steps takes on this pool now (t2d_step_form)."""
        rc = self._lib.t2d_step_form(self._h, int(n_steps))
        if rc < 0:
            raise ValueError("t2d_step_form: null pool")
        return self.STEP_FORMS[rc]

    def set_step_chaining(self, on=True, priority_rule=1):
        """on: False / True, or (measurements, tests) 2 = always the chained form, 3 = small pools loop without the
        integrator waves (t2d_set_step_chaining)"""
        self._ck(self._lib.t2d_set_step_chaining(self._h, int(on), int(priority_rule)))

    def snapshot(self):
        """Record the current state as the episode start for device-side resets."""
        self._ck(self._lib.t2d_snapshot(self._h))

    def restore(self, done_only=False, stream=None):
        """Device-side reset to the snapshot: all envs, or only terminated/truncated ones."""
        self._ck(self._lib.t2d_restore(self._h, 1 if done_only else 0, stream))

    def set_fused_step(self, on=True):
        """step() as one fused launch (default) or as integrate + check_status (two launches)."""
        self._ck(self._lib.t2d_set_fused_step(self._h, int(bool(on))))

    def set_ego_kernel(self, on=True):
        """single-ego pools: one wave per env (default) or the general one-lane-per-participant kernel"""
        self._ck(self._lib.t2d_set_ego_kernel(self._h, int(bool(on))))

    def set_auto_reset(self, on=True):
        """Fuse the reset of finished envs (to the snapshot) into every step()."""
        self._ck(self._lib.t2d_set_auto_reset(self._h, int(bool(on))))

    def sync(self):
        self._ck(self._lib.t2d_sync(self._h))

    # ---------------------------------------------------------------- multi-GPU result gather
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId (rank 0): 128 bytes for the other ranks' comm_init, shipped by the launcher's own channel."""
        from . import _ffi
        buf = (C.c_uint8 * 128)()
        _ffi.check(_ffi.lib().t2d_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        """ncclCommInitRank on this pool's device (collective).  unique_id None = a world of one without RCCL."""
        buf = None if unique_id is None else (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._ck(self._lib.t2d_comm_init(self._h, buf, int(rank), int(world)))
        self._comm_id = buf

    def gather(self, n_steps, out_ptr, stream=None, comm=None):
        """All-gather of the per-env result records of the last n_steps steps into caller-owned device memory
        (u32 [world][n_steps][n_env][2]); asynchronous, ordered after `stream`."""
        self._ck(self._lib.t2d_gather(self._h, comm, int(n_steps), out_ptr, stream))

    def gather_wait(self, stream=None, block_host=False):
        self._ck(self._lib.t2d_gather_wait(self._h, stream, int(bool(block_host))))

    def comm_info(self):
        """(native_rccl, world, rank) read back from the pool's communicator (ncclCommCount / ncclCommUserRank when RCCL
        created one): what a multi-GPU run prints as proof that RCCL saw N ranks."""
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.t2d_comm_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return bool(a.value), b.value, c.value

    def step_count(self):
        """Steps taken so far (t2d_step_count): what t2d_gather's fragment length must divide."""
        return int(self._lib.t2d_step_count(self._h))

    def step_occupancy(self):
        """(resident workgroups per CU, LDS bytes per workgroup) of the fused step kernel with this pool's geometry."""
        b, l, g = C.c_int32(), C.c_int64(), C.c_int64()
        self._ck(self._lib.t2d_step_occupancy(self._h, C.byref(b), C.byref(l), C.byref(g)))
        return b.value, l.value

    def geometry_bytes_per_launch(self):
        """bytes of packed geometry records (polygons, boxes, lane-union boundary pieces) one step launch stages into LDS"""
        b, l, g = C.c_int32(), C.c_int64(), C.c_int64()
        self._ck(self._lib.t2d_step_occupancy(self._h, C.byref(b), C.byref(l), C.byref(g)))
        return g.value

    # ---------------------------------------------------------------- profiling
    def profile_enable(self, on=True):
        self._ck(self._lib.t2d_profile_enable(self._h, int(bool(on))))

    def profile_read(self, kernel_id):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self._lib.t2d_profile_read(self._h, kernel_id, C.byref(ms), C.byref(n)))
        return ms.value, n.value
