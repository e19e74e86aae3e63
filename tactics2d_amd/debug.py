"""Test and measurement hooks (include/t2d_debug.h) -- NOT part of the product.

Everything here runs on libt2d_hip_debug.so: the product's sources built with -DT2D_DEBUG_HOOKS plus t2d_loop.hip
(`python -m tactics2d_amd.build --debug-lib`; `__graft_entry__.build()` builds it next to the product).  It exports the whole
product ABI as well, and a pool belongs to the library that created it: `pool()` / `env_groups()` below make pools there.
Loaded by tests/, bench.py's closed_loop leg and scripts/; no product module imports this one (tests/test_layout.py), and
libt2d_hip.so exports no t2d_debug_* symbol.

    pool(n_env, A)                    a ParticipantPool living in the debug library
    set_step_placement(pool, map)     which logical workgroup / wave rotation each physical workgroup of a step launch takes
    chain_fault(pool, kind)           break one hand-off of the chained t2d_step_n on purpose
    delay_gather(pool, us)            hold the pool's gather stream (a slow peer)
    feedback_policy(pool, act, ...)   the stand-in policy kernel
    env_groups(scene, G) / ClosedLoop policy kernel -> t2d_step per env group, enqueued by one C call
"""
import ctypes as C
import os

import numpy as np

from . import _ffi
from .pipeline import EnvGroups
from .pool import ParticipantPool


_vp = C.c_void_p
# every symbol include/t2d_debug.h declares: test / measurement hooks, exported by libt2d_hip_debug.so ONLY (tactics2d_amd/debug.py)
DEBUG_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libt2d_hip_debug.so")
DEBUG_SYMBOLS = {
    "t2d_debug_set_step_placement": (C.c_int, [_vp, C.POINTER(C.c_uint32), C.c_int32]),
    "t2d_debug_chain_fault": (C.c_int, [_vp, C.c_int32]),
    "t2d_debug_delay_gather": (C.c_int, [_vp, C.c_int32]),
    "t2d_debug_feedback_policy": (C.c_int, [_vp, _vp, C.c_float, C.c_float, C.c_float, _vp]),
    "t2d_debug_closed_loop_create": (C.c_int, [_vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "t2d_debug_closed_loop_run": (C.c_int, [_vp, C.c_int32]),
    "t2d_debug_closed_loop_destroy": (C.c_int, [_vp]),
    "t2d_debug_last_step_kernel": (C.c_char_p, []),
}

_debug_lib = None


def lib():
    """libt2d_hip_debug.so: everything libt2d_hip.so exports + the hooks of include/t2d_debug.h; a pool belongs to the
    library that created it."""
    global _debug_lib
    if _debug_lib is None:
        _debug_lib = _ffi._load(DEBUG_LIB_PATH, {**_ffi.SYMBOLS, **DEBUG_SYMBOLS}, " --debug-lib")
    return _debug_lib


def pool(n_env, max_agents=1, device_id=0):
    return ParticipantPool(n_env, max_agents, device_id, library=lib())


def env_groups(scene, groups, device_id=0, raw_streams=False):
    return EnvGroups(scene, groups, device_id, raw_streams, library=lib())


def _need(p):
    if p._lib is not lib():
        raise ValueError("this pool lives in libt2d_hip.so, which has no test hooks: create it with tactics2d_amd.debug.pool(...)")


def set_step_placement(p, wgmap=None):
    """Which logical workgroup (and wave rotation << 16) each physical workgroup of the step launch steps; None = identity.
    Never changes a result (t2d_debug.h: t2d_debug_set_step_placement)."""
    _need(p)
    if wgmap is None:
        p._ck(p._lib.t2d_debug_set_step_placement(p._h, None, 0))
        return
    m = np.ascontiguousarray(wgmap, np.uint32)
    p._ck(p._lib.t2d_debug_set_step_placement(p._h, m.ctypes.data_as(C.POINTER(C.c_uint32)), int(m.size)))


def chain_fault(p, kind):
    """t2d_debug_chain_fault: the CHAIN launches of step_n break one hand-off on purpose -- 1: a foreign XCC id in the word,
    2: a word that never comes, 3: a foreign XCC id at step 0; 0: off."""
    _need(p)
    p._ck(p._lib.t2d_debug_chain_fault(p._h, int(kind)))


def delay_gather(p, microseconds):
    """t2d_debug_delay_gather: one idle wave holds the pool's gather stream for that long"""
    _need(p)
    p._ck(p._lib.t2d_debug_delay_gather(p._h, int(microseconds)))


def last_step_kernel():
    """template arguments of the collide_kernel instantiation the last launch took, e.g. "(true, 1, false, true)\""""
    return lib().t2d_debug_last_step_kernel().decode()


def feedback_policy(p, act_out_ptr, v_target, k_speed, k_steer, stream=None):
    """t2d_debug_feedback_policy: the stand-in policy, one launch; act_out_ptr = device f32 [N][2] (steering, accel)"""
    _need(p)
    p._ck(p._lib.t2d_debug_feedback_policy(p._h, act_out_ptr, v_target, k_speed, k_steer, stream))


class ClosedLoop:
    """The closed loop a policy-driven caller runs -- per env group and step: policy kernel -> t2d_step, no host
    synchronisation -- enqueued `n` iterations at a time by one C call (t2d_debug_closed_loop_*, t2d_loop.hip).  The policy is
    the library's stand-in (per-participant state feedback, a few flops): what is timed is the step path under a real
    dependency -- step k + 1 of a group cannot start before its policy has read step k's state -- not a network.

    launcher: "thread" (the calling thread goes round the groups), "threads" (one host thread per group) or "graph" (one
    captured hipGraph of `graph_steps` iterations per group, replayed).  Results equal those of the same policy and t2d_step
    calls on one pool holding all the envs (tests/test_gpu_closed_loop.py)."""
    LAUNCHERS = {"thread": 0, "threads": 1, "graph": 2}

    def __init__(self, groups, launcher="thread", interval_ms=100, graph_steps=64):
        import torch
        self.groups, self.launcher = groups, launcher
        self._lib = lib()
        if any(p._lib is not self._lib for p in groups.pools):
            raise ValueError("ClosedLoop needs env groups created on libt2d_hip_debug.so: debug.env_groups(...)")
        dev = torch.device("cuda", groups.device_id)
        # one [n_g, 2] (steering, accel) tensor per group: what a policy network would return for the group's participants
        self.actions = [torch.zeros((p.n, 2), dtype=torch.float32, device=dev) for p in groups.pools]
        torch.cuda.synchronize(dev)
        G = groups.groups
        vp = C.c_void_p * G
        self._h = C.c_void_p()
        rc = self._lib.t2d_debug_closed_loop_create(vp(*[p._h for p in groups.pools]), vp(*[s.cuda_stream for s in groups.streams]),
                                                    vp(*[a.data_ptr() for a in self.actions]), G, int(interval_ms),
                                                    self.LAUNCHERS[launcher], int(graph_steps), C.byref(self._h))
        if rc:
            for p in groups.pools:
                if self._lib.t2d_last_error(p._h):
                    _ffi.check(rc, p._h, self._lib)
            _ffi.check(rc, None, self._lib)

    def run(self, n_steps):
        """enqueue n_steps iterations of (policy, step) on every group; returns when everything is enqueued"""
        rc = self._lib.t2d_debug_closed_loop_run(self._h, int(n_steps))
        if rc:
            for p in self.groups.pools:
                if self._lib.t2d_last_error(p._h):
                    _ffi.check(rc, p._h, self._lib)
            _ffi.check(rc, None, self._lib)

    def close(self):
        if self._h:
            self.groups.sync()
            self._lib.t2d_debug_closed_loop_destroy(self._h)
            self._h = None
