"""Env groups on separate HIP streams: software pipelining of independent environments on one GPU.

Environments never interact (the reference's ScenarioManager owns exactly one scene), so a batch of envs can
be cut into G groups, each with its own pool and its own stream.  A step launch is one kernel in which every
wave starts at the same time: ~2.4 us of load latency at the start and a tail in which each SIMD's last wave
runs alone are exposed once per launch.  With G groups in flight those stretches of one group overlap the
busy middle of the others -- and the tail of step k overlaps the start of step k+1 of the next group, which a
single launch per step cannot do (a kernel boundary is a device-wide barrier).  Measured on the metric scene
(4096 envs x 64 participants): 32.2 us per step as one launch, 25.0 us as 2 groups, 23.5 us as 4 groups
(scripts/env_groups_sweep.py); 8 groups become host-launch bound.  Results are identical to the single-pool run.

In an RL loop the policy of group g+1 runs while the physics of group g does (the usual double-buffered
vector env); `EnvGroups` is the plumbing for that.  torch supplies the streams (plumbing only).

HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise:
with torch's own streams in the process, 4 groups need GPU_MAX_HW_QUEUES >= 8 in the environment BEFORE the HIP
runtime initialises (bench.py sets it; measured 39 us per step with the default 4 queues, 23.6 us with 8).
"""
import ctypes as C
import os

import numpy as np

# effective only if the HIP runtime has not initialised yet (import this module before the first GPU call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _ffi, layout as L
from .pool import ParticipantPool


class EnvGroups:
    def __init__(self, scene, groups, device_id=0, raw_streams=False, library=None):
        """library: the loaded C library the groups' pools live in (default libt2d_hip.so; tactics2d_amd.debug passes
        libt2d_hip_debug.so for the closed-loop measurement)"""
        import torch
        if groups < 1 or scene.n_env % groups:
            raise ValueError(f"{groups} groups must divide {scene.n_env} envs")
        self.scene, self.groups, self.device_id = scene, groups, device_id
        per = scene.n_env // groups
        self.bounds = [(g * per, (g + 1) * per) for g in range(groups)]
        self.pools, self.streams, self._raw = [], [], []
        dev = torch.device("cuda", device_id)
        for lo, hi in self.bounds:
            pool = ParticipantPool(hi - lo, scene.A, device_id, library=library)
            scene.shard(lo, hi).load(pool)
            self.pools.append(pool)
            if raw_streams:   # streams of the library's own making, wrapped for torch's stream API (wait_stream, synchronize)
                h = C.c_void_p()
                _ffi.check((library or _ffi.lib()).t2d_stream_create(device_id, 0, C.byref(h)))
                self._raw.append(h)
                self.streams.append(torch.cuda.ExternalStream(h.value, device=dev))
            else:
                self.streams.append(torch.cuda.Stream(device=dev))
        self.n_env, self.A, self.n = scene.n_env, scene.A, scene.n
        # one C call steps every group (t2d_step_groups): handle / stream / action-pointer arrays built once
        self._lib = library if library is not None else _ffi.lib()
        vp = C.c_void_p * groups
        self._handles = vp(*[p._h for p in self.pools])
        self._streams = vp(*[s.cuda_stream for s in self.streams])
        self._act_cache = {}
        self._act = None

    def configure(self, fn):
        """fn(pool) for every group's pool (variant, auto-reset, IDM ... )."""
        for p in self.pools:
            fn(p)

    def bind_actions(self, act0, act1):
        """act0 / act1: device tensors [n_env * A] (float32, contiguous); each group reads its slice zero-copy.
        Takes effect with the next step() (the pointers travel in the same C call)."""
        key = (act0.data_ptr(), act1.data_ptr())
        arr = self._act_cache.get(key)
        if arr is None:
            vp = C.c_void_p * self.groups
            offs = [4 * lo * self.A for lo, _ in self.bounds]
            arr = (vp(*[key[0] + o for o in offs]), vp(*[key[1] + o for o in offs]))
            if len(self._act_cache) < 64:
                self._act_cache[key] = arr
        self._act = arr

    def step(self, interval_ms):
        """One step of every group, each on its own stream, in one host call; returns immediately."""
        a0, a1 = self._act if self._act is not None else (None, None)
        rc = self._lib.t2d_step_groups(self._handles, a0, a1, self._streams, self.groups, int(interval_ms))
        if rc:
            for p in self.pools:
                msg = self._lib.t2d_last_error(p._h)
                if msg:
                    _ffi.check(rc, p._h, self._lib)
            _ffi.check(rc, None, self._lib)

    def join(self, stream=None):
        """Make `stream` (a torch stream; default: the current one) wait for everything launched so far."""
        import torch
        tgt = stream if stream is not None else torch.cuda.current_stream()
        for s in self.streams:
            tgt.wait_stream(s)

    def fork(self, stream=None):
        """Make every group's stream wait for work already queued on `stream` (e.g. the policy writing actions)."""
        import torch
        src = stream if stream is not None else torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(src)

    def sync(self):
        for s in self.streams:
            s.synchronize()

    def download(self, field):
        """Concatenation over the groups in env order (per-env and per-participant fields alike)."""
        self.sync()
        parts = [p.download(field) for p in self.pools]
        axis = 1 if field == L.F_RECORD else 0
        return np.concatenate(parts, axis=axis)

    def close(self):
        self.sync()
        for p in self.pools:
            p.close()
        self.pools = []
        for h in self._raw:
            self._lib.t2d_stream_destroy(h)
        self._raw = []
