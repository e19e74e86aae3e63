"""Batched mirror of `tactics2d.map.generator.ParkingLotGenerator` (scope row f4): the rejection
sampler of map/generator/generate_parking_lot.py:239-444 run for many scenes at once on the device.

    gen = ParkingLotGenerator(vehicle_size=(4.284, 1.81), type_proportion=0.5)
    scenes = gen.generate(n_env=4096, seed=7)          # one HIP launch, one lane per scene
    scenes.scene().load(pool)                          # static geometry, target, boundary, start pose

The reference's `generate(map_)` fills a Map with obstacle Areas and returns (start_state, target_area,
target_heading) for ONE scene from numpy's global random stream; here scene e of a batch draws from the
counter stream (seed, first_env + e) -- the same scenes whatever the batch split or the number of ranks.
Parity with the reference is UNPINNED (see include/t2d.h: t2d_generate_parking); the kernel is pinned bit
for bit to its CPU restatement (oracle t2do_generate_parking).  There is no CPU fallback in this module.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _ffi
from . import layout as L
from .participant import VEHICLE_TEMPLATE, vehicle_model
from .scenarios import Scene

MAX_QUADS = 12
BAY, UNVERIFIED, START_UNVERIFIED, NONCONVEX, OVERFLOW, START_FLIPPED, TARGET_FLIPPED = 1, 2, 4, 8, 16, 32, 64


@dataclass
class ParkingScenes:
    """What `generate` returns for a batch (host arrays).  `start` / `target_heading` keep the reference's
    un-wrapped headings (+pi when the start pose was flipped, generate_parking_lot.py:409-419)."""
    quads: np.ndarray           # (n, 12, 4, 2) f32 obstacle quads in Map.areas order
    quad_id: np.ndarray         # (n, 12) reference ids, -1 = unused
    n_quads: np.ndarray         # (n,)
    start: np.ndarray           # (n, 3) f64 x, y, heading
    target: np.ndarray          # (n, 4, 2) f32
    target_heading: np.ndarray  # (n,) f64
    boundary: np.ndarray        # (n, 4) f32 xmin, xmax, ymin, ymax
    info: np.ndarray            # (n,) u32 flag bits | obstacle attempts << 8 | start attempts << 16
    vehicle_size: tuple

    @property
    def n_env(self):
        return len(self.n_quads)

    @property
    def mode(self):
        return np.where(self.info & BAY, "bay", "parallel")

    def static_csr(self):
        """(env_poly_offsets, poly_vert_offsets, verts_xy) for t2d_set_static_geometry."""
        n = self.n_quads.astype(np.int64)
        eo = np.concatenate([[0], np.cumsum(n)]).astype(np.int32)
        keep = np.arange(MAX_QUADS)[None, :] < n[:, None]
        xy = self.quads[keep].reshape(-1, 2)
        vo = (4 * np.arange(int(eo[-1]) + 1)).astype(np.int32)
        return eo, vo, np.ascontiguousarray(xy, np.float32)

    def scene(self, agent="medium_car", max_step=20000):
        """A one-ego-per-env Scene with the ParkingEnv agent (envs/parking.py:318-327: SingleTrackKinematics,
        speed +-0.5, accel +-2, steer +-0.524) at the generated start poses."""
        bad = self.info & (UNVERIFIED | START_UNVERIFIED | NONCONVEX | OVERFLOW)
        if bad.any():
            raise _ffi.GeometryError(f"{int((bad != 0).sum())} generated scenes are flagged (info bits "
                                     f"{int(np.bitwise_or.reduce(bad)):#x}); regenerate them with another seed")
        ego = vehicle_model(agent, "kinematics", speed_range=(-0.5, 0.5), accel_range=(-2.0, 2.0),
                            steer_range=(-0.524, 0.524))
        Ln, W = VEHICLE_TEMPLATE[agent][:2]
        rows = ego.param_row(L.SHAPE_OBB, Ln, W)[None]
        n = self.n_env
        return Scene("parking_generated", n, 1, rows, [agent + ":parking"], np.float32(self.start[:, 0]),
                     np.float32(self.start[:, 1]), np.float32(self.start[:, 2]), np.zeros(n, np.float32),
                     np.zeros(n, np.uint8), np.ones(n, np.uint8), static=self.static_csr(),
                     boundary=np.ascontiguousarray(self.boundary, np.float32),
                     status=dict(max_step=max_step, check_dynamic=0, check_off_lane=0, check_arrival=1,
                                 check_no_action=1, no_action_max_step=100, shaped_reward=1),
                     target=np.ascontiguousarray(self.target, np.float32),
                     target_heading=np.float32(self.target_heading))


class ParkingLotGenerator:
    """`ParkingLotGenerator(vehicle_size=(5.3, 2.5), type_proportion=0.5)` (generate_parking_lot.py:42-58):
    an invalid vehicle size falls back to the default, the proportion is clipped to [0, 1]."""
    _vehicle_size = (5.3, 2.5)

    def __init__(self, vehicle_size=(5.3, 2.5), type_proportion=0.5, device=0):
        if vehicle_size[0] < vehicle_size[1] or vehicle_size[0] <= 0 or vehicle_size[1] <= 0:
            self.vehicle_size = self._vehicle_size
        else:
            self.vehicle_size = (float(vehicle_size[0]), float(vehicle_size[1]))
        self.type_proportion = float(np.clip(type_proportion, 0, 1))
        self.device = int(device)

    def generate(self, n_env, seed, first_env=0):
        n_env = int(n_env)
        out = ParkingScenes(np.zeros((n_env, MAX_QUADS, 4, 2), np.float32), np.zeros((n_env, MAX_QUADS), np.int32),
                            np.zeros(n_env, np.int32), np.zeros((n_env, 3)), np.zeros((n_env, 4, 2), np.float32),
                            np.zeros(n_env), np.zeros((n_env, 4), np.float32), np.zeros(n_env, np.uint32),
                            self.vehicle_size)
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        _ffi.check(_ffi.lib().t2d_generate_parking(
            self.device, int(seed) & (2**64 - 1), int(first_env), n_env, self.type_proportion, self.vehicle_size[0],
            self.vehicle_size[1], ptr(out.quads), ptr(out.quad_id), ptr(out.n_quads), ptr(out.start), ptr(out.target),
            ptr(out.target_heading), ptr(out.boundary), ptr(out.info)))
        return out
