"""Host-side, opt-in history of a batch of participants (scope row a7).

The device pool keeps only the CURRENT state: the reference's per-participant frame -> State dictionary
(`tactics2d.participant.trajectory.Trajectory`, participant/trajectory/trajectory.py:12-188) grows without
bound and is exactly the Python object churn the batched path removes.  A caller who does want the history
records whole-batch `BatchedState` snapshots here.  Behaviour follows the reference's rules -- which
exception for which misuse, when `stable_freq` drops, what `reset` keeps -- restated for a batch.
"""
import logging

import numpy as np

from .physics import BatchedState

_log = logging.getLogger(__name__)


class BatchedTrajectory:
    def __init__(self, id_, fps=None, stable_freq=True):
        self.id_, self.fps, self.stable_freq = id_, fps, stable_freq
        self._stamps = []        # frames in insertion order (ms)
        self._by_frame = {}      # frame -> BatchedState
        self._now = None

    # ---- read-only views (names as in the reference) --------------------------------------------
    def __len__(self):
        return len(self._stamps)

    @property
    def frames(self):
        return self._stamps

    @property
    def history_states(self):
        return self._by_frame

    def _edge(self, i):
        return self._by_frame[self._stamps[i]] if self._stamps else None

    initial_state = property(lambda self: self._edge(0))
    last_state = property(lambda self: self._edge(-1))
    first_frame = property(lambda self: self._stamps[0] if self._stamps else None)
    last_frame = property(lambda self: self._stamps[-1] if self._stamps else None)

    @property
    def average_speed(self):
        """float64[n]: per participant, the mean speed over the recorded frames (:85-87)."""
        return np.stack([np.asarray(s.speed, np.float64) for s in self._by_frame.values()]).mean(0)

    def has_state(self, frame):
        return frame in self._by_frame

    def get_state(self, frame=None):
        if frame is None:
            return self._now
        try:
            return self._by_frame[frame]
        except KeyError:
            raise KeyError(f"trajectory {self.id_}: no state at time stamp {frame}") from None

    def get_trace(self, frame_range=None):
        """[(x[n], y[n]), ...] of the frames inside [start, end] (the whole history by default) (:151-168)."""
        lo, hi = (self.first_frame, self.last_frame) if frame_range is None else frame_range
        return [self._by_frame[f].location for f in self._stamps if lo <= f <= hi]

    # ---- mutation ----------------------------------------------------------------------------------
    def add_state(self, state):
        """Append a snapshot (:115-149): ValueError for a non-state, KeyError for a frame before the last one,
        a repeated frame overwrites (with a warning), an interval change clears `stable_freq` (with a warning)."""
        if not isinstance(state, BatchedState):
            raise ValueError("add_state expects a BatchedState")
        frame = state.frame
        if frame in self._by_frame:
            # (the reference overwrites BEFORE it checks the order, :131-135: a repeated frame that also lies before the last one
            # replaces the stored state and THEN raises -- found by replaying the reference: tests/golden/trajectory_kats.json)
            self._by_frame[frame] = state
            _log.warning("trajectory %s: state at time stamp %s overwritten", self.id_, frame)
        if self._stamps and frame < self._stamps[-1]:
            raise KeyError(f"trajectory {self.id_}: time stamp {frame} lies before the last one ({self._stamps[-1]})")
        uneven = len(self._by_frame) > 1 and frame - self._stamps[-1] != self._stamps[-1] - self._stamps[-2]
        if uneven and self.stable_freq:
            self.stable_freq = False
            _log.warning("trajectory %s: uneven time interval", self.id_)
        self._stamps.append(frame)
        self._by_frame[frame] = self._now = state

    def reset(self, state=None, keep_history=False):
        """(:170-188) no state: back to the initial state, history dropped unless keep_history;
        with a state: history dropped, the state becomes the only entry."""
        first = self.initial_state if state is None else state
        if state is None and keep_history:
            self._now = first
            return
        self._stamps, self._by_frame = [], {}
        self.add_state(first)

    def record(self, pool, frame):
        """Append the pool's current state (one download per column) as the state of `frame`."""
        from . import layout as L
        col = pool.download
        self.add_state(BatchedState(frame, col(L.F_X), col(L.F_Y), col(L.F_HEADING), col(L.F_VX), col(L.F_VY),
                                    speed=col(L.F_SPEED)))
        return self._now
