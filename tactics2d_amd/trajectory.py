"""Host-side, opt-in history of a batch of participants (scope row a7).

Reference: `tactics2d.participant.trajectory.Trajectory` (participant/trajectory/trajectory.py:12-188), an
append-only frame -> State map per participant.  The device pool keeps only the CURRENT state (an unbounded
per-env dict of Python objects is exactly what the batched path removes); a caller who wants the history
records `BatchedState` snapshots here -- one entry per frame for the whole batch -- with the reference's
rules: `add_state` rejects non-states (ValueError) and frames earlier than the last one (KeyError), warns
and overwrites on a repeated frame, clears `stable_freq` when the interval changes; `get_state`, `has_state`,
`get_trace`, `reset(state, keep_history)` and the read-only properties behave as there.
"""
import logging

import numpy as np

from .physics import BatchedState


class BatchedTrajectory:
    def __init__(self, id_, fps=None, stable_freq=True):
        self.id_ = id_
        self.fps = fps
        self.stable_freq = stable_freq
        self._history_states = {}
        self._frames = []
        self._current_state = None

    def __len__(self):
        return len(self._frames)

    frames = property(lambda self: self._frames)
    history_states = property(lambda self: self._history_states)
    initial_state = property(lambda self: None if len(self) == 0 else self._history_states[self._frames[0]])
    last_state = property(lambda self: None if len(self) == 0 else self._history_states[self._frames[-1]])
    first_frame = property(lambda self: None if len(self) == 0 else self._frames[0])
    last_frame = property(lambda self: None if len(self) == 0 else self._frames[-1])

    @property
    def average_speed(self):
        """Per participant: mean over the recorded frames (trajectory.py:85-87), float64[n]."""
        return np.mean([np.asarray(s.speed, np.float64) for s in self._history_states.values()], axis=0)

    def has_state(self, frame):
        return frame in self._history_states

    def get_state(self, frame=None):
        if frame is None:
            return self._current_state
        if frame not in self._history_states:
            raise KeyError(f"Time stamp {frame} is not found in the trajectory {self.id_}.")
        return self._history_states[frame]

    def add_state(self, state):
        if not isinstance(state, BatchedState):
            raise ValueError("The input state is not a valid State object.")
        if state.frame in self._history_states:
            self._history_states[state.frame] = state
            logging.warning(f"State at time stamp {state.frame} is already in trajectory {self.id_}. It will be overwritten.")
        if len(self._frames) > 0 and state.frame < self._frames[-1]:
            raise KeyError(f"Trying to insert an early time stamp {state.frame} happening before the last stamp "
                           f"{self._frames[-1]} in trajectory {self.id_}")
        if len(self._history_states) > 1:
            current_interval = state.frame - self._frames[-1]
            last_interval = self._frames[-1] - self._frames[-2]
            if current_interval != last_interval and self.stable_freq:
                self.stable_freq = False
                logging.warning(f"The time interval of the trajectory {self.id_} is uneven.")
        self._frames.append(state.frame)
        self._history_states[state.frame] = state
        self._current_state = state

    def get_trace(self, frame_range=None):
        """List of (x[n], y[n]) locations of the frames inside the range (all frames by default)."""
        start = self.first_frame if frame_range is None else frame_range[0]
        end = self.last_frame if frame_range is None else frame_range[1]
        return [self.get_state(f).location for f in self._frames if start <= f <= end]

    def reset(self, state=None, keep_history=False):
        if state is None:
            initial_state = self.initial_state
            if not keep_history:
                self._history_states.clear()
                self._frames.clear()
                self.add_state(initial_state)
            else:
                self._current_state = initial_state
        else:
            self._history_states.clear()
            self._frames.clear()
            self.add_state(state)

    def record(self, pool, frame):
        """Append the pool's current state (one download per column) as the state of `frame`."""
        from . import layout as L
        d = pool.download
        self.add_state(BatchedState(frame, d(L.F_X), d(L.F_Y), d(L.F_HEADING), d(L.F_VX), d(L.F_VY), speed=d(L.F_SPEED)))
        return self._current_state
