#!/usr/bin/env python3
"""Known-answer vectors of `State`'s derived fields and type coercion (participant/trajectory/state.py:108-204), produced
by IMPORTING the reference.  TEST INFRASTRUCTURE; runs only where /root/reference is mounted.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_state.py [--ref /root/reference]

Writes tests/golden/state_kats.json: derived speed / velocity / accel / acceleration for seeded inputs (inputs rounded to
fp32 first: the pool stores fp32), and the messages of the ValueError the typed __setattr__ raises.
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "state_kats.json")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    from tactics2d.participant.trajectory import State

    rng = np.random.default_rng(606)
    f = lambda v: float(np.float32(v))
    derived = []
    for _ in range(60):
        h, v, a = f(rng.uniform(-7, 7)), f(rng.uniform(-20, 40)), f(rng.uniform(-11, 5))
        vx, vy, ax, ay = (f(q) for q in rng.uniform(-9, 9, 4))
        for kw in (dict(heading=h, speed=v, accel=a),                       # scalar speed / accel only: accel = |accel| quirk
                   dict(heading=h, vx=vx, vy=vy, ax=ax, ay=ay),             # vectors only: speed / accel are norms
                   dict(heading=h, vx=vx, vy=vy, speed=v, ax=ax, ay=ay, accel=a),   # both: speed = the scalar, accel = the norm
                   dict(heading=h)):                                         # nothing: everything None
            s = State(0, x=1.0, y=2.0, **kw)
            t = lambda q: None if q is None else [float(z) for z in np.atleast_1d(q)]
            derived.append(dict(kw=kw, speed=t(s.speed), velocity=t(s.velocity), accel=t(s.accel),
                                acceleration=t(s.acceleration)))
    # SURVEY finding 10: accel of a scalar-only State is ||accel (cos h, sin h)||, not accel
    s = State(0, heading=0.3, accel=1.0)
    quirk = dict(heading=0.3, accel_in=1.0, accel_out=float(s.accel))
    errors = []
    for name, value in (("x", "abc"), ("frame", "1.5x"), ("heading", [1.0, 2.0]), ("vx", {"a": 1}), ("frame", None)):
        try:
            s = State(0)
            setattr(s, name, value)
            errors.append(dict(name=name, value=repr(value), raised=False, stored=repr(getattr(s, name))))
        except ValueError as e:
            errors.append(dict(name=name, value=repr(value), raised=True, message=str(e)))
    coerced = []
    for name, value in (("x", 3), ("x", "2.5"), ("frame", 7.9), ("frame", "12"), ("heading", True)):
        s = State(0)
        setattr(s, name, value)
        coerced.append(dict(name=name, value=repr(value), stored=getattr(s, name), type=type(getattr(s, name)).__name__))
    s = State(5, x=1, y=2, heading=0.5, speed=3.0)
    s2 = State(5, vx=3.0, vy=4.0)
    s2.set_accel(1.0, -2.0)
    setters = dict(accel_after_set_accel=float(s2.accel), _accel=float(s2._accel),
                   speed_cache_after_set_velocity=None)
    _ = s.velocity
    s.set_velocity(1.0, 1.0)
    setters["velocity_after_set_velocity"] = [float(q) for q in s.velocity]
    setters["speed_after_set_velocity"] = float(s.speed)      # the scalar that was set stays: 3.0
    with open(OUT, "w") as fh:
        json.dump(dict(derived=derived, quirk=quirk, errors=errors, coerced=coerced, setters=setters), fh, indent=1)
    print(OUT, len(derived), "derived cases;", quirk, errors)


if __name__ == "__main__":
    main()
