#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference physics.

TEST INFRASTRUCTURE.  Runs only in the build container, where the reference tree is
mounted read-only at /root/reference.  It imports `tactics2d.physics` (numpy only) and
`tactics2d.participant.trajectory.State`, drives them with seeded inputs and writes
small fixtures -- data only, never reference source -- into tests/golden/:

    physics_kats.json      known-answer tests (SURVEY.md 8c + Appendix C), recomputed
    kin_random.npz         SingleTrackKinematics single steps, several rigs
    dyn_random.npz         SingleTrackDynamics single steps (+ conditioning estimate)
    pm_random.npz          PointMass (newton) single steps
    rollouts.npz           VEHICLE_ACTION_LIST / PEDESTRIAN_ACTION_LIST roll-outs
    ctor_rows.json         constructor range-normalisation cases -> parameter rows

Inputs are rounded to fp32 BEFORE the reference sees them (the pool stores fp32);
outputs are the reference's fp64 results.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference]
"""
import argparse
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")

# parameter-row columns: keep in sync with include/t2d.h (checked by tests/test_layout.py)
P_MODEL, P_LF, P_LR, P_WB = 0, 1, 2, 3
P_STEER_LO, P_STEER_HI, P_SPEED_LO, P_SPEED_HI, P_ACCEL_LO, P_ACCEL_HI = 4, 5, 6, 7, 8, 9
P_RANGE_FLAGS, P_MASS, P_MASS_HEIGHT, P_MU, P_IZ, P_CF, P_CR = 10, 11, 12, 13, 14, 15, 16
P_DELTA_T, P_SHAPE, P_LENGTH, P_WIDTH = 17, 18, 19, 20
NCOL = 24
KIN, DYN, PM = 0, 1, 2

# action lists of the reference's tests/test_physics.py:52-73 (test *data*)
PEDESTRIAN_ACTION_LIST = [
    ((0, 0), 100), ((1, 0), 500), ((-1, 0), 500), ((1, 0), 500), ((0, 1), 500), ((0, -1), 500),
    ((1, 1), 500), ((2, 2), 500), ((-2, -2), 2000), ((-1, 2), 500), ((2, -1), 500),
]
VEHICLE_ACTION_LIST = [
    ((0, 0), 1000), ((1, 0), 1000), ((-1, 0), 1000), ((4, 0), 1000), ((-4, 0), 1000),
    ((15, 0), 2000), ((-15, 0), 500), ((1, 0), 1000),
    ((0.1, 0.3), 5000), ((0.1, -0.3), 5000), ((0.1, 0.6), 5000), ((0.1, -0.6), 5000),
]


def row_from_model(model, model_id, shape=(0, 0.0, 0.0)):
    """Parameter row built from the attributes of a *reference* model instance."""
    r = np.zeros(NCOL)
    r[P_MODEL] = model_id
    flags = 0
    if model_id in (KIN, DYN):
        r[P_LF], r[P_LR], r[P_WB] = model.lf, model.lr, model.wheel_base
        if model.steer_range is not None:
            r[P_STEER_LO], r[P_STEER_HI] = model.steer_range
            flags |= 1
    if model.speed_range is not None:
        r[P_SPEED_LO], r[P_SPEED_HI] = model.speed_range
        flags |= 2
    if model.accel_range is not None:
        r[P_ACCEL_LO], r[P_ACCEL_HI] = model.accel_range
        flags |= 4
    r[P_RANGE_FLAGS] = flags
    if model_id == DYN:
        r[P_MASS], r[P_MASS_HEIGHT] = model.mass, model.mass_height
        r[P_MU], r[P_IZ], r[P_CF], r[P_CR] = model.mu, model.I_z, model.cf, model.cr
    r[P_DELTA_T] = model.delta_t
    r[P_SHAPE], r[P_LENGTH], r[P_WIDTH] = shape
    return r


def f32(a):
    return np.asarray(a, dtype=np.float32)


def state_out(s, pm=False):
    vx = np.nan if s.vx is None else s.vx
    vy = np.nan if s.vy is None else s.vy
    return [s.x, s.y, s.heading, s.speed, vx, vy]


def conditioning(State, model, x, y, h0, v0, a0, s0, interval, o=None):
    """The largest change of the reference's own output (x, y, wrapped heading, speed) when ONE of its inputs (heading,
    speed, acceleration, steering angle) moves by one fp64 ulp, either way.  The deterministic trig of this build
    differs from numpy's by <= 1 ulp per call (a few dozen calls per step), so this is the yardstick for what any
    tolerance can mean: where the reference's explicit Euler is unstable (tyre-force branch at crawling speed) it
    reaches O(1) and beyond."""
    def run(hh, vv, aa, ss):
        return state_out(model.step(State(0, x=x, y=y, heading=hh, speed=vv), aa, ss, interval)[0])
    up = lambda z: float(np.nextafter(np.float64(z), np.inf))
    dn = lambda z: float(np.nextafter(np.float64(z), -np.inf))
    if o is None:
        o = run(h0, v0, a0, s0)
    worst = 0.0
    for probe in ((up(h0), v0, a0, s0), (dn(h0), v0, a0, s0), (h0, up(v0), a0, s0), (h0, dn(v0), a0, s0),
                  (h0, v0, up(a0), s0), (h0, v0, a0, up(s0)), (h0, v0, a0, dn(s0))):
        o2 = run(*probe)
        dh = abs(o[2] - o2[2]); dh = min(dh, abs(2 * np.pi - dh))
        worst = max(worst, abs(o[0] - o2[0]), abs(o[1] - o2[1]), dh, abs(o[3] - o2[3]))
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="comma-separated fixture names to (re)write, e.g. dyn_random.npz; "
                    "everything is still computed, so the seeded stream stays the same")
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    from tactics2d.participant.trajectory import State
    from tactics2d.physics import PointMass, SingleTrackDynamics, SingleTrackKinematics

    os.makedirs(OUT, exist_ok=True)
    only = set(filter(None, args.only.split(",")))

    import tempfile
    scratch = tempfile.mkdtemp(prefix="gen_golden_")

    def target(name):   # where a fixture is written: its place, or a scratch directory when --only excludes it
        return os.path.join(OUT if not only or name in only else scratch, name)

    rng = np.random.default_rng(20240915)

    # ------------------------------------------------------------------ rigs
    MED = dict(lf=4.284 / 2 - 0.880, lr=4.284 / 2 - 0.767)
    med_ranges = dict(steer_range=(-0.524, 0.524), speed_range=(-16.67, 69.44),
                      accel_range=(-11.0, 3.121))
    park_ranges = dict(steer_range=(-0.524, 0.524), speed_range=(-0.5, 0.5),
                       accel_range=(-2.0, 2.0))
    kin_rigs = {
        "medium_car": (MED, med_ranges),
        "parking": (MED, park_ranges),
        "unconstrained": (MED, {}),
        "cyclist": (dict(lf=0.9, lr=0.9), dict(steer_range=(-1.05, 1.05),
                                               speed_range=(0, 22.78), accel_range=(-7.8, 5.8))),
        "luxury_car": (dict(lf=5.302 / 2 - 0.989, lr=5.302 / 2 - 1.185),
                       dict(steer_range=(-0.524, 0.524), speed_range=(-16.67, 69.44),
                            accel_range=(-11.3, 4.146))),
    }
    timings = [(100, 5)] * 6 + [(50, 3), (9, 5), (20, 5), (100, 1)]

    # ------------------------------------------------------------------ KATs
    kats = []

    def kat(model_name, ctor, row, s0, action, interval, out, applied=None, sens=None):
        kats.append(dict(model=model_name, ctor=ctor, row=[float(v) for v in row],
                         state=s0, action=[float(a) for a in action], interval=interval,
                         out=[float(v) for v in out],
                         applied=None if applied is None else [float(a) for a in applied]))
        if sens is not None:   # dynamics only: conditioning of the reference at this input (see conditioning())
            kats[-1]["sens"] = float(sens)

    mk = SingleTrackKinematics(**MED, **park_ranges, interval=100)
    rk = row_from_model(mk, KIN)
    for s0, act in [((1.25, -3.5, 0.3, 0.2), (1.0, 0.2)), ((1.25, -3.5, 0.3, 0.2), (5.0, -0.9)),
                    ((0.0, 0.0, 6.2, -0.4), (-1.0, -0.5)), ((3.0, 4.0, -0.001, 0.5), (0.0, -0.5))]:
        s, a, d = mk.step(State(0, x=s0[0], y=s0[1], heading=s0[2], speed=s0[3]), act[0], act[1])
        kat("kinematics", "parking", rk, list(s0), act, 100, state_out(s), (a, d))
    # np.mod(-tiny, 2pi) == 2pi quirk
    mu = SingleTrackKinematics(**MED, interval=100)
    s, a, d = mu.step(State(0, x=0, y=0, heading=0.0, speed=-1e-15), 0.0, 0.3)
    kat("kinematics", "unconstrained", row_from_model(mu, KIN), [0, 0, 0.0, -1e-15], (0.0, 0.3),
        100, state_out(s), (a, d))

    md = SingleTrackDynamics(**MED, mass=1620, mass_height=0.726, **med_ranges, interval=100)
    rd = row_from_model(md, DYN)
    for s0, act in [((0, 0, 1.0, 0.0), (2, 0.3)), ((0, 0, 1.0, 0.05), (1, -0.4)),
                    ((0, 0, 1.0, 0.09), (3, 0.2)), ((0, 0, 1.0, 25.0), (-4, 0.05)),
                    ((5, -2, 0.5, -3.0), (-2, 0.1)), ((5, -2, 0.5, 30.0), (20.0, -0.9))]:
        s, a, d = md.step(State(0, x=s0[0], y=s0[1], heading=s0[2], speed=s0[3]), act[0], act[1])
        kat("dynamics", "medium_car", rd, list(s0), act, 100, state_out(s), (a, d),
            sens=conditioning(State, md, *[float(q) for q in s0], float(act[0]), float(act[1]), 100))
    # dynamics drops the remainder sub-step: interval 9 / delta_t 5
    md9 = SingleTrackDynamics(**MED, mass=1620, mass_height=0.726, **med_ranges, interval=9,
                              delta_t=5)
    mk9 = SingleTrackKinematics(**MED, **med_ranges, interval=9, delta_t=5)
    s, a, d = md9.step(State(0, x=0, y=0, heading=0.2, speed=2.0), 3.0, 0.1)
    kat("dynamics", "medium_car_9_5", row_from_model(md9, DYN), [0, 0, 0.2, 2.0], (3.0, 0.1), 9,
        state_out(s), (a, d), sens=conditioning(State, md9, 0.0, 0.0, 0.2, 2.0, 3.0, 0.1, 9))
    s, a, d = mk9.step(State(0, x=0, y=0, heading=0.2, speed=2.0), 3.0, 0.1)
    kat("kinematics", "medium_car_9_5", row_from_model(mk9, KIN), [0, 0, 0.2, 2.0], (3.0, 0.1), 9,
        state_out(s), (a, d))

    pm_cases = [
        (dict(speed_range=(-7.0, 7.0), accel_range=(-1.5, 1.5)), (0, 0, 1.0, 0.5), (1, -0.5)),
        (dict(speed_range=(0.0, 1.2), accel_range=(0, 1.5)), (0, 0, 1.0, 0.5), (3, 1)),
        (dict(speed_range=(0.0, 1.2), accel_range=(0, 1.5)), (0, 0, 1.0, 0.5), (-30, -15)),
        (dict(speed_range=(0.0, 1.2), accel_range=(0, 1.5)), (0, 0, 1.0, 0.5), (0, 0)),
        (dict(speed_range=(0.5, 7.0)), (0, 0, 0.6, 0.0), (-3, 0)),
        (dict(speed_range=(0.5, 7.0)), (0, 0, 0.5, 0.0), (0, 0)),
        (dict(speed_range=(0.5, 7.0)), (0, 0, 0.2, 0.0), (0, 0)),   # below lo, a == 0, b == 0?
        (dict(speed_range=(0.5, 7.0)), (0, 0, 0.2, 0.0), (1e-7, 0)),  # a_ < 1e-12 linear branch
        (dict(), (1, 2, -0.3, 0.4), (2, -1)),
    ]
    for ctor, s0, act in pm_cases:
        mp = PointMass(**ctor, interval=100)
        s = mp.step(State(0, x=s0[0], y=s0[1], vx=s0[2], vy=s0[3]), act)
        out = [s.x, s.y, s.heading, s.speed, s.vx, s.vy]
        kat("pointmass", repr(ctor), row_from_model(mp, PM), list(s0), act, 100, out)
    with open(target("physics_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)

    # ------------------------------------------------------------ kinematics random
    n_per = 1200
    rows, type_id, st_in, act_in, tim, st_out, app_out = [], [], [], [], [], [], []
    for rig_i, (name, (geo, rngs)) in enumerate(kin_rigs.items()):
        for ti, (interval, dt) in enumerate(timings):
            model = SingleTrackKinematics(**geo, **rngs, interval=interval, delta_t=dt)
            rows.append(row_from_model(model, KIN))
            tid = len(rows) - 1
            n = n_per // len(timings)
            span = 250.0 if ti % 2 == 0 else 30.0
            x = f32(rng.uniform(-span, span, n)); y = f32(rng.uniform(-span, span, n))
            h = f32(rng.uniform(-0.5, 2 * np.pi + 0.5, n))
            if name == "parking":
                v = f32(rng.uniform(-0.6, 0.6, n)); acc = f32(rng.uniform(-2.5, 2.5, n))
            elif name == "cyclist":
                v = f32(rng.uniform(0, 23, n)); acc = f32(rng.uniform(-9, 7, n))
            else:
                v = f32(rng.uniform(-17, 70, n)); acc = f32(rng.uniform(-13, 5, n))
            steer = f32(rng.uniform(-0.9, 0.9, n))
            for i in range(n):
                s, a, d = model.step(State(0, x=float(x[i]), y=float(y[i]), heading=float(h[i]),
                                           speed=float(v[i])), float(acc[i]), float(steer[i]),
                                     interval)
                st_out.append(state_out(s)); app_out.append([a, d])
            type_id += [tid] * n
            st_in.append(np.stack([x, y, h, v], 1)); act_in.append(np.stack([acc, steer], 1))
            tim += [[interval, dt]] * n
    np.savez_compressed(target("kin_random.npz"), rows=np.array(rows),
                        type_id=np.array(type_id, np.int32), state=np.concatenate(st_in),
                        action=np.concatenate(act_in), timing=np.array(tim, np.int32),
                        out=np.array(st_out), applied=np.array(app_out))

    # ------------------------------------------------------------ dynamics random
    templ = {  # L, front_overhang, rear_overhang, kerb, height, vmax, amax, decel
        "medium_car": (4.284, 0.880, 0.767, 1620, 1.452, 69.44, 3.121, 11.0),
        "mini_car": (3.540, 0.585, 0.535, 1070, 1.489, 44.44, 1.929, 10.0),
        "sports_utility_car": (4.828, 0.959, 0.954, 2200, 1.792, 88.89, 7.310, 10.29),
    }
    rows, type_id, st_in, act_in, tim, st_out, app_out, sens = [], [], [], [], [], [], [], []
    for name, (L, fo, ro, m, H, vmax, amax, dec) in templ.items():
        for (interval, dt) in [(100, 5), (100, 5), (50, 3), (9, 5)]:
            model = SingleTrackDynamics(lf=L / 2 - fo, lr=L / 2 - ro, mass=m, mass_height=H / 2,
                                        steer_range=(-0.524, 0.524), speed_range=(-16.67, vmax),
                                        accel_range=(-dec, amax), interval=interval, delta_t=dt)
            rows.append(row_from_model(model, DYN))
            tid = len(rows) - 1
            n = 500
            x = f32(rng.uniform(-250, 250, n)); y = f32(rng.uniform(-250, 250, n))
            h = f32(rng.uniform(0, 2 * np.pi, n))
            # 70 % cruise, 15 % urban, 15 % crawl (stiff / branch-crossing regime)
            u = rng.uniform(size=n)
            v = np.where(u < 0.7, rng.uniform(5, 45, n),
                         np.where(u < 0.85, rng.uniform(0.5, 5, n), rng.uniform(-0.3, 0.5, n)))
            v = f32(v)
            acc = f32(rng.uniform(-12, 5, n)); steer = f32(rng.normal(0, 0.08, n))
            steer[::7] = f32(rng.uniform(-0.7, 0.7, len(steer[::7])))
            for i in range(n):
                st = State(0, x=float(x[i]), y=float(y[i]), heading=float(h[i]), speed=float(v[i]))
                s, a, d = model.step(st, float(acc[i]), float(steer[i]), interval)
                o = state_out(s)
                sens.append(conditioning(State, model, float(x[i]), float(y[i]), float(h[i]), float(v[i]),
                                         float(acc[i]), float(steer[i]), interval, o))
                st_out.append(o); app_out.append([a, d])
            type_id += [tid] * n
            st_in.append(np.stack([x, y, h, v], 1)); act_in.append(np.stack([acc, steer], 1))
            tim += [[interval, dt]] * n
    np.savez_compressed(target("dyn_random.npz"), rows=np.array(rows),
                        type_id=np.array(type_id, np.int32), state=np.concatenate(st_in),
                        action=np.concatenate(act_in), timing=np.array(tim, np.int32),
                        out=np.array(st_out), applied=np.array(app_out), sens=np.array(sens))

    # ------------------------------------------------------------ point mass random
    pm_rigs = [dict(speed_range=(-7.0, 7.0), accel_range=(-1.5, 1.5)),
               dict(speed_range=(-3.5, 3.5), accel_range=(-1.0, 1.0)),
               dict(speed_range=(0.5, 4.5)), dict(), dict(speed_range=6.0, accel_range=1.5),
               dict(speed_range=(0.0, 1.2))]
    rows, type_id, st_in, act_in, tim, st_out = [], [], [], [], [], []
    for ctor in pm_rigs:
        for interval in (100, 100, 50, 9):
            model = PointMass(**ctor, interval=interval)
            rows.append(row_from_model(model, PM, (1, 0.24, 0.40)))
            tid = len(rows) - 1
            n = 300
            x = f32(rng.uniform(-250, 250, n)); y = f32(rng.uniform(-250, 250, n))
            vx = f32(rng.uniform(-5, 5, n)); vy = f32(rng.uniform(-5, 5, n))
            vx[::11] = 0; vy[::11] = 0
            ax = f32(rng.uniform(-20, 20, n)); ay = f32(rng.uniform(-20, 20, n))
            ax[::5] = f32(rng.uniform(-2, 2, len(ax[::5]))); ay[::5] = f32(rng.uniform(-2, 2, len(ay[::5])))
            ax[::13] = 0; ay[::13] = 0
            for i in range(n):
                s = model.step(State(0, x=float(x[i]), y=float(y[i]), vx=float(vx[i]),
                                     vy=float(vy[i])), (float(ax[i]), float(ay[i])), interval)
                st_out.append([s.x, s.y, s.heading, s.speed, s.vx, s.vy])
            type_id += [tid] * n
            st_in.append(np.stack([x, y, vx, vy], 1)); act_in.append(np.stack([ax, ay], 1))
            tim += [[interval, 5]] * n
    np.savez_compressed(target("pm_random.npz"), rows=np.array(rows),
                        type_id=np.array(type_id, np.int32), state=np.concatenate(st_in),
                        action=np.concatenate(act_in), timing=np.array(tim, np.int32),
                        out=np.array(st_out))

    # ------------------------------------------------------------ roll-outs (fp64 free-running)
    roll = {}
    for (interval, dt) in [(100, 5), (50, 3), (9, 5)]:
        mk_ = SingleTrackKinematics(**MED, **med_ranges, interval=interval, delta_t=dt)
        md_ = SingleTrackDynamics(**MED, mass=1620, mass_height=0.726, **med_ranges,
                                  interval=interval, delta_t=dt)
        for tag, model, mid in (("kin", mk_, KIN), ("dyn", md_, DYN)):
            s = State(frame=0, x=10, y=10, heading=0, speed=0)
            traj = [[s.x, s.y, s.heading, s.speed]]; acts = []
            for action, duration in VEHICLE_ACTION_LIST:
                for _ in np.arange(0, duration, interval):
                    s, _, _ = model.step(s, action[0], action[1], interval)
                    traj.append([s.x, s.y, s.heading, s.speed]); acts.append(action)
            roll[f"{tag}_{interval}_{dt}_traj"] = np.array(traj)
            roll[f"{tag}_{interval}_{dt}_act"] = np.array(acts, float)
            roll[f"{tag}_{interval}_{dt}_row"] = row_from_model(model, mid)
            roll[f"{tag}_{interval}_{dt}_frame"] = np.array([s.frame])
            if tag == "dyn":   # conditioning of every step, teacher-forced from the fp32-rounded state like the tests
                t32 = np.float64(f32(np.array(traj[:-1]))); a32 = np.float64(f32(np.array(acts, float)))
                roll[f"{tag}_{interval}_{dt}_sens"] = np.array(
                    [conditioning(State, model, *[float(q) for q in t32[k]], float(a32[k, 0]), float(a32[k, 1]), interval)
                     for k in range(len(acts))])
    for k, (sr, ar, interval, dt) in enumerate([([0, 5], [0, 2], 100, 5), ([-5, 5], [-2, 2], 9, 5),
                                                ([5, 5], [2, 2], 50, 3), (5, 2, 100, 5),
                                                (-5, -2, 100, 5), (None, None, 100, 5)]):
        mp = PointMass(sr, ar, interval, dt, "newton")
        me = PointMass(sr, ar, interval, dt, "euler")
        s = State(frame=0, x=10, y=10, heading=0, speed=0)
        se = State(frame=0, x=10, y=10, heading=0, speed=0)
        traj = [[s.x, s.y, 0.0, 0.0]]; traje = [[se.x, se.y]]; acts = []
        for action, duration in PEDESTRIAN_ACTION_LIST:
            for _ in np.arange(0, duration, interval):
                s = mp.step(s, action, interval); se = me.step(se, action, interval)
                traj.append([s.x, s.y, s.vx, s.vy]); traje.append([se.x, se.y]); acts.append(action)
        roll[f"pm_{k}_traj"] = np.array(traj); roll[f"pm_{k}_euler"] = np.array(traje)
        roll[f"pm_{k}_act"] = np.array(acts, float)
        roll[f"pm_{k}_row"] = row_from_model(mp, PM); roll[f"pm_{k}_timing"] = np.array([interval, dt])
    np.savez_compressed(target("rollouts.npz"), **roll)

    # ------------------------------------------------------------ constructor normalisation
    ctor_cases = []
    range_args = [None, 5.0, -5.0, 0.0, 5, (-1.0, 2.0), (2.0, -1.0), (1.0, 1.0), [0, 3], (-3, 0),
                  (0.5, 4.5), (1.0,)]
    for ra in range_args:
        jr = list(ra) if isinstance(ra, (tuple, list)) else ra
        for dt_arg, interval in [(None, 100), (0, 100), (7, 100), (200, 100), (3, None)]:
            mk_ = SingleTrackKinematics(1.2, 1.3, ra, ra, ra, interval, dt_arg)
            md_ = SingleTrackDynamics(1.2, 1.3, 1500.0, 0.7, steer_range=ra, speed_range=ra,
                                      accel_range=ra, interval=interval, delta_t=dt_arg)
            mp_ = PointMass(ra, ra, interval, dt_arg)
            ctor_cases.append(dict(range=jr, delta_t=dt_arg, interval=interval,
                                   kin=row_from_model(mk_, KIN).tolist(),
                                   dyn=row_from_model(md_, DYN).tolist(),
                                   pm=row_from_model(mp_, PM).tolist()))
    with open(target("ctor_rows.json"), "w") as f:
        json.dump(ctor_cases, f)
    print("golden vectors written to", os.path.normpath(OUT))


if __name__ == "__main__":
    main()
