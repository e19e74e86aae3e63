#!/usr/bin/env python3
"""Golden verdicts of StaticCollision / OutBound, made by EXECUTING the reference's detector classes and Vehicle.get_pose.

TEST INFRASTRUCTURE (build container only).  `traffic/event_detection/collision.py`, `out_bound.py` and
`participant/element/vehicle.py` import shapely; their definitions are parsed where they lie and executed unmodified with the
exact stand-ins of oracle/gen_golden_generator.py for what they ask of it (`Polygon`: a vertex list; `intersects` / `contains` in
rational arithmetic on the binary64 vertices; `affine_transform` by its documented rule).  What this pins is everything AROUND the
predicates -- the pose's vertex construction, which predicate is asked of which object, `any` over the obstacles, the boundary
tuple's order (xmin, xmax, ymin, ymax) and the negation in OutBound -- and, to the extent that exact arithmetic is what GEOS
evaluates robustly, the predicates themselves (geometry is not pinned against GEOS itself).

Output: tests/golden/events_static_outbound.npz (per scene: ego x, y, heading [fp32 values], length, width; obstacles; boundary;
the two verdicts)

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_events.py [--ref /root/reference]
"""
import argparse
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle.gen_golden_generator import Polygon, affine_transform   # noqa: E402  (the exact stand-ins)


def class_of(path, name, ns):
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == name)
    for fn in cls.body:
        if isinstance(fn, ast.FunctionDef):
            fn.returns = None
            for a in fn.args.args:
                a.annotation = None
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), ns)
    return ns[name], (cls.lineno, cls.end_lineno)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    ed = os.path.join(args.ref, "tactics2d", "traffic", "event_detection")
    import importlib.util
    spec = importlib.util.spec_from_file_location("t2d_ref_event_base", os.path.join(ed, "event_base.py"))
    base = importlib.util.module_from_spec(spec); spec.loader.exec_module(base)
    ns = {"EventBase": base.EventBase, "Polygon": Polygon}
    Static, l1 = class_of(os.path.join(ed, "collision.py"), "StaticCollision", ns)
    OutB, l2 = class_of(os.path.join(ed, "out_bound.py"), "OutBound", ns)
    vp = os.path.join(args.ref, "tactics2d", "participant", "element", "vehicle.py")
    vtree = ast.parse(open(vp).read(), filename=vp)
    vcls = next(n for n in vtree.body if isinstance(n, ast.ClassDef) and n.name == "Vehicle")
    gp = next(n for n in vcls.body if isinstance(n, ast.FunctionDef) and n.name == "get_pose")
    gp.returns = None
    for a in gp.args.args:
        a.annotation = None
    vns = {"np": np, "affine_transform": affine_transform}
    exec(compile(ast.Module(body=[gp], type_ignores=[]), vp, "exec"), vns)
    get_pose = vns["get_pose"]
    print("executing StaticCollision %d-%d, OutBound %d-%d, Vehicle.get_pose %d-%d" % (l1 + l2 + (gp.lineno, gp.end_lineno)))
    rng = np.random.default_rng(20261005)
    f32 = lambda v: float(np.float32(v))
    rec = {k: [] for k in ("ego", "size", "obs_off", "obs_vert_off", "obs_xy", "boundary", "static", "out")}
    rec["obs_off"].append(0); rec["obs_vert_off"].append(0)
    for k in range(1500):
        Lg, Wd = (4.284, 1.799) if k % 3 else (float(np.round(rng.uniform(2.0, 12.0), 3)), float(np.round(rng.uniform(1.0, 2.6), 3)))
        x, y, h = f32(rng.uniform(-20, 20)), f32(rng.uniform(-20, 20)), f32(rng.uniform(-7, 7))
        if k % 7 == 0:
            h = f32(rng.choice([0.0, np.pi / 2, np.pi, -np.pi / 2]))        # axis-aligned: edges parallel to obstacles' and the boundary's
        bbox = Polygon([[0.5 * Lg, -0.5 * Wd], [0.5 * Lg, 0.5 * Wd], [-0.5 * Lg, 0.5 * Wd], [-0.5 * Lg, -0.5 * Wd]])   # vehicle.py:132-140
        holder = types.SimpleNamespace(_bbox=bbox, trajectory=types.SimpleNamespace(
            get_state=lambda frame, x=x, y=y, h=h: types.SimpleNamespace(heading=h, location=(x, y))))
        pose = get_pose(holder, 0)
        obstacles = []
        for _ in range(int(rng.integers(0, 7))):
            n = int(rng.choice([3, 4, 4, 4, 6]))
            c = np.array([x + rng.uniform(-9, 9), y + rng.uniform(-9, 9)])
            ang = np.sort(rng.uniform(0, 2 * np.pi, n))
            r = rng.uniform(0.5, 4.0)
            poly = np.float32(c + r * np.stack([np.cos(ang), np.sin(ang)], 1))      # convex: points on a circle, in order
            obstacles.append(poly)
        # boundary (xmin, xmax, ymin, ymax): around the pose, sometimes cutting it
        m = rng.uniform(-1.0, 12.0, 4)
        b = (f32(x - m[0] - 2), f32(x + m[1] + 2), f32(y - m[2] - 2), f32(y + m[3] + 2))
        if k % 11 == 0:
            b = None
        st = Static([types.SimpleNamespace(geometry=Polygon(p.astype(np.float64))) for p in obstacles]).update(pose)
        ob = OutB(b).update(pose)
        rec["ego"].append([x, y, h]); rec["size"].append([Lg, Wd])
        for p in obstacles:
            rec["obs_xy"].append(p); rec["obs_vert_off"].append(rec["obs_vert_off"][-1] + len(p))
        rec["obs_off"].append(rec["obs_off"][-1] + len(obstacles))
        rec["boundary"].append([np.nan] * 4 if b is None else list(b))
        rec["static"].append(bool(st)); rec["out"].append(bool(ob))
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "events_static_outbound.npz"), ego=np.float64(rec["ego"]), size=np.float64(rec["size"]),
                        obs_off=np.int32(rec["obs_off"]), obs_vert_off=np.int32(rec["obs_vert_off"]),
                        obs_xy=np.concatenate(rec["obs_xy"]).astype(np.float32), boundary=np.float64(rec["boundary"]),
                        static=np.uint8(rec["static"]), out=np.uint8(rec["out"]))
    print(f"1500 scenes: static collision {int(np.sum(rec['static']))}, out of bound {int(np.sum(rec['out']))} -> tests/golden/events_static_outbound.npz")


if __name__ == "__main__":
    main()
