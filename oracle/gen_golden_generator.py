#!/usr/bin/env python3
"""Golden scenes of ParkingLotGenerator.generate, made by EXECUTING the reference's own class -- with the random draws it consumed
recorded, so that the build's restatement can be REPLAYED on the same draws.

TEST INFRASTRUCTURE (build container only; the reference tree is mounted read-only).
`tactics2d/map/generator/generate_parking_lot.py` cannot be imported (shapely).  This script parses the file where it lies and
executes the whole `ParkingLotGenerator` class, unmodified (lines 19-444), and `Map.add_area` (map/element/map.py:435-457), with
  * the reference's own `State` (importable);
  * numpy as it is, except that `np.random` is a recorder in front of numpy's global MT19937 stream (seeded per scene): every
    value `normal` / `uniform` / `rand` hands the reference goes onto a tape, with its kind (0 = a uniform in [0, 1), 1 =
    uniform(a, b), 2 = normal(mean, std));
  * stand-ins for what the class asks of shapely, each by its documented meaning, evaluated EXACTLY (rational arithmetic on the
    binary64 coordinates) where it is a predicate: `Point` (a coordinate pair), `Polygon` (a vertex list; `.exterior.coords` is the
    closed ring in the given order), `affine_transform` (x' = a x + b y + xoff, y' = d x + e y + yoff), `intersects` (the closed
    sets share a point: two edges meet, or a vertex of one lies in the other), `contains` (no vertex of the other outside), and
    `distance` (0 if they intersect, else the smallest vertex-to-edge distance, fp64);
  * data holders for `Area` and the map (`.areas`, `.ids`, `.name`, `.scenario_type`, `set_boundary`).
The build's generator draws from a counter stream of its own (t2d_oracle.c, t2d_generate.hip), so its scenes are not the reference's
-- but given the SAME draws they must be: tests/test_generator.py feeds every tape to t2do_generate_parking_replay and compares the
scene (bay / parallel, every obstacle in Map.areas order with its id, the target, the start pose, the boundary).  That pins the
restatement's draw ORDER, control flow (rejection loops, the stale obstacles of rejected attempts, the id collision on "0003"),
distributions' parameters and arithmetic against the reference; what it cannot pin is shapely's own rounding inside `distance`
(the stand-in's differs from GEOS' by an ulp at most, far inside the comparison's tolerance) -- geometry stays DESIGN.md 1c's.

Output: tests/golden/generator_replay.npz

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_generator.py [--ref /root/reference] [--n 240]
"""
import argparse
import ast
import logging
import os
import sys
import time
import types
from fractions import Fraction as Fr

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")


# ---- what the class asks of shapely ------------------------------------------------------------------------------------------
class Point:
    def __init__(self, *a):
        if len(a) == 1:
            a = (a[0].x, a[0].y) if hasattr(a[0], "x") else tuple(a[0])
        self.x, self.y = float(a[0]), float(a[1])


def _xy(v):
    return (float(v.x), float(v.y)) if hasattr(v, "x") else (float(v[0]), float(v[1]))


def _orient(p, q, r):
    return (Fr(q[0]) - Fr(p[0])) * (Fr(r[1]) - Fr(p[1])) - (Fr(q[1]) - Fr(p[1])) * (Fr(r[0]) - Fr(p[0]))


def _on_segment(p, q, r):      # r on the closed segment pq, given collinear
    return min(p[0], q[0]) <= r[0] <= max(p[0], q[0]) and min(p[1], q[1]) <= r[1] <= max(p[1], q[1])


def _segments_meet(a, b, c, d):
    o1, o2, o3, o4 = _orient(a, b, c), _orient(a, b, d), _orient(c, d, a), _orient(c, d, b)
    if ((o1 > 0) != (o2 > 0)) and o1 != 0 and o2 != 0 and ((o3 > 0) != (o4 > 0)) and o3 != 0 and o4 != 0:
        return True
    return (o1 == 0 and _on_segment(a, b, c)) or (o2 == 0 and _on_segment(a, b, d)) or \
           (o3 == 0 and _on_segment(c, d, a)) or (o4 == 0 and _on_segment(c, d, b))


def _inside_closed(poly, r):   # r in the closed region of a simple polygon: on the boundary, or an odd number of crossings
    n = len(poly)
    crossings = 0
    for i in range(n):
        p, q = poly[i], poly[(i + 1) % n]
        if _orient(p, q, r) == 0 and _on_segment(p, q, r):
            return True
        if (p[1] > r[1]) != (q[1] > r[1]):
            # x of the edge at height r.y, exactly
            xc = Fr(p[0]) + (Fr(r[1]) - Fr(p[1])) * (Fr(q[0]) - Fr(p[0])) / (Fr(q[1]) - Fr(p[1]))
            if xc > Fr(r[0]):
                crossings += 1
    return crossings % 2 == 1


class Polygon:
    def __init__(self, verts):
        if isinstance(verts, Polygon):
            verts = verts.pts
        self.pts = [_xy(v) for v in verts]
        self.exterior = types.SimpleNamespace(coords=self.pts + self.pts[:1])

    def _edges(self):
        n = len(self.pts)
        return [(self.pts[i], self.pts[(i + 1) % n]) for i in range(n)]

    def intersects(self, other):
        for a, b in self._edges():
            for c, d in other._edges():
                if _segments_meet(a, b, c, d):
                    return True
        return _inside_closed(other.pts, self.pts[0]) or _inside_closed(self.pts, other.pts[0])

    def contains(self, other):
        return all(_inside_closed(self.pts, p) for p in other.pts)

    def distance(self, other):
        if self.intersects(other):
            return 0.0
        best = numpy.inf
        for A, B in ((self, other), (other, self)):
            for p in A.pts:
                for c, d in B._edges():
                    c_, d_, p_ = numpy.array(c), numpy.array(d), numpy.array(p)
                    e = d_ - c_
                    t = numpy.clip(numpy.dot(p_ - c_, e) / numpy.dot(e, e), 0.0, 1.0)
                    best = min(best, float(numpy.hypot(*(c_ + t * e - p_))))
        return best


def affine_transform(poly, m):
    a, b, d, e, xo, yo = m
    return Polygon([(a * x + b * y + xo, d * x + e * y + yo) for x, y in poly.pts])


class Area:
    def __init__(self, id_, geometry=None, type_=None, subtype=None, color=None, **kw):
        self.id_, self.geometry, self.type_, self.subtype, self.color = id_, geometry, type_, subtype, color


class _Recorder:
    """np.random of the executed class: numpy's global stream, every value handed out noted"""

    def __init__(self):
        self.kind, self.val = [], []

    def _note(self, k, v):
        for x in numpy.asarray(v, numpy.float64).reshape(-1):
            self.kind.append(k); self.val.append(float(x))
        return v

    def normal(self, loc=0.0, scale=1.0, size=None):
        return self._note(2, numpy.random.normal(loc, scale, size))

    def uniform(self, *a, **kw):
        bounded = len(a) >= 2 or "low" in kw or "high" in kw
        return self._note(1 if bounded else 0, numpy.random.uniform(*a, **kw))

    def rand(self, *a):
        return self._note(0, numpy.random.rand(*a))


class _NP:
    def __init__(self, rec):
        self.random = rec

    def __getattr__(self, name):
        return getattr(numpy, name)


def load_reference(ref, rec):
    sys.path.insert(0, ref)
    from tactics2d.participant.trajectory import State           # the reference's own State
    path = os.path.join(ref, "tactics2d", "map", "generator", "generate_parking_lot.py")
    tree = ast.parse(open(path).read(), filename=path)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "ParkingLotGenerator")
    ns = {"np": _NP(rec), "Point": Point, "Polygon": Polygon, "affine_transform": affine_transform, "Area": Area, "Map": object,
          "State": State, "logging": logging, "time": time, "Tuple": tuple}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), ns)
    mp = os.path.join(ref, "tactics2d", "map", "element", "map.py")
    mtree = ast.parse(open(mp).read(), filename=mp)
    mcls = next(n for n in mtree.body if isinstance(n, ast.ClassDef) and n.name == "Map")
    add_area = next(n for n in mcls.body if isinstance(n, ast.FunctionDef) and n.name == "add_area")
    add_area.args.args[1].annotation = None
    mns = {"warnings": __import__("warnings"), "MapElement": types.SimpleNamespace(AREA="area")}
    exec(compile(ast.Module(body=[add_area], type_ignores=[]), mp, "exec"), mns)
    return ns["ParkingLotGenerator"], mns["add_area"], (cls.lineno, cls.end_lineno), (add_area.lineno, add_area.end_lineno)


class MapHolder:
    def __init__(self, add_area):
        self.name = self.scenario_type = None
        self.areas, self.ids = {}, {}
        self._boundary = None
        self._min_x = self._max_x = self._min_y = self._max_y = None
        self._add = add_area

    def add_area(self, area):
        self._add(self, area)

    def _add_element_to_spatial_index(self, *a):
        pass

    def _update_boundary_with_element(self, *a):
        pass

    def set_boundary(self, b):
        self._boundary = tuple(float(v) for v in b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--n", type=int, default=240)
    args = ap.parse_args()
    rec = _Recorder()
    Gen, add_area, l1, l2 = load_reference(args.ref, rec)
    print("executing generate_parking_lot.py lines %d-%d, map.py lines %d-%d" % (l1 + l2))
    import warnings
    out = {k: [] for k in ("seed", "type_proportion", "bay", "tape_off", "tape_kind", "tape_val", "area_off", "area_id", "area_quad",
                           "start", "target", "target_heading", "boundary")}
    out["tape_off"].append(0); out["area_off"].append(0)
    n_bay = 0
    for seed in range(args.n):
        tp = (0.5, 0.5, 0.5, 1.0, 0.0, 0.8)[seed % 6]
        numpy.random.seed(1000 + seed)
        rec.kind.clear(); rec.val.clear()
        gen = Gen((5.3, 2.5), tp)
        m = MapHolder(add_area)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            start, target_area, target_heading = gen.generate(m)
        out["seed"].append(1000 + seed); out["type_proportion"].append(tp); out["bay"].append(gen.mode == "bay"); n_bay += gen.mode == "bay"
        out["tape_kind"] += rec.kind; out["tape_val"] += rec.val; out["tape_off"].append(len(out["tape_kind"]))
        for key, area in m.areas.items():
            if area.subtype == "target_area":
                continue
            assert len(area.geometry.pts) == 4
            out["area_id"].append(int(key)); out["area_quad"].append(area.geometry.pts)
        out["area_off"].append(len(out["area_id"]))
        out["start"].append([float(start.x), float(start.y), float(start.heading)])
        out["target"].append(m.areas[0].geometry.pts)
        out["target_heading"].append(float(target_heading)); out["boundary"].append(m._boundary)
    os.makedirs(OUT, exist_ok=True)
    numpy.savez_compressed(os.path.join(OUT, "generator_replay.npz"),
                           seed=numpy.int32(out["seed"]), type_proportion=numpy.float64(out["type_proportion"]), bay=numpy.uint8(out["bay"]),
                           tape_off=numpy.int32(out["tape_off"]), tape_kind=numpy.int32(out["tape_kind"]), tape_val=numpy.float64(out["tape_val"]),
                           area_off=numpy.int32(out["area_off"]), area_id=numpy.int32(out["area_id"]), area_quad=numpy.float64(out["area_quad"]),
                           start=numpy.float64(out["start"]), target=numpy.float64(out["target"]),
                           target_heading=numpy.float64(out["target_heading"]), boundary=numpy.float64(out["boundary"]))
    lens = numpy.diff(out["tape_off"])
    print(f"{args.n} scenes ({n_bay} bay), {lens.sum()} draws (per scene {lens.min()} .. {lens.max()}), "
          f"{len(out['area_id'])} obstacle areas -> tests/golden/generator_replay.npz")


if __name__ == "__main__":
    main()
