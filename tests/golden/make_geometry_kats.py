#!/usr/bin/env python3
"""Hand-built geometry known-answer tests -> geometry_kats.json.

The expected flags are written down BY HAND from the documented shapely semantics the reference
relies on (closed-set `intersects`: shared edge / corner / containment count; `contains`: touching
the boundary from inside is still contained) -- they are NOT computed by the oracle or the kernels.
Coordinates are chosen exactly representable in fp32 with headings 0 (cos = 1, sin = 0 exactly),
so that "touching" cases are exact.  Flag bits: 1 dynamic, 2 static, 4 out-of-bound, 8 off-lane.
"""
import json
import os

import numpy as np

P4 = float(np.float32(np.pi / 4))


def row(shape, L, W):
    r = [0.0] * 24
    r[0] = 2 if shape == 1 else 0; r[1] = 1.2; r[2] = 1.3; r[3] = 2.5; r[17] = 5
    r[18] = shape; r[19] = L; r[20] = W
    return r


ROWS = [row(0, 4.0, 2.0), row(0, 2.0, 2.0), row(1, 1.0, 1.0)]  # box 4x2, box 2x2, circle r = 0.5


def scene(name, parts, flags, static=None, lanes=None, boundary=None):
    return dict(name=name, rows=ROWS, x=[p[0] for p in parts], y=[p[1] for p in parts],
                heading=[p[2] for p in parts], type_id=[p[3] for p in parts], static=static or [],
                lanes=lanes or [], boundary=boundary, flags=flags)


S = [
    scene("disjoint boxes", [(0, 0, 0, 0), (10, 0, 0, 0)], [0, 0]),
    scene("overlapping boxes", [(0, 0, 0, 0), (3, 0, 0, 0)], [1, 1]),
    scene("shared edge (touch)", [(0, 0, 0, 0), (4, 0, 0, 0)], [1, 1]),
    scene("shared corner (touch)", [(0, 0, 0, 0), (4, 2, 0, 0)], [1, 1]),
    scene("near miss, gap ~1e-6", [(0, 0, 0, 0), (4.000001, 0, 0, 0)], [0, 0]),
    scene("nested, inner touches outer edges", [(0, 0, 0, 0), (0, 0, 0, 1)], [1, 1]),
    scene("strictly nested", [(0, 0, 0, 0), (0.5, 0, 0, 2)], [1, 1]),
    scene("rotated 45deg miss", [(0, 0, P4, 1), (2.5, 0, 0, 1)], [0, 0]),
    scene("rotated 45deg hit", [(0, 0, P4, 1), (2.375, 0, 0, 1)], [1, 1]),
    scene("three in a row, middle hits both", [(0, 0, 0, 0), (3.5, 0, 0, 0), (7, 0, 0, 0)], [1, 1, 1]),
    scene("three in a row, ends apart", [(0, 0, 0, 0), (4.5, 0, 0, 0), (9.5, 0, 0, 0)], [0, 0, 0]),
    scene("circle touches box edge", [(0, 0, 0, 0), (2.5, 0, 0, 2)], [1, 1]),
    scene("circle misses box edge", [(0, 0, 0, 0), (2.625, 0, 0, 2)], [0, 0]),
    scene("circle near box corner hit", [(0, 0, 0, 0), (2.25, 1.25, 0, 2)], [1, 1]),
    scene("circle near box corner miss", [(0, 0, 0, 0), (2.5, 1.5, 0, 2)], [0, 0]),
    scene("circle inside box", [(0, 0, 0, 0), (1.0, 0.25, 0, 2)], [1, 1]),
    scene("circles touch", [(0, 0, 0, 2), (1, 0, 0, 2)], [1, 1]),
    scene("circles apart", [(0, 0, 0, 2), (1.125, 0, 0, 2)], [0, 0]),
    scene("box touches static triangle vertex (cw triangle)", [(3, 0, 0, 0)], [2],
          static=[[[5, -1], [6, 3], [7, -1]]]),
    scene("box clear of static triangle", [(2.5, 0, 0, 0)], [0], static=[[[5, -1], [7, -1], [6, 3]]]),
    scene("box inside big static quad", [(0, 0, 0, 0)], [2], static=[[[-8, -8], [8, -8], [8, 8], [-8, 8]]]),
    scene("static hexagon edge touch", [(0, 0, 0, 1)], [2],
          static=[[[1, -1], [3, -2], [5, -1], [5, 1], [3, 2], [1, 1]]]),
    scene("circle vs static quad touch", [(0, 0, 0, 2)], [2], static=[[[0.5, -1], [2, -1], [2, 1], [0.5, 1]]]),
    scene("circle vs static quad miss", [(0, 0, 0, 2)], [0], static=[[[0.625, -1], [2, -1], [2, 1], [0.625, 1]]]),
    scene("tight boundary: touching from inside is contained", [(0, 0, 0, 0)], [0], boundary=[-2, 2, -1, 1]),
    scene("boundary crossed by 1/1024", [(0, 0, 0, 0)], [4], boundary=[-2, 1.9990234375, -1, 1]),
    scene("far outside boundary", [(50, 0, 0, 0)], [4], boundary=[-10, 10, -10, 10]),
    scene("circle tight in boundary", [(0, 0, 0, 2)], [0], boundary=[-0.5, 0.5, -0.5, 0.5]),
    scene("circle out of boundary", [(0.125, 0, 0, 2)], [4], boundary=[-0.5, 0.5, -0.5, 0.5]),
    scene("box exactly fills lane", [(0, 0, 0, 0)], [0], lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]]]),
    scene("box pokes out of lane", [(0, 0.125, 0, 0)], [8], lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]]]),
    scene("box straddles two adjacent lanes", [(0, 1, 0, 0)], [0],
          lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]], [[-5, 1], [5, 1], [5, 3], [-5, 3]]]),
    scene("pedestrian centre on the lane's outer edge: half the disc is outside", [(0, 1, 0, 2)], [8],
          lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]]]),
    scene("pedestrian centre off lane", [(0, 1.125, 0, 2)], [8], lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]]]),
    scene("pedestrian disc touches the lane edge from inside: contained", [(0, 0.5, 0, 2)], [0],
          lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]]]),
    scene("pedestrian disc pokes out by 1/8", [(0, 0.625, 0, 2)], [8], lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]]]),
    scene("pedestrian centre on the shared edge of two lanes", [(0, 1, 0, 2)], [0],
          lanes=[[[-5, -1], [5, -1], [5, 1], [-5, 1]], [[-5, 1], [5, 1], [5, 3], [-5, 3]]]),
    scene("pedestrian over a gap of 1/8 between two lanes", [(0.0625, 0, 0, 2)], [8],
          lanes=[[[-5, -1], [0, -1], [0, 1], [-5, 1]], [[0.125, -1], [5, -1], [5, 1], [0.125, 1]]]),
    # ---- off-lane = not union(lanes).contains(pose): the edge half.  Crossing roads as diamonds
    # |x - y| <= 2 and |x + y| <= 2 (exact coordinates), reflex corners of the union at (+-2, 0), (0, +-2).
    scene("all four vertices in lanes, right edge cuts the corner of two crossing roads", [(2, 0, 0, 1)], [8],
          lanes=[[[-9, -11], [11, 9], [9, 11], [-11, -9]], [[-11, 9], [9, -11], [11, -9], [-9, 11]]]),
    scene("edge passes exactly through the reflex corner of the union: contained", [(1, 0, 0, 1)], [0],
          lanes=[[[-9, -11], [11, 9], [9, 11], [-11, -9]], [[-11, 9], [9, -11], [11, -9], [-9, 11]]]),
    scene("inside the crossing of the two roads", [(0, 0, 0, 1)], [0],
          lanes=[[[-9, -11], [11, 9], [9, 11], [-11, -9]], [[-11, 9], [9, -11], [11, -9], [-9, 11]]]),
    scene("body edge runs along the outer boundary across two abutting lanes (edge on edge)", [(0, 0, 0, 0)], [0],
          lanes=[[[-5, -1], [0, -1], [0, 1], [-5, 1]], [[0, -1], [5, -1], [5, 1], [0, 1]]]),
    scene("gap of 1/8 between two lanes, all vertices in lanes", [(0, 0, 0, 0)], [8],
          lanes=[[[-5, -1], [0, -1], [0, 1], [-5, 1]], [[0.125, -1], [5, -1], [5, 1], [0.125, 1]]]),
    scene("chain of three lanes covers the long edges", [(0, 0, 0, 0)], [0],
          lanes=[[[-6, -1], [-1, -1], [-1, 1], [-6, 1]], [[-1, -1], [1, -1], [1, 1], [-1, 1]],
                 [[1, -1], [6, -1], [6, 1], [1, 1]]]),
    scene("chain with the middle lane missing", [(0, 0, 0, 0)], [8],
          lanes=[[[-6, -1], [-1, -1], [-1, 1], [-6, 1]], [[1, -1], [6, -1], [6, 1], [1, 1]]]),
    scene("middle lane covers only the lower half: upper long edge leaves the union", [(0, 0, 0, 0)], [8],
          lanes=[[[-6, -1], [-1, -1], [-1, 1], [-6, 1]], [[-1, -1], [1, -1], [1, 0], [-1, 0]],
                 [[1, -1], [6, -1], [6, 1], [1, 1]]]),
    scene("overlapping lanes", [(0, 0, 0, 0)], [0],
          lanes=[[[-6, -1], [0.5, -1], [0.5, 1], [-6, 1]], [[-0.5, -1], [6, -1], [6, 1], [-0.5, 1]]]),
    scene("nested lanes: body crosses the inner lane's boundary inside the outer lane", [(0, 0, 0, 0)], [0],
          lanes=[[[-6, -3], [6, -3], [6, 3], [-6, 3]], [[-1, -1], [1, -1], [1, 1], [-1, 1]]]),
    scene("vertex on the shared corner of four lanes", [(2, 1, 0, 0)], [0],
          lanes=[[[-4, -4], [0, -4], [0, 0], [-4, 0]], [[0, -4], [4, -4], [4, 0], [0, 0]],
                 [[0, 0], [4, 0], [4, 4], [0, 4]], [[-4, 0], [0, 0], [0, 4], [-4, 4]]]),
    scene("frame of four lanes: the hole of the union lies inside the body", [(2, 2, 0, 0)], [8],
          lanes=[[[0, 0], [4, 0], [4, 1], [0, 1]], [[0, 1], [1, 1], [1, 4], [0, 4]],
                 [[3, 1], [4, 1], [4, 4], [3, 4]], [[0, 3], [4, 3], [4, 4], [0, 4]]]),
    scene("body exactly fills the gap between two lanes (vertices on both, interior outside)", [(0, 0, 0, 0)], [8],
          lanes=[[[-5, -3], [5, -3], [5, -1], [-5, -1]], [[-5, 1], [5, 1], [5, 3], [-5, 3]]]),
    scene("body is exactly the hole of a frame of four lanes", [(2, 2, 0, 1)], [8],
          lanes=[[[0, 0], [4, 0], [4, 1], [0, 1]], [[0, 1], [1, 1], [1, 3], [0, 3]],
                 [[3, 1], [4, 1], [4, 3], [3, 3]], [[0, 3], [4, 3], [4, 4], [0, 4]]]),
    scene("pentagon lane + triangle lane sharing an edge (clockwise triangle)", [(0, 0, 0, 1)], [0],
          lanes=[[[-3, -2], [0, -2], [0, 2], [-3, 2], [-4, 0]], [[0, -2], [0, 2], [4, 0]]]),
    scene("everything at once", [(0, 0, 0, 0), (3, 0.5, 0, 0), (30, 0, 0, 2)], [1 | 2, 1 | 8, 4 | 8],
          static=[[[-3, -3], [-1, -3], [-1, -0.5], [-3, -0.5]]], lanes=[[[-6, -1], [6, -1], [6, 1], [-6, 1]]],
          boundary=[-20, 20, -20, 20]),
]

# pairwise predicate KATs (polygons given directly, CCW): name, A, B, intersects
PAIRS = [
    ("identical squares", [[0, 0], [1, 0], [1, 1], [0, 1]], [[0, 0], [1, 0], [1, 1], [0, 1]], True),
    ("vertex on edge", [[0, 0], [2, 0], [2, 2], [0, 2]], [[2, 1], [3, 0], [3, 2]], True),
    ("vertex just off edge", [[0, 0], [2, 0], [2, 2], [0, 2]], [[2.0000002, 1], [3, 0], [3, 2]], False),
    ("triangle in square", [[0, 0], [4, 0], [4, 4], [0, 4]], [[1, 1], [2, 1], [1, 2]], True),
    ("cross (no vertex inside the other)", [[-3, -1], [3, -1], [3, 1], [-3, 1]],
     [[-1, -3], [1, -3], [1, 3], [-1, 3]], True),
    ("diagonal separation only", [[0, 0], [2, 0], [0, 2]], [[2, 2], [2, 0.5], [0.5, 2]], False),
    ("diagonal touch", [[0, 0], [2, 0], [0, 2]], [[2, 2], [2, 0], [0, 2]], True),
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "geometry_kats.json")
    with open(out, "w") as f:
        json.dump(dict(scenes=S, pairs=[dict(name=n, A=a, B=b, intersects=i) for n, a, b, i in PAIRS]), f, indent=1)
    print(out, len(S), "scenes", len(PAIRS), "pairs")
