"""From the reference's map elements to the geometry the event kernels take.

The reference keeps a lane as a ring -- `Lane.geometry = LinearRing(left_side + reversed(right_side))`
(map/element/lane.py:125-130): many vertices, not convex -- an area as an arbitrary polygon (map/element/area.py) and the map
boundary as floor / ceil of the extreme coordinates (map/element/map.py:92-167).  The kernels take, per env, CONVEX polygons of
3..8 vertices in one packed LDS record (include/t2d.h: t2d_set_static_geometry / t2d_set_lane_geometry), and the off-lane flag
is `not union(lane polygons).contains(pose)` with the rule that polygons which abut share their vertices exactly.

This module cuts the reference's rings into such polygons WITHOUT inventing a coordinate: every piece is made of the ring's
own (fp32-rounded) vertices and every cut runs from one of them to another, so neighbouring pieces share whole edges bit for
bit, and the union of the pieces is the ring.  Host-side numpy only (runs once per reset, like `OffLane.reset`).

    lanes_from_sides(left_xy, right_xy)   one lane -> a strip of abutting convex quads / triangles
    ring_to_convex(ring_xy, max_verts)    any simple ring -> convex pieces of <= max_verts vertices (ear clipping + merging)
    areas_to_convex(polys, max_verts)     the same for a list of polygons
    map_boundary(point sets)              Map.boundary's floor / ceil rule
    from_reference_map(map_)              duck-typed: lanes / obstacle areas / boundary of a tactics2d `Map`
    geometry_budget(...)                  how much of the 32 KiB record a scene takes, before it is installed
"""
import ctypes as C

import numpy as np


# ---------------------------------------------------------------------------------------------------------------- predicates
def _f32(xy):
    """the coordinates as the pool stores them: fp32, kept as fp64 values (their differences are then exact and an orientation
    is two products and one subtraction, each rounded once: the same evaluation -- and the same sign, down to areas of ~1e-13 m^2
    -- as the convexity check of t2d_set_*_geometry, which is what decides whether a piece is accepted)"""
    a = np.asarray(xy, np.float64).reshape(-1, 2)
    return a.astype(np.float32).astype(np.float64)


def _orient(a, b, c):
    return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0])


def _area2(P):
    x, y = P[:, 0], P[:, 1]
    return float(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


def _is_convex(P, idx):
    """every turn of the CCW index polygon is left or straight, and it has area (what t2d_set_*_geometry accepts)"""
    n = len(idx)
    if n < 3:
        return False
    for i in range(n):
        if _orient(P[idx[i]], P[idx[(i + 1) % n]], P[idx[(i + 2) % n]]) < 0.0:
            return False
    return _area2(P[list(idx)]) > 0.0


def _in_triangle(p, a, b, c):
    return _orient(a, b, p) >= 0.0 and _orient(b, c, p) >= 0.0 and _orient(c, a, p) >= 0.0


def _dedup(P):
    keep = [0]
    for i in range(1, len(P)):
        if not np.array_equal(P[i], P[keep[-1]]):
            keep.append(i)
    if len(keep) > 1 and np.array_equal(P[keep[-1]], P[keep[0]]):   # a closed ring repeats its first vertex
        keep.pop()
    return P[keep]


# ---------------------------------------------------------------------------------------------------------------- ear clipping
def _ear_clip(P):
    """convex pieces (index tuples, CCW) of the simple CCW ring P: its ear triangles, each with the ring's COLLINEAR vertices
    that lie on its edges kept as straight corners -- a vertex on a straight run adds no area, but a neighbouring ring that
    shares the polyline still uses it, and pieces that abut must share their vertices exactly (no T-junctions)"""
    idx = list(range(len(P)))
    tris = []
    on_chord = {}          # (u, v), neighbours in the shrinking ring -> the collinear vertices dropped between them, in order
    guard = 0

    def run(u, v):         # u, the straight corners between u and v, (v excluded)
        return [u] + on_chord.pop((u, v), [])

    while len(idx) > 3:
        n = len(idx)
        clipped = False
        for k in range(n):
            i0, i1, i2 = idx[(k - 1) % n], idx[k], idx[(k + 1) % n]
            o = _orient(P[i0], P[i1], P[i2])
            if o == 0.0:               # collinear: the vertex lies on the chord; it stays a corner of the piece that gets the chord
                on_chord[(i0, i2)] = on_chord.pop((i0, i1), []) + [i1] + on_chord.pop((i1, i2), [])
                idx.pop(k)
                clipped = True
                break
            if o < 0.0:
                continue               # reflex corner
            a, b, c = P[i0], P[i1], P[i2]
            if any(j not in (i0, i1, i2) and _in_triangle(P[j], a, b, c) and not (np.array_equal(P[j], a) or np.array_equal(P[j], b) or np.array_equal(P[j], c))
                   for j in idx):
                continue
            tris.append(tuple(run(i0, i1) + run(i1, i2) + [i2]))   # (the new chord i2 -> i0 is a cut: nothing lies on it)
            idx.pop(k)
            clipped = True
            break
        guard += 1
        if not clipped or guard > 4 * len(P) + 16:
            raise ValueError("ring is not a simple polygon (self-intersecting or degenerate): cannot be cut into convex pieces")
    if len(idx) == 3 and _orient(P[idx[0]], P[idx[1]], P[idx[2]]) > 0.0:
        tris.append(tuple(run(idx[0], idx[1]) + run(idx[1], idx[2]) + run(idx[2], idx[0])))
    return tris


def _split_large(P, poly, max_verts):
    """a convex piece with more than max_verts vertices (straight corners count) -> pieces of <= max_verts, cut along chords
    between its own vertices; both sides of every cut keep an area"""
    poly = list(poly)
    n = len(poly)
    if n <= max_verts:
        return [poly]
    best = None
    for i in range(n):
        for d in range(2, n - 1):
            a = [poly[(i + t) % n] for t in range(d + 1)]
            b = [poly[(i + d + t) % n] for t in range(n - d + 1)]
            if _area2(P[a]) > 0.0 and _area2(P[b]) > 0.0:
                score = max(len(a), len(b))
                if best is None or score < best[0]:
                    best = (score, a, b)
    if best is None:
        raise ValueError("cannot cut a piece of %d vertices down to %d" % (n, max_verts))
    return _split_large(P, best[1], max_verts) + _split_large(P, best[2], max_verts)


def _strict_corners(P, idx):
    """the vertices of the convex index polygon that are real corners (a turn), in order"""
    n = len(idx)
    return [idx[i] for i in range(n) if _orient(P[idx[i - 1]], P[idx[i]], P[idx[(i + 1) % n]]) != 0.0]


def _merge(P, polys, max_verts, strict=False):
    """Hertel-Mehlhorn style: join two pieces across a shared edge while the result stays convex and small enough.
    strict=True counts only the REAL corners of the joined piece against max_verts (its straight corners -- vertices lying
    exactly on an edge -- are kept while merging, so that further neighbours still find their shared edges, and dropped by
    the caller at the end): consecutive quads of a straight strip become one quad."""
    polys = [list(p) for p in polys]
    changed = True
    while changed:
        changed = False
        edges = {}
        for pi, p in enumerate(polys):
            for k in range(len(p)):
                edges[(p[k], p[(k + 1) % len(p)])] = (pi, k)
        for (a, b), (pi, ka) in list(edges.items()):
            other = edges.get((b, a))
            if other is None:
                continue
            pj, kb = other
            if pi == pj or (not strict and len(polys[pi]) + len(polys[pj]) - 2 > max_verts):
                continue
            p, q = polys[pi], polys[pj]
            # p runs ... a, b ...; q runs ... b, a ...: walk p from b round to a, then q from a round to b (ends dropped)
            m = [p[(ka + 1 + t) % len(p)] for t in range(len(p))] + [q[(kb + 2 + t) % len(q)] for t in range(len(q) - 2)]
            # straight corners at the two joints are kept (shared vertices stay shared); the piece must be convex
            if _is_convex(P, m) and (not strict or len(_strict_corners(P, m)) <= max_verts):
                polys[pi] = m
                polys.pop(pj)
                changed = True
                break
    return polys


def ring_to_convex(ring_xy, max_verts=8):
    """A simple ring (closed or not, either winding) -> list of convex CCW polygons, float32 (n, 2), 3 <= n <= max_verts, made
    of the ring's own vertices; their union is the ring's polygon and neighbours share whole edges."""
    if not 3 <= max_verts <= 8:
        raise ValueError("max_verts must be 3..8 (T2D_MAX_POLY_VERTS)")
    P = _dedup(_f32(ring_xy))
    if len(P) < 3:
        raise ValueError("a ring needs at least 3 distinct vertices")
    a2 = _area2(P)
    if a2 == 0.0:
        raise ValueError("ring has no area")
    if a2 < 0.0:
        P = P[::-1].copy()
    pieces = []
    for t in _ear_clip(P):              # (an ear with many straight corners on its edges is cut down first)
        pieces.extend(_split_large(P, t, max_verts))
    polys = _merge(P, pieces, max_verts)
    return [np.float32(P[list(p)]) for p in polys]


def areas_to_convex(polys, max_verts=8):
    """[ring (n, 2), ...] -> flat list of convex pieces (map/element/area.py geometries: `area.geometry.exterior.coords`)"""
    out = []
    for ring in polys:
        out.extend(ring_to_convex(ring, max_verts))
    return out


# ---------------------------------------------------------------------------------------------------------------- lanes
def simplify_polyline(xy, tol):
    """Douglas-Peucker on a side polyline: drops points that lie within `tol` metres of the chord of the points kept (end points
    always stay).  Direction-independent -- the polyline is walked from its lexicographically smaller end -- so two lanes that
    share a rail (map/element/lane.py:117-130: the left side of one is the right side of the other) keep the SAME points and
    still abut exactly.  tol = 0 drops only points that are exactly collinear in fp32 coordinates."""
    P = _dedup(_f32(xy))
    if len(P) <= 2:
        return np.float32(P)
    flip = tuple(P[0]) > tuple(P[-1])
    if flip:
        P = P[::-1]
    keep = np.zeros(len(P), bool)
    keep[0] = keep[-1] = True
    stack = [(0, len(P) - 1)]
    while stack:
        a, b = stack.pop()
        if b - a < 2:
            continue
        d = P[b] - P[a]
        L = float(np.hypot(d[0], d[1]))
        q = P[a + 1:b] - P[a]
        # distance to the chord's line inside its span, to the nearer end point beyond it
        t = np.clip((q @ d) / (L * L), 0.0, 1.0) if L > 0.0 else np.zeros(len(q))
        dist = np.hypot(q[:, 0] - t * d[0], q[:, 1] - t * d[1])
        k = int(np.argmax(dist))
        exact0 = tol == 0.0 and all(_orient(P[a], P[a + 1 + j], P[b]) == 0.0 and 0.0 < t[j] < 1.0 for j in range(len(q)))
        if (tol > 0.0 and dist[k] <= tol) or exact0:
            continue
        if tol == 0.0:      # split at the first point that is not exactly on the chord
            k = next(j for j in range(len(q)) if not (_orient(P[a], P[a + 1 + j], P[b]) == 0.0 and 0.0 < t[j] < 1.0))
        keep[a + 1 + k] = True
        stack += [(a, a + 1 + k), (a + 1 + k, b)]
    out = P[keep]
    return np.float32(out[::-1] if flip else out)


def lanes_from_sides(left_xy, right_xy, max_verts=4, simplify_tol=None):
    """One lane of the reference -- its two side polylines, same direction of travel (Lane.left_side / right_side,
    map/element/lane.py:117-130) -- as a strip of exactly-abutting convex polygons (quads where the sides allow, triangles where
    one side has more points than the other): consecutive cross cuts L[i] - R[j] advance along whichever side keeps the cut
    short, so the pieces follow the lane instead of fanning out from one corner.  Falls back to ear clipping of the ring
    left + reversed(right) when a cut would leave the lane (sides that fold back).  Returns float32 arrays (n, 2), CCW.
    Consecutive pieces of a STRAIGHT stretch -- cut points exactly collinear on both sides, in the fp32 coordinates the pool
    stores -- are joined and lose their straight corners: an axis-aligned 40-point lanelet is one quad, not 39 (the union, hence
    every off-lane verdict, is unchanged; a neighbour that keeps a point on such an edge meets it in an exact T-junction,
    which the lane-union boundary walk of t2d_set_lane_geometry covers by construction: collinear, opposite direction).
    simplify_tol (metres, None = off): first drop the side points within that distance of the chord of their neighbours
    (simplify_polyline: the same result for both lanes of a shared rail) -- a straight lanelet in any direction becomes one quad,
    a 60-m curve keeps a point every ~0.7 m at 1 mm; the lane's outline moves by at most that much."""
    if simplify_tol is not None:
        left_xy, right_xy = simplify_polyline(left_xy, simplify_tol), simplify_polyline(right_xy, simplify_tol)
    Lp, Rp = _dedup(_f32(left_xy)), _dedup(_f32(right_xy))
    if len(Lp) < 2 or len(Rp) < 2:
        raise ValueError("each side needs at least 2 distinct points")
    ring = np.concatenate([Lp, Rp[::-1]])
    sign = np.sign(_area2(ring))
    if sign == 0.0:
        raise ValueError("lane has no area")
    nL, nR = len(Lp), len(Rp)
    P = np.concatenate([Lp, Rp])           # indices: left i, right nL + j
    tris, i, j, ok = [], 0, 0, True
    while i < nL - 1 or j < nR - 1:
        adv_left = j == nR - 1 or (i < nL - 1 and np.sum((Lp[i + 1] - Rp[j]) ** 2) <= np.sum((Lp[i] - Rp[j + 1]) ** 2))
        t = (i, nL + j, i + 1) if adv_left else (i, nL + j, nL + j + 1)
        o = _orient(P[t[0]], P[t[1]], P[t[2]])
        if o * sign > 0.0:                 # (the ring runs left forward, right backward: a valid piece turns the other way round)
            ok = False
            break
        if o != 0.0:
            tris.append(t if o > 0.0 else (t[0], t[2], t[1]))
        if adv_left:
            i += 1
        else:
            j += 1
    # (pieces that all turn the right way tile the lane iff their areas add up to the ring's: a cut that crossed the far side
    # would count some ground twice)
    if ok and tris:
        total = sum(_orient(P[a], P[b], P[c]) for a, b, c in tris)
        ok = abs(total - abs(_area2(ring))) <= 1e-9 * abs(_area2(ring))
    if not ok or not tris:
        return ring_to_convex(ring, max(max_verts, 3))
    pieces = _merge(P, _merge(P, tris, max_verts), max_verts, strict=True)
    return [np.float32(P[_strict_corners(P, p)]) for p in pieces]


# ---------------------------------------------------------------------------------------------------------------- boundary
def map_boundary(*point_sets):
    """Map.boundary (map/element/map.py:92-167): (floor(min x), ceil(max x), floor(min y), ceil(max y)) over every coordinate
    given -- nodes, lane rings, area exteriors, road lines; (0, 0, 0, 0) for an empty map.  Tuple order = OutBound's."""
    pts = [np.asarray(p, np.float64).reshape(-1, 2) for p in point_sets if p is not None and len(p)]
    if not pts:
        return (0.0, 0.0, 0.0, 0.0)
    a = np.concatenate(pts)
    return (float(np.floor(a[:, 0].min())), float(np.ceil(a[:, 0].max())), float(np.floor(a[:, 1].min())), float(np.ceil(a[:, 1].max())))


def _coords(geom):
    if geom is None:
        return None
    ext = getattr(geom, "exterior", None)
    c = getattr(ext if ext is not None else geom, "coords", geom)
    a = np.asarray(list(c), np.float64)
    return a[:, :2] if a.size else None


def from_reference_map(map_, origin=(0.0, 0.0), obstacle_subtypes=("obstacle",), max_lane_verts=4, max_area_verts=8, simplify_tol=None):
    """Flatten a tactics2d `Map` (duck-typed: `.lanes`, `.areas`, optionally `.nodes` / `.roadlines`, dicts of elements with the
    reference's attributes) for ONE env: returns dict(lanes=[convex polys], static=[convex polys], boundary=(xmin, xmax, ymin,
    ymax)).  `origin` is subtracted first: the pool stores env-local fp32 coordinates and wants |x|, |y| < 256 m.
    Lanes with both sides use lanes_from_sides, the others their `geometry` ring; areas whose subtype is in
    `obstacle_subtypes` become static obstacles (envs/parking.py:416-420 passes exactly those to StaticCollision)."""
    o = np.asarray(origin, np.float64)
    lanes, static, pts = [], [], []
    for lane in getattr(map_, "lanes", {}).values():
        left, right = _coords(getattr(lane, "left_side", None)), _coords(getattr(lane, "right_side", None))
        if left is not None and right is not None:
            lanes.extend(lanes_from_sides(left - o, right - o, max_lane_verts, simplify_tol))
            pts += [left - o, right - o]
        else:
            ring = _coords(getattr(lane, "geometry", None))
            if ring is not None:
                lanes.extend(ring_to_convex(ring - o, max(max_lane_verts, 3)))
                pts.append(ring - o)
    for area in getattr(map_, "areas", {}).values():
        ring = _coords(getattr(area, "geometry", None))
        if ring is None:
            continue
        pts.append(ring - o)
        if getattr(area, "subtype", None) in obstacle_subtypes:
            static.extend(ring_to_convex(ring - o, max_area_verts))
    for node in getattr(map_, "nodes", {}).values():
        pts.append(np.array([[node.x, node.y]], np.float64) - o)
    for line in getattr(map_, "roadlines", {}).values():
        c = _coords(getattr(line, "geometry", None))
        if c is not None:
            pts.append(c - o)
    return dict(lanes=lanes, static=static, boundary=map_boundary(*pts))


# ---------------------------------------------------------------------------------------------------------------- capacity
def geometry_budget(n_env, max_agents, static=None, lanes=None):
    """What t2d_set_static_geometry + t2d_set_lane_geometry would need of the 32 KiB per-workgroup record for these scenes
    (per-env lists of convex polygons, or CSR triples), WITHOUT a device: dict(dwords_needed, dwords_budget, envs_per_workgroup,
    fits, tier).  Raises GeometryError for polygons the library rejects (not convex, degenerate, > 8 vertices)."""
    from . import _ffi
    from .traffic import polygons_to_csr

    def csr(t):
        if t is None:
            return None, None, None
        eo, vo, xy = t if isinstance(t, tuple) else polygons_to_csr(t)
        return (np.ascontiguousarray(eo, np.int32), np.ascontiguousarray(vo, np.int32), np.ascontiguousarray(xy, np.float32))

    p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    s, l = csr(static), csr(lanes)
    need, budget, epb = C.c_int32(), C.c_int32(), C.c_int32()
    _ffi.check(_ffi.lib().t2d_geometry_budget(int(n_env), int(max_agents), p(s[0]), p(s[1]), p(s[2]), p(l[0]), p(l[1]), p(l[2]),
                                                   C.byref(need), C.byref(budget), C.byref(epb)))
    fits = need.value <= budget.value
    # (a scene that does not fit the LDS record is not refused: t2d_set_*_geometry keeps it in the HBM grid tier -- one uniform
    # grid per env, tactics2d_amd/csrc/t2d_mapgrid.hip -- and the pool steps as integrate -> map events -> events + status)
    return dict(dwords_needed=need.value, dwords_budget=budget.value, envs_per_workgroup=epb.value, fits=fits,
                tier="lds_record" if fits else "hbm_grid")
