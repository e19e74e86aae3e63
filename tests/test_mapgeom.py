"""tactics2d_amd.mapgeom: the reference's lane rings / area polygons cut into the convex polygons the event kernels take
(map/element/lane.py:125-130, map/element/map.py:92-167).  CPU tests: the pieces are convex, made of
the ring's own vertices, abut exactly -- and the off-lane predicate on the pieces (the oracle's union rule) equals `ring polygon
contains box` evaluated on the UNDIVIDED ring in exact rational arithmetic."""
from fractions import Fraction as Fr

import numpy as np
import pytest

from tactics2d_amd import mapgeom as MG


def _area2(P):
    P = np.asarray(P, np.float64)
    return float(np.dot(P[:, 0], np.roll(P[:, 1], -1)) - np.dot(np.roll(P[:, 0], -1), P[:, 1]))


def _convex_ccw(P):
    P = np.asarray(P, np.float64)
    n = len(P)
    return n >= 3 and _area2(P) > 0 and all(
        (P[(i + 1) % n, 0] - P[i, 0]) * (P[(i + 2) % n, 1] - P[i, 1]) - (P[(i + 1) % n, 1] - P[i, 1]) * (P[(i + 2) % n, 0] - P[i, 0]) >= 0
        for i in range(n))


def _curved_road(n_pts=40, radius=60.0, width=3.75, arc=1.3, wiggle=0.0, seed=0):
    """two lanes side by side along an arc, their shared rail sampled ONCE (so the lanes abut exactly); sides of a lane may
    carry different numbers of points (the outer rail is resampled)"""
    rng = np.random.default_rng(seed)
    t = np.linspace(0, arc, n_pts)
    r = radius + wiggle * np.sin(7 * t)
    rail = lambda off, tt=t: np.stack([(np.interp(tt, t, r) + off) * np.cos(tt), (np.interp(tt, t, r) + off) * np.sin(tt)], 1)
    mid = rail(0.0)
    inner = rail(-width, np.sort(np.concatenate([[0, arc], rng.uniform(0, arc, n_pts - 9)])))   # fewer points than the middle rail
    outer = rail(+width, np.linspace(0, arc, n_pts + 7))
    # direction of travel = increasing angle (counter-clockwise): the LEFT side is the inner rail
    return [(inner, mid), (mid, outer)]


def _exact_box_in_ring(box, ring):
    """closed polygon(ring) contains the convex box  <=>  area(box clipped to ... ) -- here: clip the RING by the box's four
    half-planes (Sutherland-Hodgman, exact rationals) and compare areas"""
    F = lambda P: [(Fr(float(x)), Fr(float(y))) for x, y in P]
    a2 = lambda P: sum(P[i][0] * P[(i + 1) % len(P)][1] - P[(i + 1) % len(P)][0] * P[i][1] for i in range(len(P)))
    B, R = F(box), F(np.float32(ring))
    if a2(B) < 0:
        B = B[::-1]
    if a2(R) < 0:
        R = R[::-1]
    out = R
    for k in range(4):
        a, b = B[k], B[(k + 1) % 4]
        side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        nxt = []
        for i in range(len(out)):
            p, q = out[i], out[(i + 1) % len(out)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                nxt.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                nxt.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
        out = nxt
        if len(out) < 3:
            return False
    return a2(out) == a2(B)


def test_lane_strip_pieces_are_convex_made_of_the_sides_own_points_and_tile_the_lane():
    for seed, wiggle in ((0, 0.0), (1, 1.5), (2, 3.0)):
        for left, right in _curved_road(wiggle=wiggle, seed=seed):
            pieces = MG.lanes_from_sides(left, right)
            pts = {tuple(p) for p in np.float32(np.concatenate([left, right]))}
            ring = np.float32(np.concatenate([left, right[::-1]]))
            assert len(pieces) >= 30
            for q in pieces:
                assert q.dtype == np.float32 and 3 <= len(q) <= 4 and _convex_ccw(q)
                assert all(tuple(v) in pts for v in q)
            assert abs(sum(_area2(q) for q in pieces) - abs(_area2(ring))) < 1e-6 * abs(_area2(ring))
            assert sum(len(q) == 4 for q in pieces) >= 0.6 * len(pieces)    # mostly quads: the strip follows the lane
            # every cut is shared by exactly two pieces, edge for edge (no T-junction): the outline left over is the ring
            edges = {}
            for q in pieces:
                for k in range(len(q)):
                    e = (tuple(q[k]), tuple(q[(k + 1) % len(q)]))
                    edges[e] = edges.get(e, 0) + 1
            open_edges = [e for e in edges if (e[1], e[0]) not in edges]
            ring_edges = {(tuple(ring[k]), tuple(ring[(k + 1) % len(ring)])) for k in range(len(ring))}
            ring_edges |= {(b, a) for a, b in ring_edges}
            assert all(e in ring_edges for e in open_edges) and all(v == 1 for v in edges.values())


def test_off_lane_on_the_pieces_equals_contains_on_the_undivided_ring(oracle):
    """a curved two-lane road with 40-point sides: `not union(pieces).contains(pose)` (the oracle's predicate, what the kernels
    evaluate) against exact `polygon(ring of both lanes).contains(pose)` on the undivided outline"""
    rng = np.random.default_rng(5)
    n = inside = 0
    for seed, wiggle in ((0, 0.0), (3, 2.0)):
        road = _curved_road(wiggle=wiggle, seed=seed)
        pieces = [q for left, right in road for q in MG.lanes_from_sides(left, right)]
        outline = np.concatenate([road[0][0], road[1][1][::-1]])     # inner rail forward, outer rail backward: both lanes
        for _ in range(140):
            ang = rng.uniform(-0.05, 1.35); rr = 60.0 + rng.uniform(-5.5, 5.5)
            x, y = np.float32(rr * np.cos(ang)), np.float32(rr * np.sin(ang))
            h = np.float32(ang + np.pi / 2 + rng.normal(0, 0.25))
            pose = oracle.pose_obb(float(x), float(y), float(h), rng.uniform(3.5, 5.0), rng.uniform(1.6, 2.0), trig=0)
            got = oracle.pose_in_lane_union(pose, (float(x), float(y)), pieces)
            want = _exact_box_in_ring(pose, outline)
            assert got == want, (seed, float(x), float(y), float(h), got, want)
            n += 1; inside += want
    assert n == 280 and 0.2 < inside / n < 0.8, (n, inside)


def test_ring_to_convex_on_non_convex_areas():
    comb = [(0, 0), (10, 0), (10, 6), (8, 6), (8, 2), (6, 2), (6, 6), (4, 6), (4, 2), (2, 2), (2, 6), (0, 6)]
    ell = [(0, 0), (6, 0), (6, 2), (2, 2), (2, 7), (0, 7)]
    star = [(np.cos(a) * (4 if k % 2 else 9), np.sin(a) * (4 if k % 2 else 9)) for k, a in enumerate(np.linspace(0, 2 * np.pi, 14, endpoint=False))]
    for ring in (comb, ell, star, comb[::-1], ell + [ell[0]]):
        for mv in (3, 4, 8):
            pieces = MG.ring_to_convex(ring, mv)
            assert all(_convex_ccw(q) and len(q) <= mv for q in pieces)
            assert abs(sum(_area2(q) for q in pieces) - abs(_area2(np.float32(ring)))) < 1e-4
        assert len(MG.ring_to_convex(ring, 8)) <= len(MG.ring_to_convex(ring, 3))
    sq = MG.ring_to_convex([(0, 0), (2, 0), (4, 0), (4, 3), (0, 3)], 8)   # a collinear vertex adds no area; it stays a (straight) corner
    assert len(sq) == 1 and abs(_area2(sq[0]) - 24.0) < 1e-6 and len(sq[0]) == 5
    with pytest.raises(ValueError):
        MG.ring_to_convex([(0, 0), (4, 4), (4, 0), (0, 4)], 8)             # a bow tie is not a simple ring
    with pytest.raises(ValueError):
        MG.ring_to_convex([(0, 0), (1, 1), (2, 2)], 8)
    assert len(MG.areas_to_convex([ell, comb], 8)) == len(MG.ring_to_convex(ell, 8)) + len(MG.ring_to_convex(comb, 8))


def test_rings_that_share_a_straight_side_with_collinear_points_keep_those_points(oracle):
    """two lanes given as RINGS only, side by side along a straight, densely sampled rail (the common case): the collinear
    points of the shared rail stay corners of the pieces on both sides (no T-junction: every piece edge on the rail is matched by
    the neighbour's reversed edge), straight corners are accepted as convex, and the union rule sees no seam"""
    xs = np.linspace(0.0, 40.0, 11)                                    # exactly representable collinear points
    rail = np.stack([xs, np.full_like(xs, 3.75)], 1)
    low = np.concatenate([np.stack([xs, np.zeros_like(xs)], 1), rail[::-1]])          # ring of the lower lane (CCW)
    up = np.concatenate([rail, np.stack([xs[::-1], np.full_like(xs, 7.5)], 1)])        # ring of the upper lane (CCW)
    pl, pu = MG.ring_to_convex(low, 8), MG.ring_to_convex(up, 8)
    rail_pts = {tuple(np.float32(p)) for p in rail}
    for pieces in (pl, pu):
        assert all(_convex_ccw(q) and 3 <= len(q) <= 8 for q in pieces)
        assert rail_pts <= {tuple(v) for q in pieces for v in q}       # none of the rail's points was dropped
    assert abs(sum(_area2(q) for q in pl + pu) - 2 * 40.0 * 7.5) < 1e-6
    on_rail = lambda e: e[0] in rail_pts and e[1] in rail_pts and e[0][1] == e[1][1] == np.float32(3.75)
    el = {(tuple(q[k]), tuple(q[(k + 1) % len(q)])) for q in pl for k in range(len(q))}
    eu = {(tuple(q[k]), tuple(q[(k + 1) % len(q)])) for q in pu for k in range(len(q))}
    rl, ru = {e for e in el if on_rail(e)}, {e for e in eu if on_rail(e)}
    assert len(rl) == len(ru) == 10 and {(b, a) for a, b in rl} == ru  # the seam: ten edges, each shared exactly
    # a body across the seam is on the lanes, wherever it sits along the rail; one over the outer edge is not
    for x in np.linspace(3.0, 37.0, 35):
        for h in (0.0, 0.3, np.pi / 2):
            pose = oracle.pose_obb(float(x), 3.75, float(h), 4.5, 1.8, trig=0)
            assert oracle.pose_in_lane_union(pose, (float(x), 3.75), pl + pu)
        pose = oracle.pose_obb(float(x), 7.0, 0.0, 4.5, 1.8, trig=0)
        assert not oracle.pose_in_lane_union(pose, (float(x), 7.0), pl + pu)
    # a ring with MORE straight corners on one ear than a piece may have: cut down, nothing dropped
    many = np.concatenate([np.stack([np.linspace(0, 30, 31), np.zeros(31)], 1), [(30.0, 5.0), (0.0, 5.0)]])
    for mv in (4, 8):
        pm = MG.ring_to_convex(many, mv)
        assert all(_convex_ccw(q) and 3 <= len(q) <= mv for q in pm) and abs(sum(_area2(q) for q in pm) - 300.0) < 1e-6
        assert {tuple(np.float32(p)) for p in many} <= {tuple(v) for q in pm for v in q}


def test_straight_stretches_collapse_to_single_quads_without_changing_a_verdict(oracle):
    """lanes_from_sides joins the pieces of a straight stretch (cut points exactly collinear on both sides): an axis-aligned
    40-point lanelet is ONE quad.  A neighbour that keeps a point on the long edge meets it in an exact T-junction; the union --
    and every off-lane verdict -- is what the unmerged pieces give."""
    xs = np.linspace(0.0, 78.0, 40)                                    # exactly representable, exactly collinear
    rail = lambda y: np.stack([xs, np.full_like(xs, y)], 1)
    a = MG.lanes_from_sides(rail(3.75), rail(0.0))                      # (left = +y side for travel along +x)
    b = MG.lanes_from_sides(rail(7.5), rail(3.75))
    assert len(a) == 1 and len(b) == 1 and len(a[0]) == 4 and abs(_area2(a[0]) - 2 * 78.0 * 3.75) < 1e-6
    # a branch that abuts on part of lane b's outer edge, with rail points of its own on it (a T-junction on a merged edge)
    bx = np.array([20.0, 24.0, 30.0])
    branch = MG.lanes_from_sides(np.stack([bx, np.full(3, 11.25)], 1), np.stack([bx, np.full(3, 7.5)], 1))
    assert len(branch) == 1
    merged = a + b + branch
    # the same roads cut at every rail point (what round 5 produced): 39 + 39 + 2 pieces
    fine = []
    for lo, hi in ((0.0, 3.75), (3.75, 7.5)):
        fine += [np.float32([(xs[i], lo), (xs[i + 1], lo), (xs[i + 1], hi), (xs[i], hi)]) for i in range(39)]
    fine += [np.float32([(bx[i], 7.5), (bx[i + 1], 7.5), (bx[i + 1], 11.25), (bx[i], 11.25)]) for i in range(2)]
    rng = np.random.default_rng(8)
    n_in = 0
    for _ in range(400):
        x, y, h = np.float32(rng.uniform(-3, 81)), np.float32(rng.uniform(-2, 13)), np.float32(rng.uniform(0, 2 * np.pi))
        pose = oracle.pose_obb(float(x), float(y), float(h), 4.5, 1.8, trig=0)
        got = oracle.pose_in_lane_union(pose, (float(x), float(y)), merged)
        assert got == oracle.pose_in_lane_union(pose, (float(x), float(y)), fine), (x, y, h)
        n_in += got
    for x in (20.0, 22.0, 24.0, 27.0, 30.0):                           # bodies across the T-junction seam, on every rail point
        pose = oracle.pose_obb(float(x), 7.5, np.pi / 2, 4.5, 1.2, trig=0)
        assert oracle.pose_in_lane_union(pose, (float(x), 7.5), merged) == (21.0 <= x <= 29.0 or x in (20.0, 30.0) and False)
    assert 60 < n_in < 340
    assert MG.geometry_budget(4, 64, lanes=[merged] * 4)["dwords_needed"] < MG.geometry_budget(4, 64, lanes=[fine] * 4)["dwords_needed"] / 8


def test_rail_simplification_is_shared_by_both_lanes_of_a_rail_and_bounded_by_its_tolerance(oracle):
    """simplify_tol: Douglas-Peucker on each side polyline, direction-independent -- the two lanes of a shared rail keep the same
    points and still abut edge for edge; a straight lanelet in ANY direction (fp32-rounded points are not exactly collinear)
    becomes one quad; a curved road keeps what its curvature needs; the outline moves by at most the tolerance."""
    t = np.linspace(0.0, 1.0, 40)[:, None]
    d = np.array([np.cos(0.37), np.sin(0.37)])
    nrm = np.array([-d[1], d[0]])
    mid, left, right = 5.0 + 80.0 * t * d, 5.0 + 80.0 * t * d + 3.75 * nrm, 5.0 + 80.0 * t * d - 3.75 * nrm
    assert len(MG.lanes_from_sides(left, mid)) > 20                     # fp32 rounding: not exactly collinear, nothing merges
    l0, l1 = MG.lanes_from_sides(left, mid, simplify_tol=1e-4), MG.lanes_from_sides(mid, right, simplify_tol=1e-4)
    assert len(l0) == 1 and len(l1) == 1 and len(l0[0]) == 4
    shared = {tuple(v) for v in l0[0]} & {tuple(v) for v in l1[0]}
    assert len(shared) == 2                                             # the rail's two end points, bit for bit
    assert np.array_equal(MG.simplify_polyline(mid, 1e-4), MG.simplify_polyline(mid[::-1], 1e-4)[::-1])
    # curved two-lane road (tests above): fewer pieces, still abutting exactly, outline within the tolerance
    for tol in (5e-3, 2e-2):
        road = _curved_road(n_pts=160, seed=4)                          # a point every 0.5 m: what a recorded map looks like
        fine = [q for l, r in road for q in MG.lanes_from_sides(l, r)]
        coarse = [q for l, r in road for q in MG.lanes_from_sides(l, r, simplify_tol=tol)]
        assert len(coarse) < 0.6 * len(fine) and all(_convex_ccw(q) and 3 <= len(q) <= 4 for q in coarse)
        edges = {}
        for q in coarse:
            for k in range(len(q)):
                e = (tuple(q[k]), tuple(q[(k + 1) % len(q)]))
                edges[e] = edges.get(e, 0) + 1
        assert all(v == 1 for v in edges.values())
        inner = [e for e in edges if (e[1], e[0]) in edges]
        assert len(inner) >= 2 * (len(coarse) - 2)                      # every cut and the whole middle rail are shared edges
        a_f, a_c = sum(_area2(q) for q in fine), sum(_area2(q) for q in coarse)
        perimeter = 2 * (60.0 * 1.3 + 7.5)
        assert abs(a_f - a_c) <= 2 * tol * perimeter
        rng = np.random.default_rng(3)
        for _ in range(120):                                            # verdicts agree away from the outline
            ang = rng.uniform(0.05, 1.25); rr = 60.0 + rng.uniform(-1.5, 1.5)
            x, y = np.float32(rr * np.cos(ang)), np.float32(rr * np.sin(ang))
            pose = oracle.pose_obb(float(x), float(y), float(ang + np.pi / 2), 4.0, 1.8, trig=0)
            assert oracle.pose_in_lane_union(pose, (float(x), float(y)), coarse) == oracle.pose_in_lane_union(pose, (float(x), float(y)), fine)


def test_map_boundary_and_the_duck_typed_map_adapter():
    assert MG.map_boundary() == (0.0, 0.0, 0.0, 0.0)
    assert MG.map_boundary([(0.2, -1.5), (3.7, 2.01)], [(-0.1, 0.0)]) == (-1.0, 4.0, -2.0, 3.0)   # floor / ceil, map.py:149-160

    class Geo:
        def __init__(self, c): self.coords = c
    class Poly:
        def __init__(self, c): self.exterior = Geo(c)
    class Lane:
        def __init__(self, left=None, right=None, ring=None):
            self.left_side = None if left is None else Geo(left)
            self.right_side = None if right is None else Geo(right)
            self.geometry = None if ring is None else Geo(ring)
    class Area:
        def __init__(self, c, subtype): self.geometry, self.subtype = Poly(c), subtype
    class Map:
        pass
    road = _curved_road(n_pts=12)
    m = Map()
    m.lanes = {"1": Lane(road[0][0] + [1000, 2000], road[0][1] + [1000, 2000]),
               "2": Lane(ring=np.concatenate([road[1][0], road[1][1][::-1]]) + [1000, 2000])}
    m.areas = {"a": Area([(1050, 2010), (1056, 2010), (1056, 2012), (1052, 2012), (1052, 2017), (1050, 2017), (1050, 2010)], "obstacle"),
               "b": Area([(1000, 2000), (1001, 2000), (1001, 2001)], "freespace")}
    out = MG.from_reference_map(m, origin=(1000, 2000))
    assert len(out["static"]) == len(MG.ring_to_convex([(50, 10), (56, 10), (56, 12), (52, 12), (52, 17), (50, 17)], 8))
    assert all(_convex_ccw(q) for q in out["lanes"] + out["static"]) and len(out["lanes"]) > 12
    assert np.abs(np.concatenate(out["lanes"])).max() < 256
    b = out["boundary"]
    assert b[0] <= 0.0 and b[2] <= 0.0 and b[1] >= 56.0 and all(float(v).is_integer() for v in b)


def test_geometry_budget_says_how_many_polygons_an_env_can_carry():
    road = _curved_road(n_pts=40)
    pieces = [q for left, right in road for q in MG.lanes_from_sides(left, right)]
    small = MG.geometry_budget(8, 64, lanes=[pieces[:40]] * 8)
    big = MG.geometry_budget(8, 64, lanes=[pieces] * 8)
    assert small["envs_per_workgroup"] == 1 and small["dwords_budget"] == 8192
    assert small["fits"] and small["dwords_needed"] < big["dwords_needed"]
    # the more envs share a workgroup (smaller max_agents), the less each can carry
    assert MG.geometry_budget(8, 8, lanes=[pieces[:40]] * 8)["dwords_needed"] > small["dwords_needed"]
    huge = MG.geometry_budget(2, 64, lanes=[pieces * 6] * 2)
    assert not huge["fits"]
    from tactics2d_amd._ffi import GeometryError
    with pytest.raises(GeometryError):
        MG.geometry_budget(1, 64, static=[[np.float32([(0, 0), (4, 4), (4, 0), (0, 4)])]])   # not convex


def test_random_lanes_tile_exactly_and_agree_with_the_undivided_ring(oracle):
    """40 random lanes -- S-curves and arcs, 5..60 points per side, the two sides sampled independently, either direction of
    travel, widths 2.5..6 m -- cut into strips: pieces convex and of the sides' own points, areas add up, and the union
    predicate on the pieces equals exact containment in the undivided ring for poses scattered over and around the lane."""
    rng = np.random.default_rng(2024)
    n_pose = inside = 0
    for case in range(40):
        nl, nr = int(rng.integers(5, 60)), int(rng.integers(5, 60))
        width = rng.uniform(2.5, 6.0); length = rng.uniform(30, 120)
        k1, k2 = rng.uniform(-0.02, 0.02), rng.uniform(-3e-4, 3e-4)     # curvature and its rate: arcs and S-curves

        def centre(s):
            th = k1 * s + 0.5 * k2 * s * s
            # integrate the heading numerically (fine grid) for the position
            g = np.linspace(0, 1, 400)[None, :] * s[:, None]
            thg = k1 * g + 0.5 * k2 * g * g
            x = np.trapezoid(np.cos(thg), g, axis=1); y = np.trapezoid(np.sin(thg), g, axis=1)
            return np.stack([x, y], 1), th

        def side(n, off):
            s = np.sort(np.concatenate([[0.0, length], rng.uniform(0, length, n - 2)]))
            c, th = centre(s)
            return c + off * np.stack([-np.sin(th), np.cos(th)], 1)
        left, right = side(nl, +width / 2), side(nr, -width / 2)
        if case % 3 == 0:
            left, right = right[::-1], left[::-1]                        # travelling the other way: sides swap and reverse
        rot = rng.uniform(0, 2 * np.pi); R = np.array([[np.cos(rot), -np.sin(rot)], [np.sin(rot), np.cos(rot)]])
        shift = rng.uniform(-50, 50, 2)
        left, right = left @ R.T + shift, right @ R.T + shift
        pieces = MG.lanes_from_sides(left, right)
        ring = np.float32(np.concatenate([left, right[::-1]]))
        pts = {tuple(p) for p in ring}
        assert all(_convex_ccw(q) and 3 <= len(q) <= 4 and all(tuple(v) in pts for v in q) for q in pieces), case
        assert abs(sum(_area2(q) for q in pieces) - abs(_area2(ring))) < 1e-5 * abs(_area2(ring)), case
        assert MG.geometry_budget(1, 8, lanes=[pieces])["dwords_needed"] > 0      # the library's own checks accept every piece
        for _ in range(6):
            i = int(rng.integers(0, len(left)))
            c = 0.5 * (left[i] + right[min(i * len(right) // len(left), len(right) - 1)]) + rng.normal(0, width * 0.4, 2)
            x, y, h = np.float32(c[0]), np.float32(c[1]), np.float32(rng.uniform(0, 6.28))
            pose = oracle.pose_obb(float(x), float(y), float(h), rng.uniform(1.0, 4.5), rng.uniform(0.8, 2.0), trig=0)
            want = _exact_box_in_ring(pose, ring)
            assert oracle.pose_in_lane_union(pose, (float(x), float(y)), pieces) == want, (case, float(x), float(y), float(h))
            n_pose += 1; inside += want
    assert n_pose == 240 and 0.1 < inside / n_pose < 0.9, (n_pose, inside)


def test_map_boundary_equals_the_reference_property_executed():
    """tests/golden/map_boundary.json: Map.boundary (map/element/map.py:92-167) EXECUTED by oracle/gen_golden_tables.py on 40
    maps of plain data holders (nodes, lane and road-line polylines, area exteriors; coordinates of 1e1 .. 5e3 m, rounded to 0-5
    decimals; the empty map; integer coordinates) -- mapgeom.map_boundary and the duck-typed adapter give the same four numbers"""
    import json, os, types
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "map_boundary.json")))
    assert len(cases) == 40
    for c in cases:
        sets = [c["nodes"]] + c["lanes"] + c["areas"] + c["roadlines"]
        assert list(MG.map_boundary(*[s for s in sets if len(s)])) == c["boundary"], c
    assert cases[0]["boundary"] == [0.0, 0.0, 0.0, 0.0] and cases[1]["boundary"] == [2.0, 2.0, -3.0, -3.0]
