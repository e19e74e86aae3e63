"""The oracle's geometry / event predicates.  PARITY UNPINNED against the reference's engine
(shapely/GEOS is not available and the reference's tests pin no predicate result); pinned instead
by hand-derived KATs and an independent cross-check against matplotlib.path.  CPU only."""
import numpy as np
import pytest

import helpers as H


def _scene_from_kat(k):
    return dict(rows=np.array(k["rows"]), n_env=1, A=len(k["x"]), x=np.float32(k["x"]), y=np.float32(k["y"]),
                heading=np.float32(k["heading"]), type_id=np.array(k["type_id"], np.uint8),
                active=np.ones(len(k["x"]), np.uint8),
                static=H.to_csr([[np.float32(q) for q in k["static"]]]) if k["static"] else None,
                lanes=H.to_csr([[np.float32(q) for q in k["lanes"]]]) if k["lanes"] else None,
                boundary=np.float32([k["boundary"]]) if k["boundary"] else None, boundary_valid=None)


@pytest.mark.parametrize("trig", [0, 1])
def test_hand_built_scene_kats(oracle, trig):
    for k in H.load_json("geometry_kats.json")["scenes"]:
        f, e = H.oracle_collide(oracle, _scene_from_kat(k), trig)
        assert f.tolist() == k["flags"], (k["name"], f.tolist(), k["flags"])
        assert e[0] == np.bitwise_or.reduce(np.array(k["flags"], np.uint32))


def test_pairwise_kats_and_symmetry(oracle):
    for p in H.load_json("geometry_kats.json")["pairs"]:
        assert oracle.convex_intersects(p["A"], p["B"]) == p["intersects"], p["name"]
        assert oracle.convex_intersects(p["B"], p["A"]) == p["intersects"], p["name"]


def test_deterministic_sincos_is_within_one_ulp_of_numpy(oracle):
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.uniform(-7, 7, 4000), rng.uniform(-2000, 2000, 2000),
                         np.float64(np.float32(rng.uniform(0, 2 * np.pi, 2000))), [0.0, np.pi / 2, np.pi, 2 * np.pi]])
    worst = 0.0
    for x in xs:
        s, c = oracle.sincos(x)
        worst = max(worst, abs(s - np.sin(x)) / np.spacing(abs(np.sin(x))), abs(c - np.cos(x)) / np.spacing(abs(np.cos(x))))
    assert worst <= 1.0 + 1e-9, worst
    assert oracle.sincos(0.0) == (0.0, 1.0)


def test_pose_follows_the_reference_vertex_order_and_transform(oracle):
    """vehicle.py:132-142 / :263-281: (+L/2,-W/2), (+L/2,+W/2), (-L/2,+W/2), (-L/2,-W/2), matrix
    [cos, -sin, sin, cos, x, y]."""
    x, y, h, L_, W_ = 3.0, -2.0, 0.7, 4.284, 1.799
    v = oracle.pose_obb(x, y, h, L_, W_, trig=1)
    base = np.array([[L_ / 2, -W_ / 2], [L_ / 2, W_ / 2], [-L_ / 2, W_ / 2], [-L_ / 2, -W_ / 2]])
    want = np.stack([np.cos(h) * base[:, 0] - np.sin(h) * base[:, 1] + x,
                     np.sin(h) * base[:, 0] + np.cos(h) * base[:, 1] + y], 1)
    assert np.abs(v - want).max() < 1e-15
    assert np.abs(oracle.pose_obb(x, y, h, L_, W_, trig=0) - want).max() < 1e-14
    # counter-clockwise
    area2 = sum(v[i, 0] * v[(i + 1) % 4, 1] - v[(i + 1) % 4, 0] * v[i, 1] for i in range(4))
    assert area2 > 0


def test_intersects_agrees_with_matplotlib_path(oracle):
    """Independent, non-reference cross-check: matplotlib.path.Path.intersects_path(filled=True) has
    closed-set overlap/containment semantics like shapely `intersects`.  Random
    convex quads; pairs within 1e-9 of touching are excluded (different arithmetic there)."""
    mpath = pytest.importorskip("matplotlib.path")
    rng = np.random.default_rng(11)
    n_checked = n_true = 0
    for _ in range(1500):
        A, B = H.random_quads(rng, 2, (-4, 4), (-4, 4), size=(1.5, 5.0))
        A = np.float64(A); B = np.float64(B)
        if not (oracle.polygon_is_convex(np.float32(A)) and oracle.polygon_is_convex(np.float32(B))):
            continue

        def ccw(P):
            a2 = sum(P[i, 0] * P[(i + 1) % 4, 1] - P[(i + 1) % 4, 0] * P[i, 1] for i in range(4))
            return P if a2 > 0 else P[::-1].copy()
        A, B = ccw(A), ccw(B)
        got = oracle.convex_intersects(A, B)
        pa = mpath.Path(np.vstack([A, A[:1]]), closed=True); pb = mpath.Path(np.vstack([B, B[:1]]), closed=True)
        want = bool(pa.intersects_path(pb, filled=True))
        # robustness margin: shrink/grow B by 1e-7 must not flip the answer, else skip (near touching)
        cB = B.mean(0)
        g1 = oracle.convex_intersects(A, cB + (B - cB) * (1 + 1e-7)); g2 = oracle.convex_intersects(A, cB + (B - cB) * (1 - 1e-7))
        if g1 != g2:
            continue
        n_checked += 1; n_true += got
        assert got == want
    assert n_checked > 1000 and 0.2 < n_true / n_checked < 0.9


def test_flags_do_not_depend_on_libm_vs_deterministic_trig(oracle):
    """The bit-exact flag definition uses the deterministic sin/cos; the reference would use numpy's.
    Their vertices differ by <= 1 ulp, so flags agree on every random scene here."""
    rng = np.random.default_rng(3)
    diff = tot = 0
    for (n_env, A, extent) in ((40, 64, (60.0, 16.0)), (60, 8, (20.0, 12.0)), (300, 1, (30.0, 20.0))):
        sc = H.random_scene(rng, n_env, A, extent, n_static=6, n_lanes=3)
        f0, _ = H.oracle_collide(oracle, sc, 0); f1, _ = H.oracle_collide(oracle, sc, 1)
        diff += int((f0 != f1).sum()); tot += f0.size
    assert diff == 0 and tot > 3000


def test_convexity_check(oracle):
    assert oracle.polygon_is_convex(np.float32([[0, 0], [2, 0], [2, 2], [0, 2]]))
    assert oracle.polygon_is_convex(np.float32([[0, 2], [2, 2], [2, 0], [0, 0]]))        # clockwise is fine
    assert not oracle.polygon_is_convex(np.float32([[0, 0], [4, 0], [1, 1], [0, 4]]))    # dart
    assert not oracle.polygon_is_convex(np.float32([[0, 0], [1, 1], [2, 2]]))            # degenerate


def test_status_priority_order(oracle):
    """Early-return order of _ParkingScenarioManager.check_status (envs/parking.py:361-392)."""
    cfg = oracle.make_config(max_step=3, check_dynamic=1, check_off_lane=1)
    flags = np.array([0, 4, 2, 1, 8, 4 | 2 | 1, 2 | 1], np.uint32)
    cnt = np.zeros(7, np.int32); frame = np.zeros(7, np.int32)
    st, rw = oracle.status(cfg, 7, 1, flags, 100, cnt, frame)
    assert st[:, :2].tolist() == [[1, 1], [4, 1], [6, 3], [6, 4], [6, 6], [4, 1], [6, 3]]
    assert st[:, 2].tolist() == [0] * 7 and st[:, 3].tolist() == [0, 1, 1, 1, 1, 1, 1]
    assert rw[1] == -5 and rw[2] == -5 and abs(rw[0] + np.tanh(1 / 3) * 0.001) < 1e-9
    cnt[:] = 3
    st, rw = oracle.status(cfg, 7, 1, flags, 100, cnt, frame)   # cnt 4 > max_step 3: time exceed wins
    assert (st[:, 0] == 3).all() and (rw == -1).all() and (cnt == 4).all() and (frame == 200).all()


def test_oracle_batch_loops_do_not_depend_on_thread_count(oracle):
    """bench.py's all-cores CPU baseline runs the same batch entry points with OpenMP threads."""
    import helpers as H
    from tactics2d_amd import scenarios as S
    sc = S.mixed(24, 16, seed=5)
    rng = np.random.default_rng(0)
    a0, a1 = sc.sample_actions(rng)
    vx = np.float32(sc.speed * np.cos(sc.heading)); vy = np.float32(sc.speed * np.sin(sc.heading))
    res = []
    for th in (1, 4):
        oracle.set_threads(th)
        try:
            o = oracle.integrate(sc.rows, sc.x, sc.y, sc.heading, sc.speed, vx, vy, a0, a1, sc.type_id, sc.active, 100)
            f, ef = oracle.collide(sc.rows, sc.n_env, sc.A, np.float32(o[:, 0]), np.float32(o[:, 1]), np.float32(o[:, 2]),
                                   sc.type_id, sc.active, sc.static, sc.boundary, None, sc.lanes, 1)
        finally:
            oracle.set_threads(1)
        res.append((o.copy(), f.copy(), ef.copy()))
    assert np.array_equal(res[0][0], res[1][0], equal_nan=True)
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def _exact_convex_intersects(A, B):
    """Closed-set `intersects` of two convex CCW polygons in EXACT rational arithmetic: they are disjoint iff some
    edge of one has every vertex of the other strictly on its outer side (separating-axis theorem for convex sets;
    touching and containment count).  This is what GEOS's robust predicates compute for shapely's `intersects`."""
    from fractions import Fraction
    FA = [(Fraction(float(x)), Fraction(float(y))) for x, y in A]
    FB = [(Fraction(float(x)), Fraction(float(y))) for x, y in B]

    def separates(P, Q):
        n = len(P)
        for i in range(n):
            (px, py), (qx, qy) = P[i], P[(i + 1) % n]
            if (px, py) == (qx, qy):
                continue
            if all((qx - px) * (ry - py) - (qy - py) * (rx - px) < 0 for rx, ry in Q):
                return True
        return False
    return not (separates(FA, FB) or separates(FB, FA))


def test_intersects_against_exact_rational_arithmetic(oracle):
    """Pins the fp64 orientation-sign SAT of the oracle (= the GPU's predicate) against exact arithmetic on the very
    same binary64 inputs: random convex quads, and adversarial families that touch exactly (shared edge segments,
    vertex on edge, shared corner -- dyadic coordinates, so 'touching' is exact) or miss by one fp64 ulp."""
    rng = np.random.default_rng(2024)
    n_checked = n_touch = 0

    def check(A, B):
        nonlocal n_checked
        want = _exact_convex_intersects(A, B)
        got = bool(oracle.convex_intersects(np.float64(A), np.float64(B)))
        assert got == want, (A, B, got, want)
        n_checked += 1
        return want

    def quad(cx, cy, L, W, th):
        c, s = np.cos(th), np.sin(th)
        loc = np.array([[L / 2, -W / 2], [L / 2, W / 2], [-L / 2, W / 2], [-L / 2, -W / 2]])
        return loc @ np.array([[c, s], [-s, c]]) + [cx, cy]

    for _ in range(1500):          # generic position: boxes of vehicle size at vehicle distances
        A = quad(rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(2, 6), rng.uniform(1, 2.5), rng.uniform(0, 6.3))
        d = rng.uniform(0, 7); a = rng.uniform(0, 6.3)
        B = quad(A[:, 0].mean() + d * np.cos(a), A[:, 1].mean() + d * np.sin(a), rng.uniform(2, 6), rng.uniform(1, 2.5),
                 rng.uniform(0, 6.3))
        check(A, B)
    for _ in range(600):           # exactly touching, dyadic coordinates: axis-parallel boxes sharing an edge piece / a corner
        x0, y0 = rng.integers(-400, 400, 2) / 8.0
        w, h_ = rng.integers(1, 40, 2) / 8.0
        A = np.array([[x0, y0], [x0 + w, y0], [x0 + w, y0 + h_], [x0, y0 + h_]])
        kind = rng.integers(0, 3)
        w2, h2 = rng.integers(1, 40, 2) / 8.0
        if kind == 0:    # shared edge segment (to the right of A)
            y1 = y0 + rng.integers(-8, 8) / 8.0
            B = np.array([[x0 + w, y1], [x0 + w + w2, y1], [x0 + w + w2, y1 + h2], [x0 + w, y1 + h2]])
            expect = (y1 <= y0 + h_) and (y1 + h2 >= y0)
        elif kind == 1:  # shared corner only
            B = np.array([[x0 + w, y0 + h_], [x0 + w + w2, y0 + h_], [x0 + w + w2, y0 + h_ + h2], [x0 + w, y0 + h_ + h2]])
            expect = True
        else:            # a vertex of a 45-degree diamond exactly on A's right edge
            cy = y0 + rng.integers(0, int(h_ * 8) + 1) / 8.0
            r = rng.integers(1, 24) / 8.0
            B = np.array([[x0 + w, cy], [x0 + w + r, cy - r], [x0 + w + 2 * r, cy], [x0 + w + r, cy + r]])
            expect = True
        n_touch += 1
        assert check(A, B) == expect
        # ... and the same pair pulled apart by one ulp must not intersect, pushed together must
        Bm = B.copy(); Bm[:, 0] = np.nextafter(Bm[:, 0], np.inf)
        if kind != 0 or expect:
            assert check(A, Bm) is False or kind == 0 and not expect
    for _ in range(400):           # near-collinear edges at vehicle scale: rotated copies nudged along the normal by a few ulps
        th = rng.uniform(0, 6.3)
        A = quad(rng.uniform(-100, 100), rng.uniform(-100, 100), 4.5, 1.8, th)
        n = np.array([-np.sin(th), np.cos(th)])
        B = quad(A[:, 0].mean() + 1.8 * n[0], A[:, 1].mean() + 1.8 * n[1], 4.5, 1.8, th)   # side by side, ~touching
        k = int(rng.integers(-3, 4))
        B = B + n * k * np.spacing(100.0)
        check(A, B)
    assert n_checked > 3000 and n_touch == 600


# ------------------------------------------------------------------ off-lane = not union(lanes).contains(pose)
def _exact_pose_in_union(pose, lanes):
    """P subset of the closed union U of convex polygons, in EXACT rational arithmetic and by a route that shares
    nothing with the oracle's (which walks the boundary of U): subtract every lane from P by cutting along its edge
    lines -- what is cut off outside an edge line stays, what is left inside all of them is discarded -- and P is
    contained iff no piece of positive area survives (P is the closure of its interior, U is closed)."""
    from fractions import Fraction as Fr

    def F(P):
        return [(Fr(float(x)), Fr(float(y))) for x, y in P]

    def area2(P):
        return sum(P[i][0] * P[(i + 1) % len(P)][1] - P[(i + 1) % len(P)][0] * P[i][1] for i in range(len(P)))

    def cut(P, a, b):
        """split the convex polygon P by the directed line a -> b: (left-or-on part, strictly-right part)"""
        def side(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        left, right = [], []
        n = len(P)
        for i in range(n):
            p, q = P[i], P[(i + 1) % n]
            sp, sq = side(p), side(q)
            if sp >= 0:
                left.append(p)
            if sp <= 0:
                right.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                x = (p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1]))
                left.append(x); right.append(x)
        return left, right

    pieces = [F(pose)]
    for L in lanes:
        L = F(np.float32(L))
        if area2(L) < 0:
            L = L[::-1]
        nxt = []
        for piece in pieces:
            rest = piece
            for j in range(len(L)):
                if len(rest) < 3:
                    break
                rest, out = cut(rest, L[j], L[(j + 1) % len(L)])
                if len(out) >= 3 and area2(out) > 0:
                    nxt.append(out)
        pieces = nxt
    return not pieces


def _strip(cx, cy, th, length, width):
    c, s = np.cos(th), np.sin(th)
    loc = np.array([[length / 2, -width / 2], [length / 2, width / 2], [-length / 2, width / 2], [-length / 2, -width / 2]])
    return np.float32(loc @ np.array([[c, s], [-s, c]]) + [cx, cy])


def test_contains_against_exact_rational_arithmetic(oracle):
    """Pins the oracle's `union(lanes).contains(box)` (centre in the union + no boundary piece of the union inside the
    open box) against exact rational arithmetic on the same binary64 inputs, three families: random overlapping strips
    (crossing roads, corners cut with all four vertices in lanes, partial overlaps), lanes that abut exactly (shared
    fp32 vertices: straight multi-lane roads and polygonal rings, bodies straddling the shared edges), and frames with
    a hole."""
    rng = np.random.default_rng(77)
    stats = dict(n=0, inside=0, cut_corner=0)

    def check(lanes, x, y, h, L_, W_):
        x, y, h = float(np.float32(x)), float(np.float32(y)), float(np.float32(h))
        pose = oracle.pose_obb(x, y, h, L_, W_, trig=0)
        got = oracle.pose_in_lane_union(pose, (x, y), lanes)
        want = _exact_pose_in_union(pose, lanes)
        assert got == want, (lanes, x, y, h, L_, W_, got, want)
        stats["n"] += 1; stats["inside"] += want
        if not want and all(any(oracle.point_in_convex(oracle.ccw(np.float64(np.float32(q))), v) for q in lanes) for v in pose):
            stats["cut_corner"] += 1        # the case the vertex-only rule of round 1 missed

    for _ in range(500):       # random strips through a common area
        lanes = [_strip(rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(0, np.pi), rng.uniform(15, 40), rng.uniform(3, 8))
                 for _ in range(int(rng.integers(1, 5)))]
        for _ in range(4):
            check(lanes, rng.uniform(-9, 9), rng.uniform(-9, 9), rng.uniform(0, 6.3), rng.uniform(2, 5), rng.uniform(1, 2.2))
    for _ in range(150):       # straight road of abutting lanes (shared fp32 vertices), arbitrary direction
        th = rng.uniform(0, np.pi); n_l = int(rng.integers(2, 5)); w = 3.75
        c, s = np.cos(th), np.sin(th)
        rails = [np.float32([[-40 * c - o * s, -40 * s + o * c], [40 * c - o * s, 40 * s + o * c]])
                 for o in (np.arange(n_l + 1) - n_l / 2) * w]
        lanes = [np.float32([rails[k][0], rails[k][1], rails[k + 1][1], rails[k + 1][0]]) for k in range(n_l)]
        for _ in range(6):
            o = rng.uniform(-n_l * w / 2 - 1, n_l * w / 2 + 1); a = rng.uniform(-30, 30)
            check(lanes, a * c - o * s, a * s + o * c, th + rng.normal(0, 0.2), rng.uniform(3, 5), rng.uniform(1.5, 2.0))
    for _ in range(100):       # polygonal ring of trapezoids sharing their radial edges + one arm
        nseg = int(rng.integers(6, 14)); r_in, r_out = 12.0, 20.0
        ang = 2 * np.pi * np.arange(nseg + 1) / nseg
        ri = np.float32(np.stack([r_in * np.cos(ang), r_in * np.sin(ang)], 1)); ro = np.float32(np.stack([r_out * np.cos(ang), r_out * np.sin(ang)], 1))
        ri[-1], ro[-1] = ri[0], ro[0]
        lanes = [np.float32([ri[k], ro[k], ro[k + 1], ri[k + 1]]) for k in range(nseg)]
        lanes.append(np.float32([[19, -3.75], [45, -3.75], [45, 3.75], [19, 3.75]]))
        for _ in range(6):
            a = rng.uniform(0, 6.3); rr = rng.uniform(11, 22)
            check(lanes, rr * np.cos(a), rr * np.sin(a), a + np.pi / 2 + rng.normal(0, 0.3), rng.uniform(3, 5), rng.uniform(1.5, 2.0))
    for _ in range(100):       # frames: four strips around a hole that may fit inside a body
        g = rng.uniform(0.3, 3.0); t = rng.uniform(0.5, 3.0); o = g + t
        lanes = [np.float32([[-o, -o], [o, -o], [o, -g], [-o, -g]]), np.float32([[-o, g], [o, g], [o, o], [-o, o]]),
                 np.float32([[-o, -g], [-g, -g], [-g, g], [-o, g]]), np.float32([[g, -g], [o, -g], [o, g], [g, g]])]
        for _ in range(4):
            check(lanes, rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0, 6.3), rng.uniform(1, 6), rng.uniform(1, 3))
    assert stats["n"] > 3500 and 0.15 < stats["inside"] / stats["n"] < 0.85 and stats["cut_corner"] > 30, stats


def test_lane_boundary_of_abutting_lanes_is_their_outline(oracle):
    """Four 3.75 m lanes side by side: the boundary of the union is the outer rectangle -- 2 long sides and 2 x 4
    end caps -- and none of the three shared lane lines; crossing roads keep only the outline of the cross."""
    ys = [-7.5, -3.75, 0.0, 3.75, 7.5]
    lanes = [np.float32([[-210, ys[k]], [210, ys[k]], [210, ys[k + 1]], [-210, ys[k + 1]]]) for k in range(4)]
    pieces, owner = oracle.lane_boundary(lanes)
    assert len(pieces) == 10
    horiz = pieces[pieces[:, 1] == pieces[:, 3]]
    assert sorted(set(horiz[:, 1].tolist())) == [-7.5, 7.5]
    assert np.isclose(np.abs(pieces[:, 2:] - pieces[:, :2]).sum(), 2 * 420 + 2 * 15)
    cross = [np.float32([[-60, -3.75], [60, -3.75], [60, 3.75], [-60, 3.75]]),
             np.float32([[-3.75, -60], [3.75, -60], [3.75, 60], [-3.75, 60]])]
    pieces, owner = oracle.lane_boundary(cross)
    assert len(pieces) == 12 and np.isclose(np.abs(pieces[:, 2:] - pieces[:, :2]).sum(), 8 * 56.25 + 4 * 7.5)
    # overlapping duplicates of one polygon: the outline once from each copy's point of view is covered by the other
    dup = [np.float32([[0, 0], [4, 0], [4, 2], [0, 2]])] * 2
    pieces, _ = oracle.lane_boundary(dup)
    assert len(pieces) == 8       # collinear same-direction edges do not cover each other: both outlines stay


# ---------------------------------------------------------------------------------------------------------------------
# Certificates of the step kernel (round 3): short cuts that must never contradict the oracle's predicate.
def _safe_rects(lanes):
    """t2d_lane_safe_rects for one env (host-only entry point of libt2d_hip.so: no device is touched)."""
    import ctypes as C
    from tactics2d_amd import _ffi, layout as L
    lib = _ffi.lib()
    vo = np.zeros(len(lanes) + 1, np.int32)
    vo[1:] = np.cumsum([len(q) for q in lanes])
    xy = np.ascontiguousarray(np.concatenate([np.float32(q).reshape(-1, 2) for q in lanes]), np.float32)
    eo = np.array([0, len(lanes)], np.int32)
    out = np.zeros((L.SAFE_RECTS, 4), np.float32)
    rc = lib.t2d_lane_safe_rects(1, eo.ctypes.data_as(C.c_void_p), vo.ctypes.data_as(C.c_void_p),
                                       xy.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def _kernel_pose_box(lo_x, hi_x, lo_y, hi_y):
    """The step kernel's outward-rounded fp32 box of a pose (t2d_collide.hip, pose phase), operation by operation."""
    f = np.float32
    def down(v):
        b = f(v)
        return f(b - f(f(abs(b) * f(1.2e-7)) + f(1e-6)))
    def up(v):
        b = f(v)
        return f(b + f(f(abs(b) * f(1.2e-7)) + f(1e-6)))
    return down(lo_x), up(hi_x), down(lo_y), up(hi_y)


def _certified(rects, box):
    lo_x, hi_x, lo_y, hi_y = box
    return any(r[0] <= lo_x and hi_x <= r[1] and r[2] <= lo_y and hi_y <= r[3] for r in rects)


def test_safe_rectangles_of_the_metric_scenes():
    """What build_safe_rects finds: a highway's four abutting lanes merge into the carriageway, crossing roads stay two
    rectangles (fillet triangles contribute nothing), a roundabout contributes its four arms and not its ring."""
    from tactics2d_amd import scenarios as S
    sc = S.mixed(3, 64, seed=3)
    eo, vo, xy = sc.lanes
    per_env = [[xy[vo[p]:vo[p + 1]] for p in range(eo[e], eo[e + 1])] for e in range(3)]
    hw, rb, ix = (_safe_rects(l) for l in per_env)
    fin = lambda r: r[np.isfinite(r).all(1)]
    assert len(fin(hw)) == 1 and np.allclose(fin(hw)[0], [-210, 210, -7.5, 7.5], atol=2e-4)
    assert (fin(hw)[0] > np.float32([-210, 209, -7.5, 7.4])).all() and fin(hw)[0][1] < 210 and fin(hw)[0][3] < 7.5  # shrunk inwards
    assert len(fin(rb)) == 4 and len(fin(ix)) == 2
    assert sorted(np.round(fin(ix)[:, 3] - fin(ix)[:, 2]).tolist()) == [8.0, 238.0] or len(fin(ix)) == 2


def test_safe_rectangle_certificate_never_contradicts_the_oracle(oracle):
    """A pose whose kernel box lies in a safe rectangle must be `contained` for the oracle -- boxes and discs, poses pushed
    against the rectangles' edges and corners, on lane sets with everything the merge meets: abutting and overlapping
    rectangular lanes, rectangles inside others, diagonal strips crossing them, fillets, rings."""
    rng = np.random.default_rng(2024)
    n_cert = n_box = n_disc = 0

    def scenes():
        ys = [-7.5, -3.75, 0.0, 3.75, 7.5]
        yield [np.float32([[-210, ys[k]], [210, ys[k]], [210, ys[k + 1]], [-210, ys[k + 1]]]) for k in range(4)]
        cross = [np.float32([[-60, -3.75], [60, -3.75], [60, 3.75], [-60, 3.75]]), np.float32([[-3.75, -60], [3.75, -60], [3.75, 60], [-3.75, 60]])]
        for sx in (-1, 1):
            for sy in (-1, 1):
                cross.append(np.float32([[sx * 3.75, sy * 3.75], [sx * 7.75, sy * 3.75], [sx * 3.75, sy * 7.75]]))
        yield cross
        for _ in range(40):   # random stacks: abutting / overlapping / nested rectangles + a diagonal strip + a triangle
            x0, x1 = sorted(np.float32(rng.uniform(-40, 40, 2)))
            if x1 - x0 < 6:
                x1 = np.float32(x0 + 6)
            lanes, y = [], np.float32(rng.uniform(-10, 0))
            for _ in range(int(rng.integers(1, 5))):
                h = np.float32(rng.uniform(2.5, 5)); mode = rng.integers(0, 3)
                y_lo = y if mode == 0 else np.float32(y - rng.uniform(0, 1.5))      # abut exactly / overlap
                lanes.append(np.float32([[x0, y_lo], [x1, y_lo], [x1, y_lo + h], [x0, y_lo + h]])); y = np.float32(y_lo + h)
            if rng.uniform() < 0.5:      # a rectangle inside the stack
                lanes.append(np.float32([[x0 + 1, lanes[0][0][1] + 0.5], [x1 - 1, lanes[0][0][1] + 0.5], [x1 - 1, lanes[0][0][1] + 1.5], [x0 + 1, lanes[0][0][1] + 1.5]]))
            if rng.uniform() < 0.7:
                lanes.append(_strip(rng.uniform(x0, x1), rng.uniform(-8, 8), rng.uniform(0.2, 2.9), rng.uniform(10, 50), rng.uniform(2, 6)))
            if rng.uniform() < 0.5:
                lanes.append(np.float32([[x1, y - 3], [x1 + 4, y - 3], [x1, y]]))
            if rng.uniform() < 0.5:      # clockwise input
                lanes[0] = lanes[0][::-1].copy()
            yield lanes

    for lanes in scenes():
        rects = _safe_rects(lanes)
        rects = rects[np.isfinite(rects).all(1)]
        assert len(rects) >= 1
        for _ in range(60):
            r = rects[int(rng.integers(0, len(rects)))]
            disc = rng.uniform() < 0.2
            h = float(np.float32(rng.uniform(0, 6.3) if rng.uniform() < 0.7 else rng.choice([0.0, np.pi / 2, np.pi, 1e-4])))
            L_, W_ = (rng.uniform(2, 9), rng.uniform(1, 2.5)) if not disc else (0.0, rng.uniform(0.4, 1.0))
            ex = 0.5 * (L_ * abs(np.cos(h)) + W_ * abs(np.sin(h))) if not disc else 0.5 * W_
            ey = 0.5 * (L_ * abs(np.sin(h)) + W_ * abs(np.cos(h))) if not disc else 0.5 * W_
            if r[1] - r[0] <= 2 * ex + 1e-3 or r[3] - r[2] <= 2 * ey + 1e-3:
                continue
            # centres: anywhere inside, or pushed to within micrometres of an edge / a corner of the rectangle
            def coord(lo, hi, e):
                m = rng.integers(0, 4)
                d = rng.choice([0.0, 1e-6, 3e-6, 1e-5, 1e-3])
                return rng.uniform(lo + e, hi - e) if m == 0 else (lo + e + d if m == 1 else (hi - e - d if m == 2 else 0.5 * (lo + hi)))
            x = float(np.float32(coord(float(r[0]), float(r[1]), ex))); y = float(np.float32(coord(float(r[2]), float(r[3]), ey)))
            if disc:
                box = _kernel_pose_box(x - 0.5 * W_, x + 0.5 * W_, y - 0.5 * W_, y + 0.5 * W_)
            else:
                pose = oracle.pose_obb(x, y, h, L_, W_, trig=1)
                box = _kernel_pose_box(pose[:, 0].min(), pose[:, 0].max(), pose[:, 1].min(), pose[:, 1].max())
            if not _certified(rects, box):
                continue
            n_cert += 1
            if disc:
                n_disc += 1
                assert oracle.circle_in_lane_union((x, y), 0.5 * W_, lanes), (lanes, x, y, W_)
            else:
                n_box += 1
                assert oracle.pose_in_lane_union(pose, (x, y), lanes), (lanes, x, y, h, L_, W_)
    assert n_box > 800 and n_disc > 150, (n_cert, n_box, n_disc)


def test_rect_pair_filter_is_a_certificate_of_convex_intersects(oracle):
    """The four-projection filter in front of the pair SAT (t2d_geom_dev.h rect_pair_filter), restated here in numpy on the
    oracle's own fp64 vertices: whenever it answers, the oracle's 32-orientation test gives the same answer -- random pairs,
    pairs moved to exact contact, and pairs a few nanometres either side of contact (those it must leave undecided or get
    right)."""
    rng = np.random.default_rng(99)

    def filt(A, B):
        pa, qa, pb, qb = A[0] - A[3], A[1] - A[0], B[0] - B[3], B[1] - B[0]
        d = (B[0] + B[2]) - (A[0] + A[2])
        g = [abs(pa @ d) - (pa @ pa + abs(pa @ pb) + abs(pa @ qb)), abs(qa @ d) - (qa @ qa + abs(qa @ pb) + abs(qa @ qb)),
             abs(pb @ d) - (pb @ pb + abs(pa @ pb) + abs(qa @ pb)), abs(qb @ d) - (qb @ qb + abs(pa @ qb) + abs(qa @ qb))]
        m = max(g)
        return 0 if m > 1e-6 else (1 if m < -1e-6 else 2)

    counts = [0, 0, 0]
    for it in range(6000):
        La, Wa, Lb, Wb = rng.uniform(1.5, 18), rng.uniform(0.6, 2.6), rng.uniform(1.5, 18), rng.uniform(0.6, 2.6)
        xa, ya, ha = np.float32(rng.uniform(-250, 250)), np.float32(rng.uniform(-250, 250)), np.float32(rng.uniform(0, 6.3))
        hb = np.float32(ha + rng.choice([0.0, np.pi / 2, np.pi]) + rng.normal(0, 1e-3)) if it % 3 == 0 else np.float32(rng.uniform(0, 6.3))
        r = rng.uniform(0, 1.2) * 0.5 * (np.hypot(La, Wa) + np.hypot(Lb, Wb)); t = rng.uniform(0, 6.3)
        xb, yb = np.float32(xa + r * np.cos(t)), np.float32(ya + r * np.sin(t))
        if it % 4 == 0:    # slide B along the line of centres until the boxes touch to within fp32 resolution
            lo, hi = 0.0, 40.0
            for _ in range(60):
                mid = 0.5 * (lo + hi)
                Bm = oracle.pose_obb(float(np.float32(xa + mid * np.cos(t))), float(np.float32(ya + mid * np.sin(t))), float(hb), Lb, Wb, trig=1)
                if oracle.convex_intersects(oracle.pose_obb(float(xa), float(ya), float(ha), La, Wa, trig=1), Bm):
                    lo = mid
                else:
                    hi = mid
            xb, yb = np.float32(xa + lo * np.cos(t)), np.float32(ya + lo * np.sin(t))
        A = oracle.pose_obb(float(xa), float(ya), float(ha), La, Wa, trig=1)
        B = oracle.pose_obb(float(xb), float(yb), float(hb), Lb, Wb, trig=1)
        v = filt(A, B)
        counts[v] += 1
        if v != 2:
            assert bool(v) == oracle.convex_intersects(A, B), (xa, ya, ha, La, Wa, xb, yb, hb, Lb, Wb)
    assert counts[0] > 1000 and counts[1] > 1000 and counts[2] < 600, counts


def test_box_outside_a_polygon_edge_certifies_separation(oracle):
    """rect_vs_convex_filter (t2d_geom_dev.h), restated on the oracle's vertices: a polygon edge with the whole box strictly
    beyond it certifies `separate`, the box's centre strictly inside the polygon certifies `intersects` -- whenever the
    filter answers, the oracle's convex_intersects gives the same answer; it answers on most separate pairs, on deep
    overlaps, and never on touching ones."""
    rng = np.random.default_rng(5)

    def verdict(A, B):      # 0 separated, 1 intersecting, 2 undecided (rect_vs_convex_filter)
        p, q, c2 = A[0] - A[3], A[1] - A[0], A[0] + A[2]
        out, inside = False, True
        for j in range(len(B)):
            k = (j + 1) % len(B)
            n = np.array([B[k][1] - B[j][1], B[j][0] - B[k][0]])
            s2 = n @ (c2 - 2 * B[j]); m = 2e-9 * np.abs(n).sum()
            out |= s2 - (abs(n @ p) + abs(n @ q)) > m
            inside &= s2 < -m
        return 0 if out else (1 if inside else 2)

    fired = sep = hits = hit_cert = 0
    for it in range(5000):
        n = int(rng.integers(3, 5))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        if np.diff(np.append(ang, ang[0] + 2 * np.pi)).max() > np.pi - 0.2:
            continue
        rad = rng.uniform(2, 14)
        B = np.float64(np.float32(np.stack([rad * np.cos(ang), rad * np.sin(ang)], 1) + rng.uniform(-100, 100, 2)))
        if not oracle.polygon_is_convex(np.float32(B)):
            continue
        L_, W_ = rng.uniform(1.5, 14), rng.uniform(0.6, 2.6)
        d = rng.uniform(0, 1.4) * (rad + 0.5 * np.hypot(L_, W_)); t = rng.uniform(0, 6.3)
        cx, cy = np.float32(B.mean(0) + d * np.array([np.cos(t), np.sin(t)]))
        if it % 5 == 0:      # exact contact: box axis-aligned, its side on the polygon's bounding line
            h = 0.0; cx = np.float32(B[:, 0].max() + 0.5 * L_)
        else:
            h = float(np.float32(rng.uniform(0, 6.3)))
        A = oracle.pose_obb(float(cx), float(cy), h, L_, W_, trig=1)
        want = oracle.convex_intersects(A, B)
        got = verdict(A, B)
        fired += got == 0; sep += not want; hits += want; hit_cert += got == 1
        assert got == 2 or bool(got) == want, (A, B, got, want)
    assert sep > 700 and fired > 0.7 * sep and hits > 400 and hit_cert > 0.3 * hits, (fired, sep, hit_cert, hits)


def test_piece_clear_of_the_pose_support_is_a_miss(oracle):
    """The support test in front of piece_meets_quad_interior (t2d_collide.hip process_lane): when the pose lies on one side of
    a boundary piece's line, clear of it by the margin, the oracle's predicate says the piece does not meet the pose's interior."""
    import ctypes as C
    rng = np.random.default_rng(6)
    lib = oracle.lib()
    lib.t2do_piece_meets_quad_interior.restype = C.c_int
    clear_n = meets_n = 0
    for it in range(6000):
        L_, W_ = rng.uniform(1.5, 14), rng.uniform(0.6, 2.6)
        x, y, h = float(np.float32(rng.uniform(-200, 200))), float(np.float32(rng.uniform(-200, 200))), float(np.float32(rng.uniform(0, 6.3)))
        P = oracle.pose_obb(x, y, h, L_, W_, trig=1)
        mid = np.array([x, y]) + rng.uniform(-1.2, 1.2, 2) * np.hypot(L_, W_)
        th = rng.uniform(0, np.pi); half = rng.uniform(0.05, 30)
        if it % 4 == 0:   # a piece along the line of one of the pose's edges (touching from outside)
            e0, e1 = P[it % 3], P[it % 3 + 1]
            dvec = (e1 - e0) / np.linalg.norm(e1 - e0)
            a, b = e0 - half * dvec, e1 + half * dvec
        else:
            dvec = np.array([np.cos(th), np.sin(th)])
            a, b = mid - half * dvec, mid + half * dvec
        piece = np.ascontiguousarray(np.concatenate([a, b]), np.float64)
        meets = bool(lib.t2do_piece_meets_quad_interior(piece.ctypes.data_as(C.c_void_p), np.ascontiguousarray(P.reshape(8)).ctypes.data_as(C.c_void_p)))
        p, q, c2 = P[0] - P[3], P[1] - P[0], P[0] + P[2]
        n = np.array([a[1] - b[1], b[0] - a[0]])
        clear = abs(n @ (c2 - 2 * a)) > abs(n @ p) + abs(n @ q) + 2e-9 * np.abs(n).sum()
        clear_n += clear; meets_n += meets
        assert not (clear and meets), (piece, P)
    assert clear_n > 1500 and meets_n > 1000, (clear_n, meets_n)


@pytest.mark.parametrize("trig", [1, 0])
def test_static_collision_and_out_bound_equal_the_reference_detectors_executed(oracle, trig):
    """tests/golden/events_static_outbound.npz: the reference's StaticCollision and OutBound classes and Vehicle.get_pose, EXECUTED
    by oracle/gen_golden_events.py on 1500 scenes (an ego box of template or random size, 0-6 convex obstacles of 3-6 vertices
    around it, a boundary that sometimes cuts it, sometimes None) with exact stand-ins for the shapely calls they make.  The
    oracle's event step gives the same two verdicts for every scene: the pose's vertices, which predicate is asked of which
    object, `any` over the obstacles, the boundary tuple's order and the negation are the reference's.  (trig = 0: the
    deterministic sincos every kernel uses moves a vertex by <= 1 ulp -- a verdict could only differ on an exact touch.)"""
    import os
    from tactics2d_amd import layout as L
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "events_static_outbound.npz"))
    n = len(g["ego"])
    assert n == 1500
    # one env per scene, one participant; a parameter row per scene carries the box size
    rows = np.zeros((32, 24)); rows[:, L.P_SHAPE] = L.SHAPE_OBB; rows[:, L.P_MODEL] = L.MODEL_KINEMATICS
    mism = 0
    for c0 in range(0, n, 32):          # 32 type rows at a time (the table's capacity)
        c1 = min(n, c0 + 32)
        k = c1 - c0
        rows[:k, L.P_LENGTH] = g["size"][c0:c1, 0]; rows[:k, L.P_WIDTH] = g["size"][c0:c1, 1]
        eo = (g["obs_off"][c0:c1 + 1] - g["obs_off"][c0]).astype(np.int32)
        p0, p1 = g["obs_off"][c0], g["obs_off"][c1]
        vo = (g["obs_vert_off"][p0:p1 + 1] - g["obs_vert_off"][p0]).astype(np.int32)
        xy = g["obs_xy"][g["obs_vert_off"][p0]:g["obs_vert_off"][p1]]
        b = g["boundary"][c0:c1]
        valid = (~np.isnan(b[:, 0])).astype(np.uint8)
        flags, _ = oracle.collide(rows, k, 1, g["ego"][c0:c1, 0], g["ego"][c0:c1, 1], g["ego"][c0:c1, 2], np.arange(k, dtype=np.uint8),
                                  np.ones(k, np.uint8), (eo, vo, xy) if p1 > p0 else None, np.float32(np.nan_to_num(b)), valid, None, trig)
        got_static = (flags & L.FLAG_COLLISION_STATIC) != 0
        got_out = (flags & L.FLAG_OUT_BOUND) != 0
        mism += int((got_static != (g["static"][c0:c1] != 0)).sum()) + int((got_out != (g["out"][c0:c1] != 0)).sum())
    assert mism == 0, mism
    assert 300 < int(g["static"].sum()) < 800 and 200 < int(g["out"].sum()) < 800
