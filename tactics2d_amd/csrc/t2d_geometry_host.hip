// t2d_geometry_host.hip -- host side of the geometry the event kernels consume: the caller's CSR polygons
// (StaticCollision.reset traffic/event_detection/collision.py:45-46, OffLane.reset off_lane.py:19-20, Lane.geometry
// map/element/lane.py:125-130) -> CCW fans of 3- / 4-gons with boxes, the boundary pieces of each env's lane union, the safe
// rectangles, the packed per-workgroup LDS records -- or, beyond the 32 KiB record, the HBM grid tier (t2d_mapgrid.hip; the
// reference's STRtree, map/element/map.py:242-329) -- and the two host-only queries t2d_geometry_budget / t2d_lane_safe_rects.
// Split out of t2d_api.hip in round 6 (no behaviour change).
#include <math.h>
#include <string.h>

#include <algorithm>
#include <limits>
#include <memory>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "t2d_host.h"

namespace t2d {
namespace host {

double area2(const std::vector<double>& P) {
    const int n = (int)P.size() / 2;
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const int j = (i + 1) % n;
        a += P[2 * i] * P[2 * j + 1] - P[2 * j] * P[2 * i + 1];
    }
    return a;
}
double orient_h(const double* p, const double* q, const double* r) {
    double a = q[0] - p[0], b = r[1] - p[1];
    double c = q[1] - p[1], d = r[0] - p[0];
    return a * b - c * d;
}

// fp32 CSR polygons -> what the kernels consume: CCW-normalised, polygons of more than 4 vertices cut into a fan of
// quads (see HostGeo), per-part AABB; plus the undivided CCW rings for the lidar.  Validates convexity with the same
// fp64 orientation arithmetic the kernels use.
int prepare_polys(t2d_pool* p, const int32_t* env_off, const int32_t* vert_off, const float* xy,
                  t2d_pool::HostGeo& out) {
    const int E = p->v.n_env;
    if (env_off[0] != 0) return fail(p, T2D_ERR_INVALID, "env offsets must start at 0");
    for (int e = 0; e < E; ++e)
        if (env_off[e + 1] < env_off[e]) return fail(p, T2D_ERR_INVALID, "env offsets not monotone");
    const int P = env_off[E];
    if (P > 0 && vert_off[0] != 0) return fail(p, T2D_ERR_INVALID, "vertex offsets must start at 0");
    const int V = P > 0 ? vert_off[P] : 0;
    t2d_pool::HostGeo g;
    g.present = true;
    g.ring_env_off.assign(env_off, env_off + E + 1);
    g.ring_vert_off.assign(vert_off, vert_off + P + 1);
    if (P == 0) g.ring_vert_off.assign(1, 0);
    g.ring_xy.assign(2 * (size_t)V, 0.f);
    g.env_off.assign(1, 0);
    g.vert_off.assign(1, 0);
    auto emit = [&](const double* poly, const int* idx, int m) {   // one part; dropped when it has no area
        std::vector<double> part(2 * m);
        for (int k = 0; k < m; ++k) { part[2 * k] = poly[2 * idx[k]]; part[2 * k + 1] = poly[2 * idx[k] + 1]; }
        if (!(area2(part) > 0.0)) return;
        float xmin = (float)part[0], xmax = xmin, ymin = (float)part[1], ymax = ymin;
        for (int k = 0; k < m; ++k) {
            const float fx = (float)part[2 * k], fy = (float)part[2 * k + 1];  // exact: inputs are fp32
            g.xy.push_back(fx); g.xy.push_back(fy);
            xmin = std::min(xmin, fx); xmax = std::max(xmax, fx);
            ymin = std::min(ymin, fy); ymax = std::max(ymax, fy);
        }
        g.aabb.push_back(xmin); g.aabb.push_back(xmax); g.aabb.push_back(ymin); g.aabb.push_back(ymax);
        g.vert_off.push_back(g.vert_off.back() + m);
    };
    int e = 0;
    for (int q = 0; q < P; ++q) {
        while (e < E && q >= env_off[e + 1]) { g.env_off.push_back((int32_t)g.vert_off.size() - 1); ++e; }
        const int v0 = vert_off[q], n = vert_off[q + 1] - v0;
        if (n < 3 || n > T2D_MAX_POLY_VERTS)
            return fail(p, T2D_ERR_GEOMETRY, "polygon " + std::to_string(q) + " has " +
                                                 std::to_string(n) + " vertices (3..8 supported)");
        std::vector<double> poly(2 * n);
        for (int k = 0; k < 2 * n; ++k) poly[k] = (double)xy[2 * v0 + k];
        double a = area2(poly);
        if (a < 0.0) {  // clockwise -> reverse
            for (int i = 0, j = n - 1; i < j; ++i, --j) {
                std::swap(poly[2 * i], poly[2 * j]);
                std::swap(poly[2 * i + 1], poly[2 * j + 1]);
            }
            a = -a;
        }
        if (!(a > 0.0)) return fail(p, T2D_ERR_GEOMETRY, "polygon " + std::to_string(q) + " is degenerate");
        for (int i = 0; i < n; ++i)
            if (orient_h(&poly[2 * i], &poly[2 * ((i + 1) % n)], &poly[2 * ((i + 2) % n)]) < 0.0)
                return fail(p, T2D_ERR_GEOMETRY,
                            "polygon " + std::to_string(q) + " is not convex (decompose on the host)");
        for (int i = 0; i < n; ++i) {
            g.ring_xy[2 * (size_t)(v0 + i)] = (float)poly[2 * i];
            g.ring_xy[2 * (size_t)(v0 + i) + 1] = (float)poly[2 * i + 1];
        }
        if (n <= 4) {
            const int idx[4] = {0, 1, 2, 3};
            emit(poly.data(), idx, n);
        } else {
            for (int k = 1; k < n - 1; k += 2) {
                const int idx[4] = {0, k, k + 1, k + 2};
                emit(poly.data(), idx, k + 2 <= n - 1 ? 4 : 3);
            }
        }
    }
    while (e < E) { g.env_off.push_back((int32_t)g.vert_off.size() - 1); ++e; }
    out = std::move(g);
    return T2D_OK;
}

// Boundary of the union of every env's lane polygons (off-lane = not union(lanes).contains(pose); the
// reference's OffLane is a stub, the predicate mirrored is OutBound.update out_bound.py:37-48).  Same specification as
// t2do_lane_boundary in oracle/t2d_oracle.c (independent restatement, same IEEE operations in the same order): every
// edge q0 -> q1 of every CCW lane polygon minus the parts whose right-hand side is covered by another lane of the env
// -- closed parametric clip against the other polygon's half-planes with positive length, a collinear edge of the same
// direction rejects the polygon -- covered intervals merged in ascending order, joined when they meet within kLaneTau.
constexpr double kLaneTau = 1e-9;

bool edge_covered_by(const double* q0, const double* q1, const double* M, int n, double& a, double& b) {
    const double dx = q1[0] - q0[0], dy = q1[1] - q0[1];
    double t0 = 0.0, t1 = 1.0;
    for (int j = 0; j < n; ++j) {
        const double* f0 = M + 2 * j;
        const double* f1 = M + 2 * ((j + 1) % n);
        const double ex = f1[0] - f0[0], ey = f1[1] - f0[1];
        const double num = ex * (q0[1] - f0[1]) - ey * (q0[0] - f0[0]);
        const double den = ex * dy - ey * dx;
        if (den == 0.0) {
            if (num < 0.0) return false;
            if (num == 0.0 && ex * dx + ey * dy > 0.0) return false;
        } else {
            const double tc = -num / den;
            if (den > 0.0) t0 = tc > t0 ? tc : t0;
            else t1 = tc < t1 ? tc : t1;
        }
    }
    if (!(t0 < t1)) return false;
    a = t0; b = t1;
    return true;
}

void build_lane_boundary(int E, t2d_pool::HostGeo& g) {
    const int P = g.env_off[E];
    g.bnd_off.assign((size_t)P + 1, 0);
    g.bnd.clear();
    std::vector<std::pair<double, double>> iv;
    // Envs that hold the SAME lanes -- every env of a pool on one reference map -- share the result: the walk is edge x polygon
    // per env (seconds for the thousands of parts of a real map), done once per distinct geometry and copied.
    struct Done { std::vector<int32_t> count; std::vector<double> pieces; };
    std::unordered_map<std::string, Done> cache;
    for (int e = 0; e < E; ++e) {
        const int l0 = g.env_off[e], l1 = g.env_off[e + 1];
        std::string key;
        {
            const int v0 = g.vert_off[l0], v1 = g.vert_off[l1];
            key.assign(reinterpret_cast<const char*>(&g.xy[2 * (size_t)v0]), sizeof(float) * 2 * (size_t)(v1 - v0));
            for (int li = l0; li <= l1; ++li) {
                const int32_t rel = g.vert_off[li] - v0;
                key.append(reinterpret_cast<const char*>(&rel), sizeof rel);
            }
        }
        auto hit = cache.find(key);
        if (hit != cache.end()) {
            const Done& d = hit->second;
            g.bnd.insert(g.bnd.end(), d.pieces.begin(), d.pieces.end());
            int32_t at = g.bnd_off[(size_t)l0];
            for (int li = l0; li < l1; ++li) {
                at += d.count[(size_t)(li - l0)];
                g.bnd_off[(size_t)li + 1] = at;
            }
            continue;
        }
        Done d;
        const size_t bnd_before = g.bnd.size();
        for (int li = l0; li < l1; ++li) {
            const int v0 = g.vert_off[li], n = g.vert_off[li + 1] - v0;
            double L[2 * T2D_MAX_POLY_VERTS];
            for (int k = 0; k < 2 * n; ++k) L[k] = (double)g.xy[2 * (size_t)v0 + k];   // already CCW (prepare_polys)
            for (int j = 0; j < n; ++j) {
                const double* q0 = L + 2 * j;
                const double* q1 = L + 2 * ((j + 1) % n);
                const double dx = q1[0] - q0[0], dy = q1[1] - q0[1];
                if (dx == 0.0 && dy == 0.0) continue;
                iv.clear();
                for (int mi = l0; mi < l1; ++mi) {
                    if (mi == li) continue;
                    // (a polygon whose box does not reach this edge's cannot cover any of it: the clip below would say so,
                    // 4 comparisons say it first -- what keeps a 1000-part map at seconds, not minutes)
                    const float* bb = &g.aabb[4 * (size_t)mi];
                    const double ex0 = q0[0] < q1[0] ? q0[0] : q1[0], ex1 = q0[0] < q1[0] ? q1[0] : q0[0];
                    const double ey0 = q0[1] < q1[1] ? q0[1] : q1[1], ey1 = q0[1] < q1[1] ? q1[1] : q0[1];
                    if ((double)bb[0] > ex1 || (double)bb[1] < ex0 || (double)bb[2] > ey1 || (double)bb[3] < ey0) continue;
                    const int w0 = g.vert_off[mi], m = g.vert_off[mi + 1] - w0;
                    double M[2 * T2D_MAX_POLY_VERTS];
                    for (int k = 0; k < 2 * m; ++k) M[k] = (double)g.xy[2 * (size_t)w0 + k];
                    double a, b;
                    if (edge_covered_by(q0, q1, M, m, a, b)) iv.emplace_back(a, b);
                }
                std::stable_sort(iv.begin(), iv.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
                double r = 0.0;
                for (size_t k = 0; k <= iv.size(); ++k) {
                    const double a = k < iv.size() ? iv[k].first : 1.0;
                    const bool gap = k < iv.size() ? a > r + kLaneTau : r < 1.0 - kLaneTau;
                    if (gap) {
                        g.bnd.push_back(r == 0.0 ? q0[0] : q0[0] + r * dx);
                        g.bnd.push_back(r == 0.0 ? q0[1] : q0[1] + r * dy);
                        g.bnd.push_back(a == 1.0 ? q1[0] : q0[0] + a * dx);
                        g.bnd.push_back(a == 1.0 ? q1[1] : q0[1] + a * dy);
                    }
                    if (k < iv.size() && iv[k].second > r) r = iv[k].second;
                }
            }
            g.bnd_off[(size_t)li + 1] = (int32_t)(g.bnd.size() / 4);
            d.count.push_back(g.bnd_off[(size_t)li + 1] - g.bnd_off[(size_t)li]);
        }
        d.pieces.assign(g.bnd.begin() + (long)bnd_before, g.bnd.end());
        cache.emplace(std::move(key), std::move(d));
    }
}

// Rectangles inside the union of an env's lanes: the certificate of the step kernel's off-lane short cut (a pose whose
// outward-rounded box lies in one of them is contained in the union, hence not off-lane -- no polygon test needed).
// Every lane part that is an axis-aligned rectangle is one; two whose union is again a rectangle (equal extent on one axis
// -- exactly, these are the caller's fp32 coordinates -- and touching or overlapping intervals on the other) are merged,
// to a fixed point; rectangles inside another are dropped; the kSafeRects largest are kept, each shrunk by 0.1 mm (so that
// nothing that depends on how the rounding of the boundary walk treats a shared edge can sit inside a certified pose).
// Exactness of the certificate against the oracle's fp64 predicates: tests/test_oracle_geometry.py (safe rectangles).
constexpr float kSafeShrink = 1e-4f;
void build_safe_rects(int E, t2d_pool::HostGeo& g) {
    const float inf = std::numeric_limits<float>::infinity();
    g.safe.assign((size_t)E * t2d::kSafeRects * 4, 0.f);
    struct R { float x0, x1, y0, y1; };
    std::vector<R> rs;
    for (int e = 0; e < E; ++e) {
        rs.clear();
        for (int li = g.env_off[e]; li < g.env_off[e + 1]; ++li) {
            const int v0 = g.vert_off[li], n = g.vert_off[li + 1] - v0;
            if (n != 4) continue;
            const float* q = &g.xy[2 * (size_t)v0];
            float x0 = q[0], x1 = q[0], y0 = q[1], y1 = q[1];
            for (int k = 1; k < 4; ++k) {
                x0 = std::min(x0, q[2 * k]); x1 = std::max(x1, q[2 * k]);
                y0 = std::min(y0, q[2 * k + 1]); y1 = std::max(y1, q[2 * k + 1]);
            }
            // a convex CCW quad whose every vertex is a corner of its own bounding box, all four corners taken
            int seen = 0;
            bool ok = x0 < x1 && y0 < y1;
            for (int k = 0; k < 4 && ok; ++k) {
                const bool lx = q[2 * k] == x0, hx = q[2 * k] == x1, ly = q[2 * k + 1] == y0, hy = q[2 * k + 1] == y1;
                ok = (lx || hx) && (ly || hy);
                seen |= 1 << ((hx ? 1 : 0) | (hy ? 2 : 0));
            }
            if (ok && seen == 15) rs.push_back(R{x0, x1, y0, y1});
        }
        for (bool again = true; again;) {
            again = false;
            for (size_t i = 0; i < rs.size() && !again; ++i)
                for (size_t j = i + 1; j < rs.size() && !again; ++j) {
                    const R a = rs[i], b = rs[j];
                    const bool same_x = a.x0 == b.x0 && a.x1 == b.x1, same_y = a.y0 == b.y0 && a.y1 == b.y1;
                    const bool a_in_b = a.x0 >= b.x0 && a.x1 <= b.x1 && a.y0 >= b.y0 && a.y1 <= b.y1;
                    const bool b_in_a = b.x0 >= a.x0 && b.x1 <= a.x1 && b.y0 >= a.y0 && b.y1 <= a.y1;
                    if (a_in_b || b_in_a) {
                        rs[i] = a_in_b ? b : a;
                    } else if (same_x && !(a.y1 < b.y0 || b.y1 < a.y0)) {
                        rs[i] = R{a.x0, a.x1, std::min(a.y0, b.y0), std::max(a.y1, b.y1)};
                    } else if (same_y && !(a.x1 < b.x0 || b.x1 < a.x0)) {
                        rs[i] = R{std::min(a.x0, b.x0), std::max(a.x1, b.x1), a.y0, a.y1};
                    } else {
                        continue;
                    }
                    rs.erase(rs.begin() + (long)j);
                    again = true;
                }
        }
        std::stable_sort(rs.begin(), rs.end(), [](const R& a, const R& b) {
            return ((double)a.x1 - a.x0) * ((double)a.y1 - a.y0) > ((double)b.x1 - b.x0) * ((double)b.y1 - b.y0);
        });
        float* out = &g.safe[(size_t)e * t2d::kSafeRects * 4];
        for (int k = 0; k < t2d::kSafeRects; ++k) {
            R r{inf, -inf, inf, -inf};
            if (k < (int)rs.size()) {
                // inwards by the margin, rounded further inwards
                r.x0 = std::nextafter(rs[k].x0 + kSafeShrink, inf); r.x1 = std::nextafter(rs[k].x1 - kSafeShrink, -inf);
                r.y0 = std::nextafter(rs[k].y0 + kSafeShrink, inf); r.y1 = std::nextafter(rs[k].y1 - kSafeShrink, -inf);
                if (!(r.x0 < r.x1 && r.y0 < r.y1)) r = R{inf, -inf, inf, -inf};
            }
            out[4 * k] = r.x0; out[4 * k + 1] = r.x1; out[4 * k + 2] = r.y0; out[4 * k + 3] = r.y1;
        }
    }
}

int log2_pad(int A) {  // lanes per env = 2^l >= A, at least 2: the second lane of a one-agent env evaluates the Arrival IoU
    int l = 1;        // while the first evaluates the NoAction IoU (one SIMT pass instead of two calls in a row)
    while ((1 << l) < A) ++l;
    return l;
}

// dword offsets of one workgroup's record: env polygon ranges, polygon vertex ranges, AABBs, vertices (static, lanes)
void fill_layout(t2d::GeoLayout& gl, int epb, const int mp[2], const int mv[2], int mb) {
    int off = 0;
    for (int k = 0; k < 2; ++k) { gl.off_pstart[k] = off; off += epb + 1; }
    for (int k = 0; k < 2; ++k) { gl.off_vstart[k] = off; off += mp[k] + 1; }
    gl.off_bstart = off; off += mp[1] + 1;
    off = (off + 3) & ~3;  // 16-B align the float4 AABBs
    for (int k = 0; k < 2; ++k) { gl.off_aabb[k] = off; off += 4 * mp[k]; }
    for (int k = 0; k < 2; ++k) { gl.off_xy[k] = off; off += 2 * mv[k]; }  // even -> 8-B aligned
    off = (off + 3) & ~3;  // 16-B align the fp64 boundary pieces
    gl.off_bnd = off; off += 8 * mb;
    off = (off + 3) & ~3;

    gl.off_safe = off; off += mp[1] > 0 ? 4 * t2d::kSafeRects * epb : 0;
    gl.stride = (off + 3) & ~3;
    gl.epb = epb;
}

// Envs per workgroup of the step launch: as many as 256 lanes hold -- fewer while that leaves compute units without a
// workgroup (a pool of 512 envs x 32 participants is 64 workgroups of 8 envs, or 256 workgroups of 2: one per CU instead of
// three CUs in four idle), down to one wave per workgroup.
int envs_per_workgroup(t2d_pool* p, int log2A) {
    if (p->device_cus == 0) (void)hipDeviceGetAttribute(&p->device_cus, hipDeviceAttributeMultiprocessorCount, p->device);
    int epb = 256 >> log2A;
    // (halved only while the narrower workgroups still number at most the CUs: the looping forms of t2d_step_n with their
    // extra sets of waves want one workgroup per CU, not two)
    while (epb > 1 && (epb << log2A) > 64 && (p->v.n_env + (epb >> 1) - 1) / (epb >> 1) <= p->device_cus) epb >>= 1;
    return epb;
}

// The HBM grid tier (t2d_mapgrid.hip): the env's parts -- static and lane, the same fans of 3- / 4-gons and the same boundary
// pieces the LDS record would hold -- stay in global memory, indexed by one uniform grid per env.  Cell edge: the square root of
// the env's extent over its number of parts (about one part per cell), between 4 m and 64 m, at most 4096 cells per env.
int build_map_grid(t2d_pool* p) {
    const int E = p->v.n_env;
    std::vector<t2d::MapGridEnv> env((size_t)E);
    std::vector<int32_t> cell_start(1, 0);
    std::vector<t2d::MapItem> items;
    std::vector<std::vector<t2d::MapItem>> cells;
    const float m = t2d::kGridMargin;
    // Envs that hold the SAME static and lane polygons -- every env of a pool on one reference map -- share ONE grid and one set of
    // registrations (their boundary pieces are equal too: build_lane_boundary copies them): the walk's records then stay in the L2
    // instead of being each env's own 70 KB of a 70 MB array.
    std::unordered_map<std::string, int> first_env;
    for (int e = 0; e < E; ++e) {
        {
            std::string key;
            for (int k = 0; k < 2; ++k) {
                const auto& g = p->hgeo[k];
                const int32_t present = g.present ? 1 : 0;
                key.append(reinterpret_cast<const char*>(&present), sizeof present);
                if (!g.present) continue;
                const int l0 = g.env_off[e], l1 = g.env_off[e + 1];
                const int v0 = g.vert_off[l0], v1 = g.vert_off[l1];
                if (v1 > v0) key.append(reinterpret_cast<const char*>(&g.xy[2 * (size_t)v0]), sizeof(float) * 2 * (size_t)(v1 - v0));
                for (int li = l0; li <= l1; ++li) {
                    const int32_t rel = g.vert_off[li] - v0;
                    key.append(reinterpret_cast<const char*>(&rel), sizeof rel);
                }
            }
            const auto hit = first_env.find(key);
            if (hit != first_env.end()) {
                env[(size_t)e] = env[(size_t)hit->second];
                continue;
            }
            first_env.emplace(std::move(key), e);
        }
        float x0 = 0, x1 = 0, y0 = 0, y1 = 0;
        int n_parts = 0;
        for (int k = 0; k < 2; ++k) {
            const auto& g = p->hgeo[k];
            if (!g.present) continue;
            for (int q = g.env_off[e]; q < g.env_off[e + 1]; ++q, ++n_parts) {
                const float* b = &g.aabb[4 * (size_t)q];   // xmin, xmax, ymin, ymax
                if (n_parts == 0) { x0 = b[0]; x1 = b[1]; y0 = b[2]; y1 = b[3]; }
                x0 = std::min(x0, b[0]); x1 = std::max(x1, b[1]); y0 = std::min(y0, b[2]); y1 = std::max(y1, b[3]);
            }
        }
        t2d::MapGridEnv& h = env[(size_t)e];
        x0 -= 2 * m; y0 -= 2 * m; x1 += 2 * m; y1 += 2 * m;
        const double w = (double)x1 - x0, ht = (double)y1 - y0;
        double cell = n_parts > 0 ? sqrt(w * ht / n_parts) : 64.0;
        cell = std::min(64.0, std::max(4.0, cell));
        while ((floor(w / cell) + 1) * (floor(ht / cell) + 1) > 4096.0) cell *= 1.25;
        h.x0 = x0; h.y0 = y0; h.inv_cell = (float)(1.0 / cell);
        h.nx = (int)floor(w * (double)h.inv_cell) + 1;
        h.ny = (int)floor(ht * (double)h.inv_cell) + 1;
        h.cell_off = (int32_t)cell_start.size() - 1;
        h.has_lanes = p->hgeo[1].present && p->hgeo[1].env_off[e + 1] > p->hgeo[1].env_off[e];
        h.pad = 0;
        cells.assign((size_t)h.nx * h.ny, {});
        auto cell_of = [&](float v, float o, int n) {   // the kernel's expression: floor((v - o) * inv_cell) in fp64, clamped
            int c = (int)floor(((double)v - (double)o) * (double)h.inv_cell);
            return c < 0 ? 0 : (c >= n ? n - 1 : c);
        };
        for (int k = 0; k < 2; ++k) {
            const auto& g = p->hgeo[k];
            if (!g.present) continue;
            for (int q = g.env_off[e]; q < g.env_off[e + 1]; ++q) {
                const float* b = &g.aabb[4 * (size_t)q];
                const int ix0 = cell_of(b[0] - m, h.x0, h.nx), ix1 = cell_of(b[1] + m, h.x0, h.nx);
                const int iy0 = cell_of(b[2] - m, h.y0, h.ny), iy1 = cell_of(b[3] + m, h.y0, h.ny);
                t2d::MapItem it{};
                const int v0 = g.vert_off[q], n = g.vert_off[q + 1] - v0;
                for (int j = 0; j < 4; ++j) {   // (load_quad_f32's padding: a triangle repeats its first vertex)
                    const int jj = j < n ? j : 0;
                    it.xy[2 * j] = g.xy[2 * (size_t)(v0 + jj)];
                    it.xy[2 * j + 1] = g.xy[2 * (size_t)(v0 + jj) + 1];
                }
                it.bnd0 = k == 1 && !g.bnd_off.empty() ? g.bnd_off[q] : 0;
                it.bnd1 = k == 1 && !g.bnd_off.empty() ? g.bnd_off[q + 1] : 0;
                it.cell_lo = (uint32_t)ix0 | ((uint32_t)iy0 << 16);
                it.kind = (uint32_t)k;
                for (int iy = iy0; iy <= iy1; ++iy)
                    for (int ix = ix0; ix <= ix1; ++ix) cells[(size_t)iy * h.nx + ix].push_back(it);
            }
        }
        for (const auto& c : cells) {
            items.insert(items.end(), c.begin(), c.end());
            cell_start.push_back((int32_t)items.size());
        }
    }
    int rc;
    if ((rc = dev_replace(p, &p->d_grid_env, env.data(), env.size()))) return rc;
    if ((rc = dev_replace(p, &p->d_grid_cell_start, cell_start.data(), cell_start.size()))) return rc;
    // (a queue entry of the kernels names a registration or a boundary piece in 25 bits)
    if (items.size() >= (size_t(1) << 25) || p->hgeo[1].bnd.size() / 4 >= (size_t(1) << 25))
        return fail(p, T2D_ERR_GEOMETRY, "the map's grid holds more than 2^25 part registrations or boundary pieces");
    if (items.empty()) items.push_back(t2d::MapItem{});
    if ((rc = dev_replace(p, &p->d_grid_items, items.data(), items.size()))) return rc;
    t2d::MapGridView mg{};
    mg.env = p->d_grid_env; mg.cell_start = p->d_grid_cell_start; mg.items = p->d_grid_items;
    {
        const auto& g = p->hgeo[1];
        std::vector<double> bd = g.present && !g.bnd.empty() ? g.bnd : std::vector<double>(4, 0.0);
        if ((rc = dev_replace(p, &p->d_grid_bnd, bd.data(), bd.size()))) return rc;
        mg.bnd = p->d_grid_bnd;
    }
    if (!p->d_map_flags) {
        T2D_HIP(p, hipMalloc((void**)&p->d_map_flags, sizeof(uint32_t) * (size_t)p->v.N));
        T2D_HIP(p, hipMemset(p->d_map_flags, 0, sizeof(uint32_t) * (size_t)p->v.N));
    }
    if (!p->d_grid_seg) {
        const size_t bytes = t2d::map_segment_bytes(p->v.N);
        T2D_HIP(p, hipMalloc(&p->d_grid_seg, bytes));
        T2D_HIP(p, hipMemset(p->d_grid_seg, 0, bytes));
    }
    p->mapgrid = mg;
    p->grid_tier = true;
    p->v.map_flags = p->d_map_flags;
    return T2D_OK;
}

// (Re)build the packed per-workgroup geometry records from the host CSR copies and upload them.
int rebuild_geo(t2d_pool* p) {
    const int E = p->v.n_env;
    const int log2A = log2_pad(p->v.A);
    const int epb_max = envs_per_workgroup(p, log2A);
    constexpr int kBudgetDwords = 8192;  // 32 KiB of dynamic LDS for the record
    t2d::GeoLayout gl{};
    gl.epb = epb_max;
    p->grid_tier = false;
    p->v.map_flags = nullptr;
    if (!p->hgeo[0].present && !p->hgeo[1].present) {
        int rc = dev_replace<uint32_t>(p, &p->d_geo, nullptr, 0);
        p->v.geo = nullptr;
        p->v.geo_layout = gl;
        p->v.wgmap = nullptr;   // the launch shape may have changed
        return rc;
    }
    int epb = epb_max;
    int mp[2], mv[2], mb;
    for (;; epb >>= 1) {
        const int nb = (E + epb - 1) / epb;
        mb = 0;
        for (int k = 0; k < 2; ++k) {
            mp[k] = mv[k] = 0;
            const auto& g = p->hgeo[k];
            if (!g.present) continue;
            for (int b = 0; b < nb; ++b) {
                const int e0 = b * epb, e1 = std::min(E, e0 + epb);
                const int p0 = g.env_off[e0], p1 = g.env_off[e1];
                mp[k] = std::max(mp[k], p1 - p0);
                mv[k] = std::max(mv[k], g.vert_off[p1] - g.vert_off[p0]);
                if (k == 1) mb = std::max(mb, g.bnd_off[p1] - g.bnd_off[p0]);
            }
        }
        fill_layout(gl, epb, mp, mv, mb);
        if (gl.stride <= kBudgetDwords) break;
        if ((epb << log2A) <= 64 || epb == 1) {
            // Too large for the LDS record even at one wave per workgroup: the map goes to the HBM grid tier (t2d_mapgrid.hip).
            // The event kernel then carries no static / lane record at all; the step runs as integrate -> map events -> events +
            // status (t2d_step_form: "unfused"), same results, any map size.
            if (p->scene_mode) return fail(p, T2D_ERR_GEOMETRY, "static + lane geometry of one workgroup exceeds the 32 KiB LDS record");
            int rc = dev_replace<uint32_t>(p, &p->d_geo, nullptr, 0);
            if (rc != T2D_OK) return rc;
            t2d::GeoLayout none{};
            none.epb = epb_max;
            p->v.geo = nullptr;
            p->v.geo_layout = none;
            p->v.wgmap = nullptr;
            return build_map_grid(p);
        }
    }
    for (int k = 0; k < 2; ++k) gl.has[k] = p->hgeo[k].present && mp[k] > 0;
    const int nb = (E + epb - 1) / epb;
    std::vector<uint32_t> rec((size_t)nb * gl.stride, 0u);
    for (int b = 0; b < nb; ++b) {
        uint32_t* r = rec.data() + (size_t)b * gl.stride;
        const int e0 = b * epb;
        for (int k = 0; k < 2; ++k) {
            const auto& g = p->hgeo[k];
            int32_t* pstart = reinterpret_cast<int32_t*>(r) + gl.off_pstart[k];
            int32_t* vstart = reinterpret_cast<int32_t*>(r) + gl.off_vstart[k];
            if (!g.present) continue;  // zeros: every env has an empty range
            const int pb = g.env_off[std::min(E, e0)];
            for (int el = 0; el <= epb; ++el) pstart[el] = g.env_off[std::min(E, e0 + el)] - pb;
            const int np = pstart[epb];
            const int vb = g.vert_off[pb];
            for (int q = 0; q <= np; ++q) vstart[q] = g.vert_off[pb + q] - vb;
            float* bb = reinterpret_cast<float*>(r) + gl.off_aabb[k];
            float* xy = reinterpret_cast<float*>(r) + gl.off_xy[k];
            memcpy(bb, g.aabb.data() + 4 * (size_t)pb, sizeof(float) * 4 * np);
            memcpy(xy, g.xy.data() + 2 * (size_t)vb, sizeof(float) * 2 * vstart[np]);
            if (k == 1) {   // boundary pieces of the lane unions, grouped by lane polygon
                int32_t* bstart = reinterpret_cast<int32_t*>(r) + gl.off_bstart;
                const int bb0 = g.bnd_off[pb];
                for (int q = 0; q <= np; ++q) bstart[q] = g.bnd_off[pb + q] - bb0;
                memcpy(r + gl.off_bnd, g.bnd.data() + 4 * (size_t)bb0, sizeof(double) * 4 * bstart[np]);
                if (mp[1] > 0) {
                    float* safe = reinterpret_cast<float*>(r) + gl.off_safe;
                    const float inf = std::numeric_limits<float>::infinity();
                    for (int el = 0; el < epb; ++el)
                        for (int q = 0; q < t2d::kSafeRects; ++q) {
                            float* o = safe + 4 * (el * t2d::kSafeRects + q);
                            if (e0 + el < E) memcpy(o, &g.safe[4 * ((size_t)(e0 + el) * t2d::kSafeRects + q)], 16);
                            else { o[0] = inf; o[1] = -inf; o[2] = inf; o[3] = -inf; }
                        }
                }
            }
        }
    }
    int rc = dev_replace(p, &p->d_geo, rec.data(), rec.size());
    if (rc != T2D_OK) return rc;
    p->v.geo = p->d_geo;
    p->v.geo_layout = gl;
    p->v.wgmap = nullptr;   // the launch shape may have changed
    return T2D_OK;
}


}  // namespace host
}  // namespace t2d

using namespace t2d::host;

extern "C" {

int t2d_lane_safe_rects(int32_t n_env, const int32_t* env_lane_offsets, const int32_t* lane_vert_offsets,
                              const float* verts_xy, float* out) {
    if (n_env <= 0 || !env_lane_offsets || !lane_vert_offsets || !out) return T2D_ERR_INVALID;
    std::unique_ptr<t2d_pool> tmp(new (std::nothrow) t2d_pool());   // host bookkeeping only: no device call below
    if (!tmp) return T2D_ERR_NOMEM;
    tmp->v.n_env = n_env;
    t2d_pool::HostGeo g;
    const int rc = prepare_polys(tmp.get(), env_lane_offsets, lane_vert_offsets, verts_xy, g);
    if (rc != T2D_OK) return fail(nullptr, rc, tmp->err);
    build_safe_rects(n_env, g);
    memcpy(out, g.safe.data(), sizeof(float) * g.safe.size());
    return T2D_OK;
}

// Host-only: what t2d_set_static_geometry + t2d_set_lane_geometry would make of these polygons -- the dwords of the packed
// record of the fullest workgroup at the narrowest workgroup the step kernels accept (one wave: 64 / padded max_agents envs)
// against the 32 KiB budget; T2D_ERR_GEOMETRY (with the message t2d_set_*_geometry would give) for polygons it rejects.
int t2d_geometry_budget(int32_t n_env, int32_t max_agents, const int32_t* env_poly_offsets, const int32_t* poly_vert_offsets,
                              const float* poly_xy, const int32_t* env_lane_offsets, const int32_t* lane_vert_offsets,
                              const float* lane_xy, int32_t* dwords_needed, int32_t* dwords_budget, int32_t* envs_per_workgroup_out) {
    if (n_env <= 0 || max_agents <= 0 || max_agents > T2D_MAX_AGENTS || !dwords_needed) return T2D_ERR_INVALID;
    std::unique_ptr<t2d_pool> tmp(new (std::nothrow) t2d_pool());   // host bookkeeping only: no device call below
    if (!tmp) return T2D_ERR_NOMEM;
    tmp->v.n_env = n_env;
    tmp->v.A = max_agents;
    int rc;
    if (env_poly_offsets) {
        if ((rc = prepare_polys(tmp.get(), env_poly_offsets, poly_vert_offsets, poly_xy, tmp->hgeo[0])) != T2D_OK) return fail(nullptr, rc, tmp->err);
    }
    if (env_lane_offsets) {
        if ((rc = prepare_polys(tmp.get(), env_lane_offsets, lane_vert_offsets, lane_xy, tmp->hgeo[1])) != T2D_OK) return fail(nullptr, rc, tmp->err);
        build_lane_boundary(n_env, tmp->hgeo[1]);
    }
    const int log2A = log2_pad(max_agents);
    const int epb = std::max(1, 64 >> log2A);
    const int nb = (n_env + epb - 1) / epb;
    int mp[2] = {0, 0}, mv[2] = {0, 0}, mb = 0;
    for (int k = 0; k < 2; ++k) {
        const auto& g = tmp->hgeo[k];
        if (!g.present) continue;
        for (int b = 0; b < nb; ++b) {
            const int e0 = b * epb, e1 = std::min((int)n_env, e0 + epb);
            const int p0 = g.env_off[e0], p1 = g.env_off[e1];
            mp[k] = std::max(mp[k], p1 - p0);
            mv[k] = std::max(mv[k], g.vert_off[p1] - g.vert_off[p0]);
            if (k == 1) mb = std::max(mb, g.bnd_off[p1] - g.bnd_off[p0]);
        }
    }
    t2d::GeoLayout gl{};
    fill_layout(gl, epb, mp, mv, mb);
    *dwords_needed = gl.stride;
    if (dwords_budget) *dwords_budget = 8192;
    if (envs_per_workgroup_out) *envs_per_workgroup_out = epb;
    return T2D_OK;
}


}  // extern "C"
