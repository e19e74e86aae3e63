"""numpy restatement of the reference's SingleLineLidar scan -- TEST INFRASTRUCTURE ONLY.

Follows tactics2d/sensor/lidar.py line by line (the module itself cannot be imported here: it pulls in
shapely at import time):
    _rotate_and_filter_obstacles   lidar.py:98-126   affine matrix [a, b, -b, a, x_off, y_off]
    _scan_obstacles                lidar.py:128-221  (rays x edges) determinant solve + filters
    SingleLineLidar.__init__       lidar.py:33-57    point_density = max(int(freq_detect / freq_scan), 1)
The obstacle-level range filter of _rotate_and_filter_obstacles (`distance(origin) <
max_perception_distance`) cannot change a result -- an obstacle entirely beyond the range only yields
intersections the per-ray range filter removes anyway -- and is not restated.
PARITY: PINNED.  oracle/gen_golden_lidar.py executes the reference's statements where they lie -- _rotate_and_filter_obstacles
(lidar.py:97-126: the matrix, the range filter) whole and _scan_obstacles from the beam table to the end (:160-221: the
determinant solve, its eight filters, the parallel-line rule, min / clip / inf) -- with stand-ins for exactly two shapely
calls, each its documented rule (affine_transform with a 6-element matrix; distance from a ring to a point), and
tests/test_lidar.py holds this restatement against the result bit for bit (tests/golden/lidar.npz, 167 scenes).
"""
import numpy as np


def scan(ego_xyh, rings, max_range, point_density):
    """ego_xyh: (x, y, heading) of the sensor; rings: list of (n, 2) fp64 vertex arrays (closed
    implicitly: edge k goes from vertex k to vertex k+1 mod n).  Returns float64[point_density]
    with inf where nothing is hit (lidar.py:218-221)."""
    x, y, theta = (float(v) for v in ego_xyh)
    a_ = np.cos(theta); b_ = np.sin(theta)                       # :110-114
    x_off = -x * a_ - y * b_
    y_off = x * b_ - y * a_
    x1s, x2s, y1s, y2s = [], [], [], []
    for ring in rings:
        ring = np.asarray(ring, np.float64)
        rx = a_ * ring[:, 0] + b_ * ring[:, 1] + x_off           # affine_transform, matrix [a, b, -b, a, xoff, yoff]
        ry = -b_ * ring[:, 0] + a_ * ring[:, 1] + y_off
        x1s.extend(rx); x2s.extend(np.roll(rx, -1)); y1s.extend(ry); y2s.extend(np.roll(ry, -1))   # :166-172
    if len(x1s) == 0:                                            # :173-175
        return np.full(point_density, np.inf)
    theta_b = np.linspace(0, 2 * np.pi, point_density, endpoint=False)      # :160
    a = np.sin(theta_b).reshape(-1, 1); b = -np.cos(theta_b).reshape(-1, 1); c = 0   # :161-163
    x1s, x2s, y1s, y2s = (np.array(v).reshape(1, -1) for v in (x1s, x2s, y1s, y2s))
    d = (y2s - y1s); e = (x1s - x2s); f = (y1s * x2s - x1s * y2s)           # :183-185
    det = a * e - b * d                                                     # :188
    parallel = det == 0
    det[parallel] = 1
    raw_x = (b * f - c * e) / det                                           # :191-192
    raw_y = (c * d - a * f) / det
    tmp_inf = max_range * 10; tmp_zero = 1e-8                               # :198-199
    lx = (np.cos(theta_b) * max_range).reshape(-1, 1); ly = (np.sin(theta_b) * max_range).reshape(-1, 1)
    raw_x[raw_x > np.maximum(tmp_zero, lx) + tmp_zero] = tmp_inf           # :205-208
    raw_x[raw_x < np.minimum(-tmp_zero, lx) - tmp_zero] = tmp_inf
    raw_y[raw_y > np.maximum(tmp_zero, ly) + tmp_zero] = tmp_inf
    raw_y[raw_y < np.minimum(-tmp_zero, ly) - tmp_zero] = tmp_inf
    raw_x[raw_x > np.maximum(x1s, x2s) + tmp_zero] = tmp_inf               # :210-213
    raw_x[raw_x < np.minimum(x1s, x2s) - tmp_zero] = tmp_inf
    raw_y[raw_y > np.maximum(y1s, y2s) + tmp_zero] = tmp_inf
    raw_y[raw_y < np.minimum(y1s, y2s) - tmp_zero] = tmp_inf
    raw_x[parallel] = tmp_inf                                               # :215
    obs = np.min(np.sqrt(raw_x ** 2 + raw_y ** 2), axis=1)                  # :218
    obs = np.clip(obs, 0, max_range)
    obs[obs == max_range] = np.inf
    return obs
