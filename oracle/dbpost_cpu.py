"""ORACLE (test infrastructure only -- never imported by the product path).

Restatement of the reference's DB post-processing, ``DBPostProcessor.__call__`` / ``box_score_fast`` / ``unclip``
(src/det/test.py:20-106) and the rescale + ``crop_image`` step of the pipeline (src/pipeline/pipeline2.py:324-343,
src/det/test.py:123-130).  The reference delegates the geometry to third-party compiled libraries that are absent from the build
container and ship no fixtures, so each is restated here from its published algorithm -- **parity unpinned**:

* ``cv2.findContours(img, RETR_LIST, CHAIN_APPROX_SIMPLE)``: Suzuki & Abe (1985) border following on the zero-padded image
  (8-connected foreground, outer and hole borders alike), points emitted where the chain direction changes; contours returned
  last-found first.
* ``cv2.arcLength`` (float32 segment lengths summed in double), ``cv2.approxPolyDP`` (OpenCV's iterative Douglas-Peucker for closed
  curves: 3 farthest-point passes to pick the two anchors, slice stack, final clean-up pass), ``cv2.contourArea`` (shoelace).
* ``cv2.fillPoly`` + ``cv2.mean(roi, mask)``: polygon boundary lines (8-connected Bresenham) plus interior pixels.
* ``shapely.Polygon.area / .length`` (shoelace / perimeter in double) and ``pyclipper.PyclipperOffset`` with JT_ROUND,
  ET_CLOSEDPOLYGON, default arc tolerance 0.25 (ClipperOffset::DoOffset / OffsetPoint / DoRound of Clipper 6.4.2) followed by
  Execute's closing self-union (ctUnion, pftPositive), restated as the outline of the positive-winding region of the raw path with
  Clipper's rounding of crossing points -- see clipper_union_outline for what is and is not claimed about the vertex order.

The product implementation (csrc/dbpost.hip + csrc/clip_union.h, host C++) is an independent statement of the same algorithms and must agree exactly.
"""
from __future__ import annotations

import math
from typing import List, Tuple

import numpy as np

# direction codes as in OpenCV's contour tracer: 0 = E, then counter-clockwise on screen (y down): NE, N, NW, W, SW, S, SE
DX = (1, 1, 0, -1, -1, -1, 0, 1)
DY = (0, -1, -1, -1, 0, 1, 1, 1)


def find_contours(binary: np.ndarray) -> List[np.ndarray]:
    """binary: 2-D array, non-zero = foreground.  Returns int32 arrays [n,2] of (x, y), last-found contour first."""
    h, w = binary.shape
    f = np.zeros((h + 2, w + 2), dtype=np.int32)
    f[1:-1, 1:-1] = (binary != 0)
    nbd = 1
    found = []
    # candidate start pixels in raster order: left neighbour background (outer border) or right neighbour background (hole border)
    fg = f == 1
    cand = fg & (~fg[:, np.r_[0, 0:w + 1]] | ~fg[:, np.r_[1:w + 2, w + 1]])
    ys, xs = np.nonzero(cand)
    for y, x in zip(ys.tolist(), xs.tolist()):
        v = f[y, x]
        if v == 1 and f[y, x - 1] == 0:
            is_hole = False
        elif v >= 1 and f[y, x + 1] == 0:
            is_hole = True
        else:
            continue
        nbd += 1
        found.append(_follow(f, x, y, is_hole, nbd))
    found.reverse()
    return [np.asarray(c, dtype=np.int32).reshape(-1, 2) - 1 for c in found]   # undo the 1-pixel padding


def _follow(f, x0, y0, is_hole, nbd):
    pts = []
    s_end = s = 0 if is_hole else 4
    while True:                                    # clockwise search for the first non-zero neighbour
        s = (s - 1) & 7
        if f[y0 + DY[s], x0 + DX[s]] != 0 or s == s_end:
            break
    if s == s_end and f[y0 + DY[s], x0 + DX[s]] == 0:   # isolated pixel
        f[y0, x0] = -nbd
        return [(x0, y0)]
    x1, y1 = x0 + DX[s], y0 + DY[s]
    x3, y3 = x0, y0
    prev_s = s ^ 4
    while True:
        s_end = s
        while True:                                # counter-clockwise search
            s += 1
            x4, y4 = x3 + DX[s & 7], y3 + DY[s & 7]
            if f[y4, x4] != 0:
                break
        s &= 7
        if (s - 1) & 0xFFFFFFFF < s_end & 0xFFFFFFFF:   # the right-hand neighbour (direction 0) was examined and is background
            f[y3, x3] = -nbd
        elif f[y3, x3] == 1:
            f[y3, x3] = nbd
        if s != prev_s:                            # CHAIN_APPROX_SIMPLE: keep the point only where the direction changes
            pts.append((x3, y3))
        prev_s = s
        if x4 == x0 and y4 == y0 and x3 == x1 and y3 == y1:
            break
        x3, y3 = x4, y4
        s = (s + 4) & 7
    return pts


def arc_length_closed(pts: np.ndarray) -> float:
    p = pts.astype(np.float32)
    d = p - np.roll(p, 1, axis=0)
    seg = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32)
    return float(np.cumsum(seg.astype(np.float64))[-1]) if len(seg) else 0.0     # sequential double accumulation, as cv::arcLength


def approx_poly_dp_closed(src: np.ndarray, eps: float) -> np.ndarray:
    count = len(src)
    if count == 0:
        return src.copy()
    src = [(int(x), int(y)) for x, y in src]
    dst = []
    eps2 = eps * eps
    stack = []
    right_start, pos = 0, 0
    le_eps = False
    start_pt = (-1000000, -1000000)
    for _ in range(3):                              # 1. approximately the two farthest points
        max_dist = 0.0
        pos = (pos + right_start) % count
        start_pt = src[pos]; pos = (pos + 1) % count
        for j in range(1, count):
            pt = src[pos]; pos = (pos + 1) % count
            dx, dy = pt[0] - start_pt[0], pt[1] - start_pt[1]
            dist = float(dx * dx + dy * dy)
            if dist > max_dist:
                max_dist = dist
                right_start = j
        le_eps = max_dist <= eps2
    if not le_eps:                                  # 2. initialise the stack
        slice_start = pos % count
        right_end = slice_start
        slice_end = right_start = (right_start + slice_start) % count
        stack.append((right_start, right_end))
        stack.append((slice_start, slice_end))
    else:
        dst.append(start_pt)
    while stack:                                    # 3. the recursive process, iteratively
        s_start, s_end = stack.pop()
        end_pt = src[s_end]
        pos = s_start
        start_pt = src[pos]; pos = (pos + 1) % count
        if pos != s_end:
            dx, dy = end_pt[0] - start_pt[0], end_pt[1] - start_pt[1]
            max_dist = 0.0
            r_start = 0
            while pos != s_end:
                pt = src[pos]; pos = (pos + 1) % count
                dist = abs(float((pt[1] - start_pt[1]) * dx - (pt[0] - start_pt[0]) * dy))
                if dist > max_dist:
                    max_dist = dist
                    r_start = (pos + count - 1) % count
            le = max_dist * max_dist <= eps2 * float(dx * dx + dy * dy)
        else:
            le = True
            start_pt = src[s_start]
            r_start = 0
        if le:
            dst.append(start_pt)
        else:
            stack.append((r_start, s_end))
            stack.append((s_start, r_start))
    # 4. final clean-up: drop points on [almost] straight lines
    count = new_count = len(dst)
    if count == 0:
        return np.zeros((0, 2), np.int32)
    pos = count - 1
    start_pt = dst[pos]; pos = (pos + 1) % count
    wpos = pos
    pt = dst[pos]; pos = (pos + 1) % count
    i = 0
    while i < count and new_count > 2:
        end_pt = dst[pos]; pos = (pos + 1) % count
        dx, dy = end_pt[0] - start_pt[0], end_pt[1] - start_pt[1]
        dist = abs(float((pt[0] - start_pt[0]) * dy - (pt[1] - start_pt[1]) * dx))
        sip = (pt[0] - start_pt[0]) * (end_pt[0] - pt[0]) + (pt[1] - start_pt[1]) * (end_pt[1] - pt[1])
        if dist * dist <= 0.5 * eps2 * float(dx * dx + dy * dy) and dx != 0 and dy != 0 and sip >= 0:
            new_count -= 1
            dst[wpos] = start_pt = end_pt
            wpos = (wpos + 1) % count
            pt = dst[pos]; pos = (pos + 1) % count
            i += 2
            continue
        dst[wpos] = start_pt = pt
        wpos = (wpos + 1) % count
        pt = end_pt
        i += 1
    return np.asarray(dst[:new_count], dtype=np.int32).reshape(-1, 2)


def contour_area(pts: np.ndarray) -> float:
    """cv2.contourArea (not oriented): shoelace in double over float32 points."""
    p = pts.astype(np.float32).astype(np.float64)
    if len(p) == 0:
        return 0.0
    q = np.roll(p, 1, axis=0)
    return abs(float(np.sum(q[:, 0] * p[:, 1] - p[:, 0] * q[:, 1]))) * 0.5


def _line_pixels(x0, y0, x1, y1):
    """8-connected Bresenham walk (cv::LineIterator semantics: the major axis advances every step)."""
    dx, dy = abs(x1 - x0), abs(y1 - y0)
    sx = 1 if x1 >= x0 else -1
    sy = 1 if y1 >= y0 else -1
    out = []
    if dx >= dy:
        err = dx - 2 * dy
        x, y = x0, y0
        for _ in range(dx + 1):
            out.append((x, y))
            if err < 0:
                y += sy
                err += 2 * dx
            err -= 2 * dy
            x += sx
    else:
        err = dy - 2 * dx
        x, y = x0, y0
        for _ in range(dy + 1):
            out.append((x, y))
            if err < 0:
                x += sx
                err += 2 * dy
            err -= 2 * dx
            y += sy
    return out


def polygon_mask(pts: np.ndarray, h: int, w: int) -> np.ndarray:
    """cv2.fillPoly(mask, [pts], 1) for integer vertices: boundary lines plus the even-odd interior at pixel centres."""
    mask = np.zeros((h, w), np.uint8)
    n = len(pts)
    P = [(int(x), int(y)) for x, y in pts]
    for i in range(n):
        for x, y in _line_pixels(*P[i], *P[(i + 1) % n]):
            if 0 <= x < w and 0 <= y < h:
                mask[y, x] = 1
    for y in range(h):                              # even-odd scanline at the pixel-centre row y
        xs = []
        for i in range(n):
            (xa, ya), (xb, yb) = P[i], P[(i + 1) % n]
            if ya == yb:
                continue
            if (ya <= y < yb) or (yb <= y < ya):    # half-open rule: each vertex counted once
                xs.append(xa + (y - ya) * (xb - xa) / (yb - ya))
        xs.sort()
        for a, b in zip(xs[0::2], xs[1::2]):
            lo, hi = max(int(math.ceil(a)), 0), min(int(math.floor(b)), w - 1)
            if lo <= hi:
                mask[y, lo:hi + 1] = 1
    return mask


def box_score_fast(bitmap: np.ndarray, box: np.ndarray) -> float:
    """src/det/test.py:20-34."""
    h, w = bitmap.shape[:2]
    if len(box) == 0:
        return 0.0
    xmin = int(np.clip(np.floor(box[:, 0].min()), 0, w - 1)); xmax = int(np.clip(np.ceil(box[:, 0].max()), 0, w - 1))
    ymin = int(np.clip(np.floor(box[:, 1].min()), 0, h - 1)); ymax = int(np.clip(np.ceil(box[:, 1].max()), 0, h - 1))
    rel = box.astype(np.int64) - np.array([xmin, ymin])
    mask = polygon_mask(rel, ymax - ymin + 1, xmax - xmin + 1)
    roi = bitmap[ymin:ymax + 1, xmin:xmax + 1].astype(np.float64)
    cnt = int(mask.sum())
    vals = roi[mask != 0]                                  # row-major order
    return float(np.cumsum(vals)[-1] / cnt) if cnt else 0.0   # sequential double accumulation


def _round_half_away(v: float) -> int:
    return int(v - 0.5) if v < 0 else int(v + 0.5)


def clipper_offset_round(pts: np.ndarray, delta: float) -> np.ndarray:
    """ClipperOffset (6.4.2) AddPath(JT_ROUND, ET_CLOSEDPOLYGON) + the DoOffset half of Execute(delta), arc tolerance 0.25: the raw path."""
    P = [(int(x), int(y)) for x, y in pts]
    # AddPath: strip duplicate consecutive points (and a closing duplicate)
    Q = [P[0]]
    for p in P[1:]:
        if p != Q[-1]:
            Q.append(p)
    if len(Q) > 1 and Q[-1] == Q[0]:
        Q.pop()
    n = len(Q)
    if n < 3:
        return np.zeros((0, 2), np.int64)
    # Clipper requires outer paths to have positive orientation (ClipperOffset::FixOrientations): reverse if the area is negative
    a2 = sum(Q[i][0] * Q[(i + 1) % n][1] - Q[(i + 1) % n][0] * Q[i][1] for i in range(n))
    if a2 < 0:
        Q.reverse()
    arc_tol = 0.25
    y = arc_tol
    if y > abs(delta) * 0.25:
        y = abs(delta) * 0.25
    steps = math.pi / math.acos(1 - y / abs(delta))
    if steps > abs(delta) * math.pi:
        steps = abs(delta) * math.pi
    m_sin, m_cos = math.sin(2 * math.pi / steps), math.cos(2 * math.pi / steps)
    steps_per_rad = steps / (2 * math.pi)
    if delta < 0:
        m_sin = -m_sin
    normals = []
    for j in range(n):
        (x1, y1), (x2, y2) = Q[j], Q[(j + 1) % n]
        dx, dy = float(x2 - x1), float(y2 - y1)
        f = 1.0 / math.sqrt(dx * dx + dy * dy)
        normals.append((dy * f, -dx * f))
    out = []
    k = n - 1
    for j in range(n):
        sx, sy = Q[j]
        nk, nj = normals[k], normals[j]
        sin_a = nk[0] * nj[1] - nj[0] * nk[1]
        if abs(sin_a * delta) < 1.0:
            cos_a = nk[0] * nj[0] + nj[1] * nk[1]
            if cos_a > 0:                            # (almost) straight: one point; note Clipper returns WITHOUT advancing k here
                out.append((_round_half_away(sx + nk[0] * delta), _round_half_away(sy + nk[1] * delta)))
                continue
        elif sin_a > 1.0:
            sin_a = 1.0
        elif sin_a < -1.0:
            sin_a = -1.0
        if sin_a * delta < 0:                        # concave join: three points
            out.append((_round_half_away(sx + nk[0] * delta), _round_half_away(sy + nk[1] * delta)))
            out.append((sx, sy))
            out.append((_round_half_away(sx + nj[0] * delta), _round_half_away(sy + nj[1] * delta)))
        else:                                        # DoRound
            a = math.atan2(sin_a, nk[0] * nj[0] + nk[1] * nj[1])
            st = max(_round_half_away(steps_per_rad * abs(a)), 1)
            X, Y = nk
            for _ in range(st):
                out.append((_round_half_away(sx + X * delta), _round_half_away(sy + Y * delta)))
                X2 = X
                X = X * m_cos - m_sin * Y
                Y = X2 * m_sin + Y * m_cos
            out.append((_round_half_away(sx + nj[0] * delta), _round_half_away(sy + nj[1] * delta)))
        k = j
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


# ------------------------------------------------------------------------------------------------------------------------------
# ClipperOffset::Execute's closing pass: clpr.AddPaths(destPolys); clpr.Execute(ctUnion, solution, pftPositive, pftPositive).
# The raw offset path crosses itself wherever the source polygon is concave (each concave join leaves an inverted loop, and the
# offsets of the two walls of a notch narrower than 2*delta overlap completely); the union keeps the outline of the region whose
# winding number is positive.  Restated here as: exact planar arrangement of the path (rational arithmetic) -> winding number of
# every face -> outer boundary loop -> Clipper's integer rounding of the crossing points (IntersectPoint / TopX) -> FixupOutPolygon
# (duplicates and collinear vertices dropped) -> BuildResult's emission order.  What is claimed is the CYCLIC vertex sequence and
# its orientation; the start vertex follows the rule Clipper's sweep gives for an outline with a single top vertex (the vertex
# after the top-most one comes first, the top-most -- right-most on a tie -- last) and is a modelling choice otherwise.
# ------------------------------------------------------------------------------------------------------------------------------
from fractions import Fraction as _Fr


def _slopes_equal(p1, p2, p3) -> bool:
    """ClipperLib::SlopesEqual(pt1, pt2, pt3) (exact integer arithmetic)."""
    return (p1[1] - p2[1]) * (p2[0] - p3[0]) == (p1[0] - p2[0]) * (p2[1] - p3[1])


def _addpath_cleanup(path):
    """ClipperBase::AddPath, closed path: duplicate vertices and vertices collinear with their neighbours are removed until none is left."""
    q = [(int(p[0]), int(p[1])) for p in path]
    changed = True
    while changed and len(q) >= 3:
        changed = False
        i = 0
        while i < len(q) and len(q) >= 3:
            a, b, c = q[i - 1], q[i], q[(i + 1) % len(q)]
            if b == c or _slopes_equal(a, b, c):
                del q[i]
                changed = True
                i = max(i - 1, 0)
            else:
                i += 1
    return q if len(q) >= 3 else []


class _Edge:
    """TEdge fields IntersectPoint / TopX read (InitEdge2 + SetDx)."""
    HORIZONTAL = -1.0e40

    def __init__(self, a, b):
        (ax, ay), (bx, by) = a, b
        if ay >= by:
            self.bot, self.top = (ax, ay), (bx, by)
        else:
            self.top, self.bot = (ax, ay), (bx, by)
        dy = self.top[1] - self.bot[1]
        self.dx = self.HORIZONTAL if dy == 0 else (self.top[0] - self.bot[0]) / dy

    def top_x(self, y):
        if y == self.top[1]:
            return self.top[0]
        return _round_half_away(self.bot[0] + self.dx * (y - self.bot[1]))


def _clipper_intersect_point(sa, sb):
    """ClipperLib::IntersectPoint for two crossing path edges (ProcessHorizontal's rule when one of them is horizontal)."""
    e1, e2 = _Edge(*sa), _Edge(*sb)
    H = _Edge.HORIZONTAL
    if e1.dx == H or e2.dx == H:
        h, o = (e1, e2) if e1.dx == H else (e2, e1)
        return (o.top_x(h.bot[1]), h.bot[1])
    if e2.dx < e1.dx:                      # Edge1 = the edge on the left below the crossing (BuildIntersectList's AEL order)
        e1, e2 = e2, e1
    if e1.dx == 0:
        x = e1.bot[0]
        b2 = e2.bot[1] - (e2.bot[0] / e2.dx)
        return (x, _round_half_away(x / e2.dx + b2))
    if e2.dx == 0:
        x = e2.bot[0]
        b1 = e1.bot[1] - (e1.bot[0] / e1.dx)
        return (x, _round_half_away(x / e1.dx + b1))
    b1 = e1.bot[0] - e1.bot[1] * e1.dx
    b2 = e2.bot[0] - e2.bot[1] * e2.dx
    q = (b2 - b1) / (e1.dx - e2.dx)
    y = _round_half_away(q)
    x = _round_half_away(e1.dx * q + b1) if abs(e1.dx) < abs(e2.dx) else _round_half_away(e2.dx * q + b2)
    return (x, y)


def _fixup_and_emit(outline):
    """(Multi-path results -- a region that encloses a pocket: only the OUTER loop is returned, as the product does; the reference's
    np.array(Execute(d)) either raises on the ragged list (box skipped) or yields Clipper's first path, depending on its numpy: a stated
    modelling choice, see csrc/clip_union.h.)
    outline: the union's outer loop in emission direction (positive area).  Clipper holds it as a ring whose Next direction is the
    reverse, with OutRec.Pts at the top vertex; FixupOutPolygon walks that ring from Pts, and BuildResult emits from Pts->Prev along Prev."""
    m = len(outline)
    if m < 3:
        return []
    k = min(range(m), key=lambda i: (outline[i][1], -outline[i][0]))
    pt = [outline[(k - i) % m] for i in range(m)]           # Next order, Pts first
    nxt = [(i + 1) % m for i in range(m)]
    prv = [(i - 1) % m for i in range(m)]
    pp, last_ok = 0, None
    while True:
        if prv[pp] == pp or prv[pp] == nxt[pp]:
            return []
        if pt[pp] == pt[nxt[pp]] or pt[pp] == pt[prv[pp]] or _slopes_equal(pt[prv[pp]], pt[pp], pt[nxt[pp]]):
            last_ok = None
            nxt[prv[pp]], prv[nxt[pp]] = nxt[pp], prv[pp]
            pp = prv[pp]
        elif pp == last_ok:
            break
        else:
            if last_ok is None:
                last_ok = pp
            pp = nxt[pp]
    out, p = [], prv[pp]
    while True:
        out.append(pt[p])
        if p == pp:
            break
        p = prv[p]
    return out


def clipper_union_outline(path) -> List[Tuple[int, int]]:
    """Outer polygon of Clipper's ctUnion / pftPositive of one closed integer path (see the block comment above)."""
    P = _addpath_cleanup(path)
    n = len(P)
    if n < 3:
        return []
    seg = [(P[i], P[(i + 1) % n]) for i in range(n)]
    cuts = [[_Fr(0), _Fr(1)] for _ in range(n)]
    for i in range(n):
        (ax, ay), (bx, by) = seg[i]
        rx, ry = bx - ax, by - ay
        for j in range(i + 1, n):
            (cx, cy), (ex, ey) = seg[j]
            if max(ax, bx) < min(cx, ex) or max(cx, ex) < min(ax, bx) or max(ay, by) < min(cy, ey) or max(cy, ey) < min(ay, by):
                continue
            sx, sy = ex - cx, ey - cy
            wx, wy = cx - ax, cy - ay
            d = rx * sy - ry * sx
            if d != 0:
                t, u = wx * sy - wy * sx, wx * ry - wy * rx
                if d < 0:
                    d, t, u = -d, -t, -u
                if 0 <= t <= d and 0 <= u <= d:
                    cuts[i].append(_Fr(t, d))
                    cuts[j].append(_Fr(u, d))
            elif wx * ry - wy * rx == 0:                     # collinear: the end points of each that fall inside the other
                rr, ss = rx * rx + ry * ry, sx * sx + sy * sy
                for (qx, qy) in seg[j]:
                    t = (qx - ax) * rx + (qy - ay) * ry
                    if 0 <= t <= rr:
                        cuts[i].append(_Fr(t, rr))
                for (qx, qy) in seg[i]:
                    u = (qx - cx) * sx + (qy - cy) * sy
                    if 0 <= u <= ss:
                        cuts[j].append(_Fr(u, ss))
    vid, vpt = {}, []

    def vertex(x, y):
        key = (x, y)
        if key not in vid:
            vid[key] = len(vpt)
            vpt.append(key)
        return vid[key]

    mult, edir, eseg = {}, {}, {}                          # directed atomic edges: traversal count, integer direction, a source segment
    for i in range(n):
        (ax, ay), (bx, by) = seg[i]
        rx, ry = bx - ax, by - ay
        ts = sorted(set(cuts[i]))
        ids = [vertex(ax + t * rx, ay + t * ry) for t in ts]
        for u, v in zip(ids[:-1], ids[1:]):
            if u == v:
                continue
            mult[(u, v)] = mult.get((u, v), 0) + 1
            mult.setdefault((v, u), 0)
            edir[(u, v)], edir[(v, u)] = (rx, ry), (-rx, -ry)
            eseg.setdefault((u, v), i)
            eseg.setdefault((v, u), i)
    out_edges = {}
    for (u, v) in mult:
        out_edges.setdefault(u, []).append(v)

    def angle_key(u):
        import functools

        def cmp(v1, v2):                                     # counter-clockwise from the +x axis, exact
            (x1, y1), (x2, y2) = edir[(u, v1)], edir[(u, v2)]
            h1 = 0 if (y1 > 0 or (y1 == 0 and x1 > 0)) else 1
            h2 = 0 if (y2 > 0 or (y2 == 0 and x2 > 0)) else 1
            if h1 != h2:
                return h1 - h2
            c = x1 * y2 - y1 * x2
            return -1 if c > 0 else (1 if c < 0 else 0)
        return functools.cmp_to_key(cmp)

    pos = {}
    for u, vs in out_edges.items():
        vs.sort(key=angle_key(u))
        for k, v in enumerate(vs):
            pos[(u, v)] = k
    # faces: the face on the left of u->v continues with the edge clockwise-next to v->u around v
    face_of, faces = {}, []
    for h in mult:
        if h in face_of:
            continue
        cyc, g = [], h
        while g not in face_of:
            face_of[g] = len(faces)
            cyc.append(g)
            u, v = g
            vs = out_edges[v]
            g = (v, vs[(pos[(v, u)] - 1) % len(vs)])
        faces.append(cyc)
    # the unbounded face: every edge at the left-most (then lowest) vertex leaves into the right half plane, and the face holding the
    # direction (-1, 0) there is on the left of the edge with the largest angle
    u0 = min(range(len(vpt)), key=lambda i: vpt[i])
    v0 = out_edges[u0][0]
    for v in out_edges[u0][1:]:
        (x1, y1), (x2, y2) = edir[(u0, v0)], edir[(u0, v)]
        if x1 * y2 - y1 * x2 > 0:
            v0 = v
    outer = face_of[(u0, v0)]
    wind = {outer: 0}
    stack = [outer]
    while stack:
        f = stack.pop()
        for (u, v) in faces[f]:
            g = face_of[(v, u)]
            if g not in wind:
                wind[g] = wind[f] - (mult[(u, v)] - mult[(v, u)])   # the left of a forward edge is one turn up on its right
                stack.append(g)
    is_b = {h: (wind[face_of[h]] >= 1 and wind[face_of[(h[1], h[0])]] <= 0) for h in mult}
    # boundary loops (interior on the left); at a vertex the next boundary edge is the first one counter-clockwise from the way back
    best, seen = None, set()
    for h0 in mult:
        if not is_b[h0] or h0 in seen:
            continue
        loop, g = [], h0
        while g not in seen:
            seen.add(g)
            loop.append(g)
            u, v = g
            vs = out_edges[v]
            k = pos[(v, u)]
            for s in range(1, len(vs) + 1):
                cand = (v, vs[(k + s) % len(vs)])
                if is_b[cand]:
                    g = cand
                    break
        lo = min(vpt[u] for u, _ in loop)                    # the outer loop is the one through the left-most boundary vertex
        if best is None or lo < best[0]:
            best = (lo, loop)
    if best is None:
        return []
    loop = best[1]
    outline = []
    for k, (u, v) in enumerate(loop):                        # vertex u: reached on the previous edge, left on this one
        x, y = vpt[u]
        if x.denominator == 1 and y.denominator == 1:
            outline.append((int(x), int(y)))
        else:
            outline.append(_clipper_intersect_point(seg[eseg[loop[k - 1]]], seg[eseg[(u, v)]]))
    return _fixup_and_emit(outline)


def unclip(box: np.ndarray, unclip_ratio: float = 1.5) -> np.ndarray:
    """src/det/test.py:37-43 (shapely area / length in double; pyclipper JT_ROUND offset)."""
    p = box.astype(np.float64)
    q = np.roll(p, -1, axis=0)
    area = abs(float(np.sum(p[:, 0] * q[:, 1] - q[:, 0] * p[:, 1]))) * 0.5
    d = q - p
    length = float(np.cumsum(np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]))[-1])   # sequential
    if length == 0:
        return np.zeros((0, 2), np.int64)
    distance = area * unclip_ratio / length
    if distance <= 0:
        return np.zeros((0, 2), np.int64)
    out = clipper_union_outline(clipper_offset_round(box, distance))
    return np.asarray(out, dtype=np.int64).reshape(-1, 2)


def db_postprocess(pred: np.ndarray, thresh=0.3, box_thresh=0.6, max_candidates=1000, unclip_ratio=1.5, min_area=10.0
                   ) -> Tuple[List[np.ndarray], List[float]]:
    """DBPostProcessor.__call__ (src/det/test.py:55-106).  pred: (1,H,W) or (H,W) float probability map."""
    prob = pred[0] if pred.ndim == 3 else pred
    contours = find_contours(prob > thresh)
    boxes, scores = [], []
    for i, contour in enumerate(contours):
        if i >= max_candidates:
            break
        eps = 0.002 * arc_length_closed(contour)
        points = approx_poly_dp_closed(contour, eps)
        if points.shape[0] < 4:
            continue
        score = box_score_fast(prob, points)
        if box_thresh > score:
            continue
        if contour_area(points) < min_area:
            continue
        box = unclip(points, unclip_ratio)
        if len(box) < 4:
            continue
        boxes.append(box)
        scores.append(score)
    return boxes, scores


def components(prob: np.ndarray, thresh: float):
    """Oracle for the device half (ocrvi_db_components): threshold, 8-connected components (the connectivity cv2.findContours follows),
    per component (x0, y0, x1, y1, count, root, sum) with root = index y*W+x of its first pixel in raster order and
    sum = sum of round(prob * 2^20); rows sorted by root.  Also the 1-bit mask packed as the device packs it."""
    from scipy import ndimage as ndi
    H, W = prob.shape
    seg = prob > np.float32(thresh)
    lab, n = ndi.label(seg, structure=np.ones((3, 3), int))
    rows = []
    q = np.rint(prob.astype(np.float32) * np.float32(1048576.0)).astype(np.int64)
    for sl_i, sl in enumerate(ndi.find_objects(lab), start=1):
        m = lab[sl] == sl_i
        ys, xs = np.nonzero(m)
        y0, x0 = sl[0].start, sl[1].start
        root = int((ys[0] + y0) * W + xs[0] + x0)                 # np.nonzero is row-major: the first hit is the raster-first pixel
        rows.append((x0, y0, sl[1].stop - 1, sl[0].stop - 1, int(m.sum()), root, int(q[sl][m].sum())))
    rows.sort(key=lambda r: r[5])
    bits = np.packbits(seg.reshape(H, W // 32, 32), axis=-1, bitorder="little").view(np.uint32).reshape(H, W // 32)
    return rows, bits


def rescale_and_rect(box: np.ndarray, scale_w: float, scale_h: float, img_h: int, img_w: int):
    """pipeline2.py:324-328 (in-place true-divide into the integer array truncates toward zero; then astype(int32)) followed by the
    rectangle crop_image slices (src/det/test.py:123-130: cv2.boundingRect of the integer points, clamped)."""
    b = box.astype(np.int64).copy()
    b[:, 0] = np.trunc(b[:, 0] / scale_w).astype(np.int64)
    b[:, 1] = np.trunc(b[:, 1] / scale_h).astype(np.int64)
    x0, y0 = int(b[:, 0].min()), int(b[:, 1].min())
    bw, bh = int(b[:, 0].max()) - x0 + 1, int(b[:, 1].max()) - y0 + 1
    x, y = max(0, x0), max(0, y0)
    return b.astype(np.int32), (x, y, max(min(bw, img_w - x), 0), max(min(bh, img_h - y), 0))
