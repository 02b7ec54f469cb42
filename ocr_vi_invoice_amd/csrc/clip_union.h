// The closing pass of ClipperOffset::Execute (Clipper 6.4.2, called by pyclipper's PyclipperOffset.Execute, src/det/test.py:37-43):
//     clpr.AddPaths(m_destPolys, ptSubject, true);  clpr.Execute(ctUnion, solution, pftPositive, pftPositive);
// The raw offset path of a polygon crosses itself wherever the polygon is concave -- every concave join leaves an inverted loop, and
// the offsets of the two walls of a notch narrower than 2*delta overlap completely.  The union keeps the outline of the region whose
// winding number is positive.  Stated here without the sweep: exact planar arrangement of the path (rational coordinates, 128-bit
// predicates) -> winding number of every face -> the outer boundary loop -> Clipper's integer rounding of crossing points
// (IntersectPoint / TopX) -> FixupOutPolygon (duplicate and collinear vertices dropped) -> BuildResult's emission order.
// What is claimed is the CYCLIC vertex sequence and its orientation; the start vertex follows the rule Clipper's sweep gives for an
// outline with one top vertex (the top-most vertex -- right-most on a tie -- is emitted last) and is a modelling choice otherwise.
// MULTI-PATH RESULTS.  When the positive region encloses a pocket (a C-shaped polygon whose mouth is narrower than 2*delta), Clipper returns
// two paths, outer loop and hole.  The reference wraps them in np.array() inside a try (src/det/test.py:37-43, 84-90): ragged paths raise on
// numpy >= 1.24 (the box is then skipped) and give an object array on older numpy (element 0 = Clipper's first path).  Neither pyclipper nor
// the reference's numpy version can be pinned here, so this is a stated modelling choice: the box is KEPT and its polygon is the OUTER loop
// (tests/test_unclip_union_cpu.py::test_union_with_an_enclosed_pocket_emits_the_outer_loop_only pins product == oracle on such a shape).
// Host-only code; oracle/dbpost_cpu.py (clipper_union_outline) is the independent Python statement it must agree with exactly.
#pragma once
#include <math.h>

#include <algorithm>
#include <map>
#include <numeric>
#include <utility>
#include <vector>

namespace clipu {

typedef long long i64;
typedef __int128 i128;
struct P2 { int x, y; };

inline bool slopes_equal(P2 a, P2 b, P2 c) {  // ClipperLib::SlopesEqual(pt1, pt2, pt3)
    return (i64)(a.y - b.y) * (b.x - c.x) == (i64)(a.x - b.x) * (b.y - c.y);
}
inline i64 cround(double v) { return v < 0 ? (i64)(v - 0.5) : (i64)(v + 0.5); }  // ClipperLib::Round

// ClipperBase::AddPath on a closed path: duplicate vertices and vertices collinear with their neighbours go, until none is left
inline void addpath_cleanup(std::vector<P2>& q) {
    bool changed = true;
    while (changed && q.size() >= 3) {
        changed = false;
        size_t i = 0;
        while (i < q.size() && q.size() >= 3) {
            const P2 a = q[(i + q.size() - 1) % q.size()], b = q[i], c = q[(i + 1) % q.size()];
            if ((b.x == c.x && b.y == c.y) || slopes_equal(a, b, c)) {
                q.erase(q.begin() + i);
                changed = true;
                i = i > 0 ? i - 1 : 0;
            } else {
                ++i;
            }
        }
    }
    if (q.size() < 3) q.clear();
}

struct TEdge {  // the fields IntersectPoint / TopX read (InitEdge2 + SetDx)
    double botx, boty, topx, topy, dx;
    bool horizontal;
    TEdge(P2 a, P2 b) {
        if (a.y >= b.y) { botx = a.x; boty = a.y; topx = b.x; topy = b.y; }
        else { topx = a.x; topy = a.y; botx = b.x; boty = b.y; }
        const double dy = topy - boty;
        horizontal = dy == 0;
        dx = horizontal ? -1.0e40 : (topx - botx) / dy;
    }
    i64 top_x(double y) const { return y == topy ? (i64)topx : cround(botx + dx * (y - boty)); }
};

// ClipperLib::IntersectPoint for two crossing path edges (ProcessHorizontal's rule when one of them is horizontal)
inline P2 clipper_intersect_point(P2 p0, P2 p1, P2 q0, P2 q1) {
    TEdge e1(p0, p1), e2(q0, q1);
    if (e1.horizontal || e2.horizontal) {
        const TEdge &h = e1.horizontal ? e1 : e2, &o = e1.horizontal ? e2 : e1;
        return {(int)o.top_x(h.boty), (int)h.boty};
    }
    if (e2.dx < e1.dx) std::swap(e1, e2);  // Edge1 = the edge on the left below the crossing (AEL order in BuildIntersectList)
    if (e1.dx == 0) {
        const double x = e1.botx, b2 = e2.boty - (e2.botx / e2.dx);
        return {(int)x, (int)cround(x / e2.dx + b2)};
    }
    if (e2.dx == 0) {
        const double x = e2.botx, b1 = e1.boty - (e1.botx / e1.dx);
        return {(int)x, (int)cround(x / e1.dx + b1)};
    }
    const double b1 = e1.botx - e1.boty * e1.dx, b2 = e2.botx - e2.boty * e2.dx;
    const double q = (b2 - b1) / (e1.dx - e2.dx);
    const i64 y = cround(q);
    const i64 x = fabs(e1.dx) < fabs(e2.dx) ? cround(e1.dx * q + b1) : cround(e2.dx * q + b2);
    return {(int)x, (int)y};
}

// FixupOutPolygon on Clipper's ring (Next direction = reverse of the emission direction, OutRec.Pts at the top vertex), then
// BuildResult: emit from Pts->Prev along Prev
inline void fixup_and_emit(const std::vector<P2>& outline, std::vector<P2>& out) {
    out.clear();
    const int m = (int)outline.size();
    if (m < 3) return;
    int k = 0;
    for (int i = 1; i < m; ++i)
        if (outline[i].y < outline[k].y || (outline[i].y == outline[k].y && outline[i].x > outline[k].x)) k = i;
    std::vector<P2> pt(m);
    std::vector<int> nxt(m), prv(m);
    for (int i = 0; i < m; ++i) {
        pt[i] = outline[((k - i) % m + m) % m];
        nxt[i] = (i + 1) % m;
        prv[i] = (i + m - 1) % m;
    }
    auto same = [&](int a, int b) { return pt[a].x == pt[b].x && pt[a].y == pt[b].y; };
    int pp = 0, last_ok = -1;
    for (;;) {
        if (prv[pp] == pp || prv[pp] == nxt[pp]) return;
        if (same(pp, nxt[pp]) || same(pp, prv[pp]) || slopes_equal(pt[prv[pp]], pt[pp], pt[nxt[pp]])) {
            last_ok = -1;
            nxt[prv[pp]] = nxt[pp];
            prv[nxt[pp]] = prv[pp];
            pp = prv[pp];
        } else if (pp == last_ok) {
            break;
        } else {
            if (last_ok < 0) last_ok = pp;
            pp = nxt[pp];
        }
    }
    for (int p = prv[pp];; p = prv[p]) {
        out.push_back(pt[p]);
        if (p == pp) break;
    }
}

struct Rat { i64 n, d; };  // n / d, d > 0
inline bool rat_less(const Rat& a, const Rat& b) { return (i128)a.n * b.d < (i128)b.n * a.d; }
inline bool rat_eq(const Rat& a, const Rat& b) { return (i128)a.n * b.d == (i128)b.n * a.d; }

struct Vtx { i64 xn, yn, d; };  // (xn / d, yn / d), d > 0, gcd(xn, yn, d) = 1
struct VtxLess {
    bool operator()(const Vtx& a, const Vtx& b) const {  // lexicographic (x, y)
        const i128 l = (i128)a.xn * b.d, r = (i128)b.xn * a.d;
        if (l != r) return l < r;
        return (i128)a.yn * b.d < (i128)b.yn * a.d;
    }
};

struct Half { int u, v, mult, dx, dy, seg, twin, face, pos; bool boundary; };

// Outer polygon of Clipper's ctUnion / pftPositive of one closed integer path
inline void union_outline(const std::vector<P2>& path, std::vector<P2>& out) {
    out.clear();
    std::vector<P2> P = path;
    addpath_cleanup(P);
    const int n = (int)P.size();
    if (n < 3) return;
    auto A = [&](int i) { return P[i]; };
    auto B = [&](int i) { return P[(i + 1) % n]; };
    std::vector<std::vector<Rat>> cuts(n);
    for (int i = 0; i < n; ++i) { cuts[i].push_back({0, 1}); cuts[i].push_back({1, 1}); }
    for (int i = 0; i < n; ++i) {
        const P2 a = A(i), b = B(i);
        const i64 rx = b.x - a.x, ry = b.y - a.y;
        const int ax0 = std::min(a.x, b.x), ax1 = std::max(a.x, b.x), ay0 = std::min(a.y, b.y), ay1 = std::max(a.y, b.y);
        for (int j = i + 1; j < n; ++j) {
            const P2 c = A(j), e = B(j);
            if (ax1 < std::min(c.x, e.x) || std::max(c.x, e.x) < ax0 || ay1 < std::min(c.y, e.y) || std::max(c.y, e.y) < ay0) continue;
            const i64 sx = e.x - c.x, sy = e.y - c.y, wx = c.x - a.x, wy = c.y - a.y;
            i64 d = rx * sy - ry * sx;
            if (d != 0) {
                i64 t = wx * sy - wy * sx, u = wx * ry - wy * rx;
                if (d < 0) { d = -d; t = -t; u = -u; }
                if (t >= 0 && t <= d && u >= 0 && u <= d) {
                    cuts[i].push_back({t, d});
                    cuts[j].push_back({u, d});
                }
            } else if (wx * ry - wy * rx == 0) {  // collinear: the end points of each that fall inside the other
                const i64 rr = rx * rx + ry * ry, ss = sx * sx + sy * sy;
                for (const P2 q : {c, e}) {
                    const i64 t = (q.x - a.x) * rx + (q.y - a.y) * ry;
                    if (t >= 0 && t <= rr) cuts[i].push_back({t, rr});
                }
                for (const P2 q : {a, b}) {
                    const i64 u = (q.x - c.x) * sx + (q.y - c.y) * sy;
                    if (u >= 0 && u <= ss) cuts[j].push_back({u, ss});
                }
            }
        }
    }
    std::map<Vtx, int, VtxLess> vid;
    std::vector<Vtx> vpt;
    auto vertex = [&](i64 xn, i64 yn, i64 d) {
        i64 g = std::gcd(std::gcd(xn < 0 ? -xn : xn, yn < 0 ? -yn : yn), d);
        if (g > 1) { xn /= g; yn /= g; d /= g; }
        const Vtx key{xn, yn, d};
        auto it = vid.find(key);
        if (it != vid.end()) return it->second;
        const int id = (int)vpt.size();
        vid.emplace(key, id);
        vpt.push_back(key);
        return id;
    };
    std::vector<Half> he;  // directed atomic edges, twin pairs at 2k / 2k + 1, in creation order
    std::map<std::pair<int, int>, int> hid;
    for (int i = 0; i < n; ++i) {
        const P2 a = A(i), b = B(i);
        const i64 rx = b.x - a.x, ry = b.y - a.y;
        std::vector<Rat>& ts = cuts[i];
        std::sort(ts.begin(), ts.end(), rat_less);
        ts.erase(std::unique(ts.begin(), ts.end(), rat_eq), ts.end());
        int prev = -1;
        for (const Rat& t : ts) {
            const int v = vertex((i64)a.x * t.d + t.n * rx, (i64)a.y * t.d + t.n * ry, t.d);
            if (prev >= 0 && prev != v) {
                auto it = hid.find({prev, v});
                if (it == hid.end()) {
                    const int k = (int)he.size();
                    he.push_back({prev, v, 1, (int)rx, (int)ry, i, k + 1, -1, 0, false});
                    he.push_back({v, prev, 0, (int)-rx, (int)-ry, i, k, -1, 0, false});
                    hid[{prev, v}] = k;
                    hid[{v, prev}] = k + 1;
                } else {
                    he[it->second].mult += 1;
                }
            }
            prev = v;
        }
    }
    const int nv = (int)vpt.size(), nh = (int)he.size();
    std::vector<std::vector<int>> adj(nv);  // outgoing half-edges, counter-clockwise from the +x axis (exact)
    for (int h = 0; h < nh; ++h) adj[he[h].u].push_back(h);
    auto ang_less = [&](int h1, int h2) {
        const i64 x1 = he[h1].dx, y1 = he[h1].dy, x2 = he[h2].dx, y2 = he[h2].dy;
        const int s1 = (y1 > 0 || (y1 == 0 && x1 > 0)) ? 0 : 1, s2 = (y2 > 0 || (y2 == 0 && x2 > 0)) ? 0 : 1;
        if (s1 != s2) return s1 < s2;
        return x1 * y2 - y1 * x2 > 0;
    };
    for (int u = 0; u < nv; ++u) {
        std::sort(adj[u].begin(), adj[u].end(), ang_less);
        for (size_t k = 0; k < adj[u].size(); ++k) he[adj[u][k]].pos = (int)k;
    }
    // faces: the face on the left of u->v continues with the edge clockwise-next to v->u around v
    int nf = 0;
    std::vector<std::vector<int>> faces;
    for (int h0 = 0; h0 < nh; ++h0) {
        if (he[h0].face >= 0) continue;
        faces.emplace_back();
        for (int g = h0; he[g].face < 0;) {
            he[g].face = nf;
            faces.back().push_back(g);
            const int tw = he[g].twin, v = he[g].v, deg = (int)adj[v].size();
            g = adj[v][(he[tw].pos + deg - 1) % deg];
        }
        ++nf;
    }
    // the unbounded face: every edge at the left-most (then lowest) vertex leaves into the right half plane, and the face holding the
    // direction (-1, 0) there is on the left of the edge with the largest angle
    int u0 = 0;
    for (int i = 1; i < nv; ++i)
        if (VtxLess()(vpt[i], vpt[u0])) u0 = i;
    int h0 = adj[u0][0];
    for (size_t k = 1; k < adj[u0].size(); ++k) {
        const int h = adj[u0][k];
        if ((i64)he[h0].dx * he[h].dy - (i64)he[h0].dy * he[h].dx > 0) h0 = h;
    }
    std::vector<int> wind(nf, 0), known(nf, 0), stack;
    known[he[h0].face] = 1;
    stack.push_back(he[h0].face);
    while (!stack.empty()) {
        const int f = stack.back();
        stack.pop_back();
        for (int h : faces[f]) {
            const int g = he[he[h].twin].face;
            if (!known[g]) {
                known[g] = 1;
                wind[g] = wind[f] - (he[h].mult - he[he[h].twin].mult);  // the left of a forward edge is one turn up on its right
                stack.push_back(g);
            }
        }
    }
    for (int h = 0; h < nh; ++h) he[h].boundary = wind[he[h].face] >= 1 && wind[he[he[h].twin].face] <= 0;
    // boundary loops (interior on the left): at a vertex the next boundary edge is the first one counter-clockwise from the way back;
    // the outer loop is the one through the left-most boundary vertex
    std::vector<char> seen(nh, 0);
    std::vector<int> best, loop;
    int best_lo = -1;
    for (int s = 0; s < nh; ++s) {
        if (!he[s].boundary || seen[s]) continue;
        loop.clear();
        for (int g = s; !seen[g];) {
            seen[g] = 1;
            loop.push_back(g);
            const int v = he[g].v, deg = (int)adj[v].size(), k = he[he[g].twin].pos;
            for (int t = 1; t <= deg; ++t) {
                const int cand = adj[v][(k + t) % deg];
                if (he[cand].boundary) { g = cand; break; }
            }
        }
        int lo = he[loop[0]].u;
        for (int g : loop)
            if (VtxLess()(vpt[he[g].u], vpt[lo])) lo = he[g].u;
        if (best_lo < 0 || VtxLess()(vpt[lo], vpt[best_lo])) { best_lo = lo; best = loop; }
    }
    if (best.empty()) return;
    std::vector<P2> outline;
    for (size_t k = 0; k < best.size(); ++k) {  // vertex u of edge k: reached on the previous edge, left on this one
        const Half& cur = he[best[k]];
        const Half& prv = he[best[(k + best.size() - 1) % best.size()]];
        const Vtx& p = vpt[cur.u];
        if (p.d == 1) outline.push_back({(int)p.xn, (int)p.yn});
        else outline.push_back(clipper_intersect_point(A(prv.seg), B(prv.seg), A(cur.seg), B(cur.seg)));
    }
    fixup_and_emit(outline, out);
}

}  // namespace clipu
