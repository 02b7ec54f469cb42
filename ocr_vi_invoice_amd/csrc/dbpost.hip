// Host-side DB post-processing (replaces DBPostProcessor.__call__, src/det/test.py:46-106): threshold -> border following ->
// polygon approximation -> polygon-masked score -> area filter -> round-join offset ("unclip").  Pure CPU code behind the C ABI:
// the reference runs this stage on the host as well (cv2 / pyclipper / shapely).  Each step restates the published algorithm of
// the library call it replaces (named at each function); those libraries are absent from the build container, so parity with
// them is unpinned -- oracle/dbpost_cpu.py is an independent Python statement of the same algorithms and must agree exactly.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

#include "clip_union.h"
#include "common.h"

namespace {

struct Pt { int x, y; };
const int kDX[8] = {1, 1, 0, -1, -1, -1, 0, 1};   // chain codes: 0 = east, then counter-clockwise on screen (y grows downwards)
const int kDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// One border of the Suzuki-Abe scan (cv2.findContours, CHAIN_APPROX_SIMPLE): follow from (x0,y0) in the padded label image.
void follow_border(signed char* f, int stride, int x0, int y0, bool is_hole, int nbd, std::vector<Pt>& out) {
    auto at = [&](int x, int y) -> signed char& { return f[(size_t)y * stride + x]; };
    int s_end = is_hole ? 0 : 4, s = s_end;
    do {
        s = (s - 1) & 7;
    } while (at(x0 + kDX[s], y0 + kDY[s]) == 0 && s != s_end);
    if (s == s_end) {  // isolated pixel
        at(x0, y0) = -nbd;
        out.push_back({x0, y0});
        return;
    }
    const int x1 = x0 + kDX[s], y1 = y0 + kDY[s];
    int x3 = x0, y3 = y0, prev_s = s ^ 4;
    for (;;) {
        s_end = s;
        int x4, y4;
        for (;;) {
            ++s;
            x4 = x3 + kDX[s & 7];
            y4 = y3 + kDY[s & 7];
            if (at(x4, y4) != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) at(x3, y3) = -nbd;   // the east neighbour was examined and is background
        else if (at(x3, y3) == 1) at(x3, y3) = nbd;
        if (s != prev_s) out.push_back({x3, y3});
        prev_s = s;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4; y3 = y4;
        s = (s + 4) & 7;
    }
}

// cv2.findContours(RETR_LIST, CHAIN_APPROX_SIMPLE) on the thresholded map; contours in discovery order (the caller reverses).
// Labels: 0 background, 1 unvisited foreground, 2 visited border, -2 visited border whose east neighbour is background.  RETR_LIST
// needs no per-border numbering, so one byte per pixel is enough; the padded label image is a per-thread buffer that is reused
// across calls (a fresh multi-megabyte allocation per page costs more in page faults than the scan itself).
// `label_buf` overrides the per-thread buffer (the batch entry point keeps one per worker slot across calls: its threads are short-lived).
thread_local std::vector<signed char>* label_buf = nullptr;
// `bits` (optional): the thresholded map as one bit per pixel, bit (x & 31) of word y * (W / 32) + (x >> 5) -- what the device half
// (dbcomp.hip) hands over; the segmentation is then taken from it instead of `prob > thresh`.
void find_contours(const float* prob, float thresh, int H, int W, std::vector<std::vector<Pt>>& contours, const uint32_t* bits = nullptr) {
    const int stride = W + 2;
    static thread_local std::vector<signed char> own;
    std::vector<signed char>& fbuf = label_buf ? *label_buf : own;
    fbuf.resize((size_t)(H + 2) * stride);
    signed char* f = fbuf.data();
    memset(f, 0, stride);
    memset(f + (size_t)(H + 1) * stride, 0, stride);
    for (int y = 0; y < H; ++y) {
        signed char* row = f + (size_t)(y + 1) * stride;
        row[0] = 0;
        if (bits) {
            const uint32_t* br = bits + (size_t)y * (W >> 5);
            for (int xw = 0; xw < (W >> 5); ++xw) {
                const uint32_t wv = br[xw];
                if (!wv) { memset(row + 1 + 32 * xw, 0, 32); continue; }
                for (int k = 0; k < 32; ++k) row[1 + 32 * xw + k] = (wv >> k) & 1;
            }
        } else {
            const float* pr = prob + (size_t)y * W;
            for (int x = 0; x < W; ++x) row[x + 1] = pr[x] > thresh;
        }
        row[W + 1] = 0;
    }
    const int nbd = 2;
    for (int y = 1; y <= H; ++y) {
        signed char* row = f + (size_t)y * stride;
        for (int x = 1; x <= W; ++x) {
            const int v = row[x];
            if (v == 0) {   // skip background eight pixels at a time
                while (x + 8 <= W) {
                    unsigned long long q;
                    memcpy(&q, row + x + 1, 8);
                    if (q) break;
                    x += 8;
                }
                continue;
            }
            bool is_hole;
            if (v == 1 && row[x - 1] == 0) is_hole = false;
            else if (v >= 1 && row[x + 1] == 0) is_hole = true;
            else continue;
            contours.emplace_back();
            follow_border(f, stride, x, y, is_hole, nbd, contours.back());
            for (Pt& p : contours.back()) { p.x -= 1; p.y -= 1; }
        }
    }
}

double arc_length_closed(const std::vector<Pt>& c) {  // cv2.arcLength(closed=True): float32 segment lengths, double sum
    double sum = 0;
    const size_t n = c.size();
    for (size_t i = 0; i < n; ++i) {
        const Pt &a = c[(i + n - 1) % n], &b = c[i];
        const float dx = (float)(b.x - a.x), dy = (float)(b.y - a.y);
        sum += (double)sqrtf(dx * dx + dy * dy);
    }
    return sum;
}

// cv2.approxPolyDP(curve, eps, closed=True): OpenCV's iterative Douglas-Peucker.
void approx_poly_closed(const std::vector<Pt>& src, double eps, std::vector<Pt>& dst) {
    dst.clear();
    const int count = (int)src.size();
    if (count == 0) return;
    const double eps2 = eps * eps;
    std::vector<std::pair<int, int>> stack;
    int right_start = 0, pos = 0;
    bool le_eps = false;
    Pt start_pt{-1000000, -1000000};
    for (int it = 0; it < 3; ++it) {
        double max_dist = 0;
        pos = (pos + right_start) % count;
        start_pt = src[pos]; pos = (pos + 1) % count;
        for (int j = 1; j < count; ++j) {
            const Pt pt = src[pos]; pos = (pos + 1) % count;
            const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
            const double dist = dx * dx + dy * dy;
            if (dist > max_dist) { max_dist = dist; right_start = j; }
        }
        le_eps = max_dist <= eps2;
    }
    if (!le_eps) {
        const int slice_start = pos % count;
        const int right_end = slice_start;
        const int slice_end = (right_start + slice_start) % count;
        stack.push_back({slice_end, right_end});
        stack.push_back({slice_start, slice_end});
    } else {
        dst.push_back(start_pt);
    }
    while (!stack.empty()) {
        const std::pair<int, int> sl = stack.back();
        stack.pop_back();
        const Pt end_pt = src[sl.second];
        pos = sl.first;
        start_pt = src[pos]; pos = (pos + 1) % count;
        bool le;
        int r_start = 0;
        if (pos != sl.second) {
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            double max_dist = 0;
            while (pos != sl.second) {
                const Pt pt = src[pos]; pos = (pos + 1) % count;
                const double dist = fabs((double)(pt.y - start_pt.y) * dx - (double)(pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; r_start = (pos + count - 1) % count; }
            }
            le = max_dist * max_dist <= eps2 * (dx * dx + dy * dy);
        } else {
            le = true;
            start_pt = src[sl.first];
        }
        if (le) dst.push_back(start_pt);
        else {
            stack.push_back({r_start, sl.second});
            stack.push_back({sl.first, r_start});
        }
    }
    // final clean-up of [almost] collinear points
    const int cnt = (int)dst.size();
    int new_count = cnt;
    if (cnt == 0) return;
    pos = cnt - 1;
    start_pt = dst[pos]; pos = (pos + 1) % cnt;
    int wpos = pos;
    Pt pt = dst[pos]; pos = (pos + 1) % cnt;
    for (int i = 0; i < cnt && new_count > 2; ++i) {
        const Pt end_pt = dst[pos]; pos = (pos + 1) % cnt;
        const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        const double dist = fabs((double)(pt.x - start_pt.x) * dy - (double)(pt.y - start_pt.y) * dx);
        const long long sip = (long long)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (long long)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps2 * (dx * dx + dy * dy) && dx != 0 && dy != 0 && sip >= 0) {
            --new_count;
            dst[wpos] = start_pt = end_pt;
            wpos = (wpos + 1) % cnt;
            pt = dst[pos]; pos = (pos + 1) % cnt;
            ++i;
            continue;
        }
        dst[wpos] = start_pt = pt;
        wpos = (wpos + 1) % cnt;
        pt = end_pt;
    }
    dst.resize(new_count);
}

double shoelace_abs_half(const std::vector<Pt>& p) {  // cv2.contourArea / shapely Polygon.area on integer vertices
    const size_t n = p.size();
    long long a2 = 0;
    for (size_t i = 0; i < n; ++i) {
        const Pt &a = p[i], &b = p[(i + 1) % n];
        a2 += (long long)a.x * b.y - (long long)b.x * a.y;
    }
    return fabs((double)a2) * 0.5;
}

// 8-connected Bresenham walk from a to b (cv::LineIterator), marking mask pixels inside [0,w) x [0,h).
void draw_line(std::vector<unsigned char>& mask, int w, int h, Pt a, Pt b) {
    const int dx = abs(b.x - a.x), dy = abs(b.y - a.y);
    const int sx = b.x >= a.x ? 1 : -1, sy = b.y >= a.y ? 1 : -1;
    int x = a.x, y = a.y;
    auto put = [&]() { if (x >= 0 && x < w && y >= 0 && y < h) mask[(size_t)y * w + x] = 1; };
    if (dx >= dy) {
        int err = dx - 2 * dy;
        for (int i = 0; i <= dx; ++i) {
            put();
            if (err < 0) { y += sy; err += 2 * dx; }
            err -= 2 * dy;
            x += sx;
        }
    } else {
        int err = dy - 2 * dx;
        for (int i = 0; i <= dy; ++i) {
            put();
            if (err < 0) { x += sx; err += 2 * dy; }
            err -= 2 * dx;
            y += sy;
        }
    }
}

// box_score_fast (src/det/test.py:20-34): mean of the map over cv2.fillPoly's mask of the polygon inside its bounding box.
double box_score(const float* prob, int H, int W, const std::vector<Pt>& box) {
    if (box.empty()) return 0;
    int xmin = box[0].x, xmax = box[0].x, ymin = box[0].y, ymax = box[0].y;
    for (const Pt& p : box) { xmin = std::min(xmin, p.x); xmax = std::max(xmax, p.x); ymin = std::min(ymin, p.y); ymax = std::max(ymax, p.y); }
    xmin = std::min(std::max(xmin, 0), W - 1); xmax = std::min(std::max(xmax, 0), W - 1);
    ymin = std::min(std::max(ymin, 0), H - 1); ymax = std::min(std::max(ymax, 0), H - 1);
    const int w = xmax - xmin + 1, h = ymax - ymin + 1;
    std::vector<Pt> rel(box.size());
    for (size_t i = 0; i < box.size(); ++i) rel[i] = {box[i].x - xmin, box[i].y - ymin};
    std::vector<unsigned char> mask((size_t)w * h, 0);
    const size_t n = rel.size();
    for (size_t i = 0; i < n; ++i) draw_line(mask, w, h, rel[i], rel[(i + 1) % n]);
    std::vector<double> xs;
    for (int y = 0; y < h; ++y) {  // even-odd interior at pixel centres, half-open vertex rule
        xs.clear();
        for (size_t i = 0; i < n; ++i) {
            const Pt a = rel[i], b = rel[(i + 1) % n];
            if (a.y == b.y) continue;
            if ((a.y <= y && y < b.y) || (b.y <= y && y < a.y))
                xs.push_back((double)a.x + (double)((long long)(y - a.y) * (b.x - a.x)) / (double)(b.y - a.y));
        }
        std::sort(xs.begin(), xs.end());
        for (size_t k = 0; k + 1 < xs.size(); k += 2) {
            const int lo = std::max((int)ceil(xs[k]), 0), hi = std::min((int)floor(xs[k + 1]), w - 1);
            for (int x = lo; x <= hi; ++x) mask[(size_t)y * w + x] = 1;
        }
    }
    double sum = 0;
    long long cnt = 0;
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
            if (mask[(size_t)y * w + x]) { sum += (double)prob[(size_t)(y + ymin) * W + x + xmin]; ++cnt; }
    return cnt ? sum / (double)cnt : 0.0;
}

inline long long cround(double v) { return v < 0 ? (long long)(v - 0.5) : (long long)(v + 0.5); }  // ClipperLib::Round

// pyclipper.PyclipperOffset().AddPath(box, JT_ROUND, ET_CLOSEDPOLYGON); Execute(delta), first half: the raw offset path (Clipper 6.4.2
// DoOffset / OffsetPoint / DoRound, arc tolerance 0.25).  Execute's closing self-union is unclip_polygon() below.
void clipper_offset_round(const std::vector<Pt>& in, double delta, std::vector<Pt>& out) {
    out.clear();
    if (in.empty()) return;
    std::vector<Pt> q;
    q.push_back(in[0]);
    for (size_t i = 1; i < in.size(); ++i)
        if (in[i].x != q.back().x || in[i].y != q.back().y) q.push_back(in[i]);
    if (q.size() > 1 && q.back().x == q[0].x && q.back().y == q[0].y) q.pop_back();
    const int n = (int)q.size();
    if (n < 3) return;
    long long a2 = 0;
    for (int i = 0; i < n; ++i) a2 += (long long)q[i].x * q[(i + 1) % n].y - (long long)q[(i + 1) % n].x * q[i].y;
    if (a2 < 0) std::reverse(q.begin(), q.end());   // FixOrientations: outer paths must have positive area
    const double pi = 3.14159265358979323846;
    double y = 0.25;
    if (y > fabs(delta) * 0.25) y = fabs(delta) * 0.25;
    double steps = pi / acos(1 - y / fabs(delta));
    if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
    double m_sin = sin(2 * pi / steps);
    const double m_cos = cos(2 * pi / steps), steps_per_rad = steps / (2 * pi);
    if (delta < 0) m_sin = -m_sin;
    std::vector<double> nx(n), ny(n);
    for (int j = 0; j < n; ++j) {
        const double dx = (double)(q[(j + 1) % n].x - q[j].x), dy = (double)(q[(j + 1) % n].y - q[j].y);
        const double f = 1.0 / sqrt(dx * dx + dy * dy);
        nx[j] = dy * f;
        ny[j] = -dx * f;
    }
    auto push = [&](double px, double py) { out.push_back({(int)cround(px), (int)cround(py)}); };
    int k = n - 1;
    for (int j = 0; j < n; ++j) {
        const double sx = q[j].x, sy = q[j].y;
        double sin_a = nx[k] * ny[j] - nx[j] * ny[k];
        if (fabs(sin_a * delta) < 1.0) {
            const double cos_a = nx[k] * nx[j] + ny[j] * ny[k];
            if (cos_a > 0) {  // (almost) straight: one point; Clipper returns here WITHOUT advancing k
                push(sx + nx[k] * delta, sy + ny[k] * delta);
                continue;
            }
        } else if (sin_a > 1.0) sin_a = 1.0;
        else if (sin_a < -1.0) sin_a = -1.0;
        if (sin_a * delta < 0) {
            push(sx + nx[k] * delta, sy + ny[k] * delta);
            out.push_back(q[j]);
            push(sx + nx[j] * delta, sy + ny[j] * delta);
        } else {
            const double a = atan2(sin_a, nx[k] * nx[j] + ny[k] * ny[j]);
            const int st = std::max((int)cround(steps_per_rad * fabs(a)), 1);
            double X = nx[k], Y = ny[k];
            for (int i = 0; i < st; ++i) {
                push(sx + X * delta, sy + Y * delta);
                const double X2 = X;
                X = X * m_cos - m_sin * Y;
                Y = X2 * m_sin + Y * m_cos;
            }
            push(sx + nx[j] * delta, sy + ny[j] * delta);
        }
        k = j;
    }
}

// unclip (src/det/test.py:37-43) for a given distance: raw offset path -> Clipper's closing union (clip_union.h) -> its outer polygon
void unclip_polygon(const std::vector<Pt>& in, double delta, std::vector<Pt>& out) {
    std::vector<Pt> raw;
    clipper_offset_round(in, delta, raw);
    std::vector<clipu::P2> path(raw.size()), outline;
    for (size_t i = 0; i < raw.size(); ++i) path[i] = {raw[i].x, raw[i].y};
    clipu::union_outline(path, outline);
    out.resize(outline.size());
    for (size_t i = 0; i < outline.size(); ++i) out[i] = {outline[i].x, outline[i].y};
}

}  // namespace

extern "C" int ocrvi_unclip_polygon(const int32_t* pts, int n_pts, double distance, int32_t* out, int cap_pts, int* n_out) {
    using namespace ocrvi;
    OCRVI_CHECK(pts && out && n_out && n_pts >= 0 && cap_pts >= 0 && distance > 0, OCRVI_EINVAL, "unclip_polygon: bad argument");
    std::vector<Pt> in(n_pts), res;
    for (int i = 0; i < n_pts; ++i) in[i] = {pts[2 * i], pts[2 * i + 1]};
    unclip_polygon(in, distance, res);
    OCRVI_CHECK((int)res.size() <= cap_pts, OCRVI_ENOMEM, "unclip_polygon: %d points exceed the capacity %d", (int)res.size(), cap_pts);
    for (size_t i = 0; i < res.size(); ++i) { out[2 * i] = res[i].x; out[2 * i + 1] = res[i].y; }
    *n_out = (int)res.size();
    return OCRVI_OK;
}

static int db_postprocess_core(const float* prob, const uint32_t* bits, int H, int W, float thresh, float box_thresh, int max_candidates,
                               float unclip_ratio, float min_area, int32_t* points, int cap_points, int32_t* box_offsets, float* scores,
                               int cap_boxes, int* n_boxes) {
    using namespace ocrvi;
    OCRVI_CHECK(prob && points && box_offsets && scores && n_boxes && H > 0 && W > 0 && cap_points > 0 && cap_boxes > 0, OCRVI_EINVAL,
                "db_postprocess: bad argument");
    std::vector<std::vector<Pt>> contours;
    find_contours(prob, thresh, H, W, contours, bits);
    int nb = 0, np = 0;
    box_offsets[0] = 0;
    std::vector<Pt> approx, box;
    int idx = 0;
    for (auto it = contours.rbegin(); it != contours.rend(); ++it, ++idx) {   // cv2 returns the last-found contour first
        if (idx >= max_candidates) break;
        const double eps = 0.002 * arc_length_closed(*it);
        approx_poly_closed(*it, eps, approx);
        if (approx.size() < 4) continue;
        const double score = box_score(prob, H, W, approx);
        if ((double)box_thresh > score) continue;
        const double area = shoelace_abs_half(approx);
        if (area < (double)min_area) continue;
        double length = 0;
        for (size_t i = 0; i < approx.size(); ++i) {
            const Pt &a = approx[i], &b = approx[(i + 1) % approx.size()];
            const double dx = (double)(b.x - a.x), dy = (double)(b.y - a.y);
            length += sqrt(dx * dx + dy * dy);
        }
        if (length == 0) continue;
        const double distance = area * (double)unclip_ratio / length;
        if (distance <= 0) continue;
        unclip_polygon(approx, distance, box);
        if (box.size() < 4) continue;
        OCRVI_CHECK(nb < cap_boxes && np + (int)box.size() <= cap_points, OCRVI_ENOMEM, "db_postprocess: output capacity exceeded (%d boxes, %d points)",
                    cap_boxes, cap_points);
        for (const Pt& p : box) { points[2 * np] = p.x; points[2 * np + 1] = p.y; ++np; }
        scores[nb] = (float)score;
        box_offsets[++nb] = np;
    }
    *n_boxes = nb;
    return OCRVI_OK;
}

extern "C" int ocrvi_db_postprocess(const float* prob, int H, int W, float thresh, float box_thresh, int max_candidates, float unclip_ratio,
                                    float min_area, int32_t* points, int cap_points, int32_t* box_offsets, float* scores, int cap_boxes,
                                    int* n_boxes) {
    return db_postprocess_core(prob, nullptr, H, W, thresh, box_thresh, max_candidates, unclip_ratio, min_area, points, cap_points, box_offsets,
                               scores, cap_boxes, n_boxes);
}

// The host middle of the reference's per-image loop for a batch of pages (src/pipeline/pipeline2.py:320-343): DBPostProcessor on each
// page's probability map -> boxes rescaled to the original image (int64 truncation of :324-328) -> the clamped bounding rectangle
// crop_image slices (src/det/test.py:123-130).  Pages are independent, so they are spread over `threads` host threads (the reference
// handles one page at a time in Python); everything stays in C so no interpreter lock is held.
namespace {
struct PageView { const float* prob; const uint32_t* bits; bool ok; };

// Shared body of the batch entries: page(pg, slot) hands over the map (and optionally its bit mask) of page pg for host thread `slot`.
template <typename PageFn>
int boxes_batch_impl(PageFn page, int n_pages, int H, int W, float thresh, float box_thresh, int max_candidates, float unclip_ratio,
                     float min_area, double scale_w, double scale_h, int orig_h, int orig_w, int page_base, int32_t* rects, float* scores,
                     int cap_per_page, int32_t* counts, int threads, int32_t* skipped) {
    using namespace ocrvi;
    std::atomic<int> next(0), failed(0);
    static std::vector<std::vector<signed char>> label_pool;    // padded label images, kept across calls (page faults cost more than the scan)
    const int nt = std::max(1, std::min(threads, n_pages));
    if ((int)label_pool.size() < nt) label_pool.resize(nt);
    auto work = [&](int slot) {
        label_buf = &label_pool[slot];
        std::vector<int32_t> pts;
        std::vector<int32_t> offs((size_t)cap_per_page + 1);
        std::vector<float> sc((size_t)cap_per_page);
        for (;;) {
            const int pg = next.fetch_add(1);
            if (pg >= n_pages) break;
            const PageView pv = page(pg, slot);
            if (skipped) skipped[pg] = pv.ok ? 0 : 1;
            if (!pv.ok) { counts[pg] = 0; continue; }
            int nb = 0, rc;
            size_t cap_pts = (size_t)4 * (H + W) + 4096;
            for (;;) {   // polygon vertex capacity is a guess: grow on OCRVI_ENOMEM
                pts.resize(2 * cap_pts);
                rc = db_postprocess_core(pv.prob, pv.bits, H, W, thresh, box_thresh, max_candidates, unclip_ratio, min_area, pts.data(), (int)cap_pts,
                                         offs.data(), sc.data(), cap_per_page, &nb);
                if (rc != OCRVI_ENOMEM || cap_pts >= ((size_t)1 << 26)) break;
                cap_pts *= 4;
            }
            if (rc != OCRVI_OK) { failed.store(rc); counts[pg] = 0; continue; }
            int32_t* out = rects + (size_t)pg * cap_per_page * 5;
            for (int b = 0; b < nb; ++b) {
                long long x0 = 0, x1 = 0, y0 = 0, y1 = 0;
                for (int i = offs[b]; i < offs[b + 1]; ++i) {
                    // rescaled_box[:, 0] = rescaled_box[:, 0] / scale_w into an integer array: C truncation toward zero; then astype(int32)
                    const long long x = (long long)((double)pts[2 * i] / scale_w), y = (long long)((double)pts[2 * i + 1] / scale_h);
                    if (i == offs[b]) { x0 = x1 = x; y0 = y1 = y; }
                    x0 = std::min(x0, x); x1 = std::max(x1, x); y0 = std::min(y0, y); y1 = std::max(y1, y);
                }
                const long long bw = x1 - x0 + 1, bh = y1 - y0 + 1;          // cv2.boundingRect of integer points is inclusive
                const long long cx = std::max(0LL, x0), cy = std::max(0LL, y0);
                const long long cw = std::max(0LL, std::min(bw, (long long)orig_w - cx)), ch = std::max(0LL, std::min(bh, (long long)orig_h - cy));
                out[5 * b] = page_base + pg; out[5 * b + 1] = (int32_t)cx; out[5 * b + 2] = (int32_t)cy; out[5 * b + 3] = (int32_t)cw; out[5 * b + 4] = (int32_t)ch;
                if (scores) scores[(size_t)pg * cap_per_page + b] = sc[b];
            }
            counts[pg] = nb;
        }
        label_buf = nullptr;
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& t : pool) t.join();
    const int rc = failed.load();
    OCRVI_CHECK(rc == OCRVI_OK, rc, "db_boxes_batch: a page failed (%s)", ocrvi_last_error());
    return OCRVI_OK;
}
std::mutex g_batch_mu;   // one batch call at a time per process (it uses every core it is given anyway)
}  // namespace

extern "C" int ocrvi_db_boxes_batch(const float* prob, int n_pages, int H, int W, float thresh, float box_thresh, int max_candidates,
                                    float unclip_ratio, float min_area, double scale_w, double scale_h, int orig_h, int orig_w, int page_base,
                                    int32_t* rects, float* scores, int cap_per_page, int32_t* counts, int threads) {
    using namespace ocrvi;
    OCRVI_CHECK(prob && rects && counts && n_pages > 0 && H > 0 && W > 0 && cap_per_page > 0 && scale_w > 0 && scale_h > 0 && orig_h > 0 && orig_w > 0,
                OCRVI_EINVAL, "db_boxes_batch: bad argument");
    std::lock_guard<std::mutex> lk(g_batch_mu);
    auto page = [&](int pg, int) { return PageView{prob + (size_t)pg * H * W, nullptr, true}; };
    return boxes_batch_impl(page, n_pages, H, W, thresh, box_thresh, max_candidates, unclip_ratio, min_area, scale_w, scale_h, orig_h, orig_w,
                            page_base, rects, scores, cap_per_page, counts, threads, nullptr);
}

// The same stage fed by the device half (ocrvi_db_components, dbcomp.hip): per page the 1-bit thresholded mask, the component table
// (8 ints per component: x0, y0, x1, y1, count, root, sum) with its count, and the probability values inside the component boxes packed
// back to back (offsets[id] = start of box id, offsets[n] = total).  The segmentation is taken from the mask; the map is rebuilt only
// inside the component boxes -- every polygon box_score_fast looks at lies inside the box of the component its contour belongs to -- so
// the results equal ocrvi_db_boxes_batch's on the full map.  A page whose table (count > cap) or boxes (total > pack_cap) overflowed is
// not processed: skipped[page] = 1 and the caller falls back to the full map for it.
extern "C" int ocrvi_db_boxes_batch_sparse(const uint32_t* mask_bits, const int32_t* comps, const int32_t* comp_counts, int cap,
                                           const long long* offsets, const float* packed, long long pack_cap, int n_pages, int H, int W,
                                           float box_thresh, int max_candidates, float unclip_ratio, float min_area, double scale_w, double scale_h,
                                           int orig_h, int orig_w, int page_base, int32_t* rects, float* scores, int cap_per_page, int32_t* counts,
                                           int threads, int32_t* skipped) {
    using namespace ocrvi;
    OCRVI_CHECK(mask_bits && comps && comp_counts && offsets && packed && rects && counts && skipped && cap > 0 && pack_cap > 0 && n_pages > 0 &&
                    H > 0 && W > 0 && W % 32 == 0 && cap_per_page > 0 && scale_w > 0 && scale_h > 0 && orig_h > 0 && orig_w > 0,
                OCRVI_EINVAL, "db_boxes_batch_sparse: bad argument");
    std::lock_guard<std::mutex> lk(g_batch_mu);
    static std::vector<std::vector<float>> map_pool;   // one page-sized map per host thread; only component boxes are ever (re)written or read
    const int nt = std::max(1, std::min(threads, n_pages));
    if ((int)map_pool.size() < nt) map_pool.resize(nt);
    auto page = [&](int pg, int slot) {
        const int n = comp_counts[pg];
        const long long* o = offsets + (size_t)pg * (cap + 1);
        if (n > cap || o[std::min(n, cap)] > pack_cap) return PageView{nullptr, nullptr, false};
        std::vector<float>& m = map_pool[slot];
        m.resize((size_t)H * W);
        const float* src = packed + (size_t)pg * pack_cap;
        for (int id = 0; id < n; ++id) {
            const int32_t* c = comps + ((size_t)pg * cap + id) * 8;
            const int x0 = c[0], y0 = c[1], w = c[2] - c[0] + 1, h = c[3] - c[1] + 1;
            if (x0 < 0 || y0 < 0 || w <= 0 || h <= 0 || x0 + w > W || y0 + h > H) return PageView{nullptr, nullptr, false};   // corrupt table
            for (int yy = 0; yy < h; ++yy) memcpy(m.data() + (size_t)(y0 + yy) * W + x0, src + o[id] + (size_t)yy * w, (size_t)w * sizeof(float));
        }
        return PageView{m.data(), mask_bits + (size_t)pg * H * (W >> 5), true};
    };
    return boxes_batch_impl(page, n_pages, H, W, 0.f, box_thresh, max_candidates, unclip_ratio, min_area, scale_w, scale_h, orig_h, orig_w, page_base,
                            rects, scores, cap_per_page, counts, threads, skipped);
}
