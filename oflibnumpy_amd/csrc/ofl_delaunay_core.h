// ofl_delaunay_core.h -- geometry core of the exact scatter path: the Delaunay star of ONE site by half-plane clipping.
//
// scipy.interpolate.griddata(..., 'linear') (src/oflibnumpy/utils.py:253; flow_class.py:1407) interpolates on the Delaunay
// triangulation Qhull builds from the warped points.  The triangles of that triangulation incident to a site p are
// its Delaunay star: the sites whose Voronoi cells touch p's, in angular order.  The Voronoi cell of p is the
// intersection of the half-planes { v : |v - p| <= |v - c| } over all other sites c; it is built here by clipping a
// huge box with the bisectors of candidate sites, nearest first.  Once every site within twice the distance of the
// farthest cell vertex has been applied the cell is final (security radius), so a star needs only a local
// neighbourhood -- stars are independent of each other and map to one GPU thread (or workgroup) per site.
//
// Robustness: whether candidate c cuts the cell vertex between the edges of sites a and b is the in-circle
// predicate "c inside circle(p, a, b)"; it is evaluated from the SITE coordinates (float64, relative to p) whenever
// the cheap test on the stored vertex coordinates is not decisive, so all stars decide a near-degenerate
// configuration from the same determinant.  Exactly co-circular sites (|det| at rounding level) are legitimately
// ambiguous -- Qhull's own choice there is arbitrary -- and the raster pass resolves overlapping stars by triangle id.
//
// The header compiles for the device (hipcc) and for the host (g++, tests/native/dl_core_cpu.cpp: the CPU suite
// checks this very code against SciPy's Delaunay on the reference fixtures).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define DL_HD __host__ __device__ __forceinline__
#else
#define DL_HD inline
#endif

namespace ofl_dl {

struct P2 { double x, y; };

constexpr double kBox = 1e9;          // half size of the initial cell; cell vertices beyond it mean "unbounded"
constexpr double kDecide = 1e-9;      // relative margin inside which the vertex test defers to the in-circle predicate

// Polygon of the cell under construction: vertex k at (vx, vy)[k * stride] relative to the site, tag[k * stride] =
// the site whose bisector carries the edge from vertex k to vertex k + 1 (-1 .. -4: the box sides y = -B, x = B,
// y = B, x = -B).  Counter-clockwise in (x, y).
template <typename R>       // R = double, or float for the per-thread pass (half the LDS; decisions that come close go to
struct PolyT {              // the in-circle predicate on the float64 SITE coordinates either way)
    R   *vx, *vy;
    int *tag;
    int  stride, cap, n;
    DL_HD R   &X(int k) const { return vx[k * stride]; }
    DL_HD R   &Y(int k) const { return vy[k * stride]; }
    DL_HD int &T(int k) const { return tag[k * stride]; }
    // relative margin inside which the test on the stored vertex coordinates defers to the predicate
    static constexpr double decide = sizeof(R) == 8 ? kDecide : 4e-6;
};
using Poly = PolyT<double>;

template <class PolyX>
DL_HD void poly_init(PolyX &P)
{
    P.X(0) = -kBox; P.Y(0) = -kBox; P.T(0) = -1;
    P.X(1) =  kBox; P.Y(1) = -kBox; P.T(1) = -2;
    P.X(2) =  kBox; P.Y(2) =  kBox; P.T(2) = -3;
    P.X(3) = -kBox; P.Y(3) =  kBox; P.T(3) = -4;
    P.n = 4;
}

// > 0 when c lies inside the circle through the origin, a and b (any orientation of a, b); exactly 0 for four
// co-circular points in exact arithmetic that float64 represents exactly (lattice points: translations, the tiled
// Sintel field) -- vertex_cut breaks such ties by index
DL_HD double incircle_origin(const P2 &a, const P2 &b, const P2 &c)
{
    const double a2 = a.x * a.x + a.y * a.y, b2 = b.x * b.x + b.y * b.y, c2 = c.x * c.x + c.y * c.y;
    const double det3 = a.x * (b.y * c2 - b2 * c.y) - a.y * (b.x * c2 - b2 * c.x) + a2 * (b.x * c.y - b.y * c.x);
    const double o = a.x * b.y - a.y * b.x;
    return o > 0.0 ? -det3 : (o < 0.0 ? det3 : 0.0);
}

// line n . v = h carrying the edges of `tag`
template <class RelFn>
DL_HD void edge_line(int tag, RelFn rel, double &nx, double &ny, double &h)
{
    if (tag >= 0) { const P2 t = rel(tag); nx = t.x; ny = t.y; h = 0.5 * (t.x * t.x + t.y * t.y); }
    else if (tag == -1) { nx = 0.0; ny = -1.0; h = kBox; }
    else if (tag == -2) { nx = 1.0; ny = 0.0; h = kBox; }
    else if (tag == -3) { nx = 0.0; ny = 1.0; h = kBox; }
    else { nx = -1.0; ny = 0.0; h = kBox; }
}

// does the bisector of candidate c (relative position C, h = |C|^2 / 2) cut off vertex k of an n-vertex polygon?
// Exact ties (determinant == 0: the sites p, a, c, b lie on one circle) are broken the way the certified mesh path
// breaks them for its square cells: of the two diagonals of the quadrilateral the one through the site with the
// smallest index exists.  c cutting the vertex makes p - c an edge, so it cuts iff the smallest index is p's or c's;
// seen from any of the four sites the same diagonal wins, so their stars agree.
template <class PolyX, class RelFn>
DL_HD bool vertex_cut(const PolyX &P, int k, int n, const P2 &C, int ctag, int ptag, double h, RelFn rel)
{
    const int ta = P.T(k == 0 ? n - 1 : k - 1), tb = P.T(k);
    if (ctag == ta || ctag == tb) return false;       // the candidate already carries an edge at this vertex
    const double tx = (double)P.X(k) * C.x, ty = (double)P.Y(k) * C.y;
    const double d = tx + ty - h;
    const double m = PolyX::decide * (fabs(tx) + fabs(ty) + h);
    if (d > m) return true;
    if (d < -m) return false;
    if (ta >= 0 && tb >= 0 && ta != tb) {
        const P2 A = rel(ta), B = rel(tb);
        if ((A.x == C.x && A.y == C.y) || (B.x == C.x && B.y == C.y)) return false;     // a duplicate of an edge's site
        if (A.x * B.y - A.y * B.x != 0.0) {          // (p, a, b collinear: parallel bisectors, the vertex is a box vertex)
            const double ic = incircle_origin(A, B, C);
            if (ic != 0.0) return ic > 0.0;
            const int lo_pc = ptag < ctag ? ptag : ctag, lo_ab = ta < tb ? ta : tb;
            return lo_pc < lo_ab;
        }
    }
    // a vertex on the box (unbounded cell): the stored coordinates (float32 in the per-thread pass, and ~1e9 in size)
    // cannot resolve a candidate whose bisector passes within a few units of it -- re-derive the vertex in float64 from
    // the two lines that define it
    double ax, ay, ah, bx, by, bh;
    edge_line(ta, rel, ax, ay, ah);
    edge_line(tb, rel, bx, by, bh);
    const double det = ax * by - bx * ay;
    if (det != 0.0) {
        const double vx = (ah * by - bh * ay) / det, vy = (ax * bh - bx * ah) / det;
        if (isfinite(vx) && isfinite(vy)) return vx * C.x + vy * C.y - h > 0.0;
    }
    return d > 0.0;
}

// intersection of the edge line of `tag` with the bisector (C, h); falls back to the point of the segment u -> w
// where the bisector's signed distance changes sign when the two lines are parallel to rounding
template <class RelFn>
DL_HD P2 cut_point(int tag, const P2 &C, double h, RelFn rel, double ux, double uy, double wx, double wy)
{
    double nx, ny, hh;
    edge_line(tag, rel, nx, ny, hh);
    const double det = nx * C.y - C.x * ny;
    P2 r;
    if (fabs(det) > 1e-300 * (fabs(nx * C.y) + fabs(C.x * ny)) && det != 0.0) {
        r.x = (hh * C.y - h * ny) / det;
        r.y = (nx * h - C.x * hh) / det;
        if (isfinite(r.x) && isfinite(r.y)) return r;
    }
    const double du = ux * C.x + uy * C.y - h, dw = wx * C.x + wy * C.y - h;
    double t = du / (du - dw);
    if (!(t >= 0.0 && t <= 1.0)) t = 0.5;
    r.x = ux + t * (wx - ux); r.y = uy + t * (wy - uy);
    return r;
}

// Clips the polygon with the bisector of candidate `ctag` at relative position C.  Returns 0 (unchanged), 1 (clipped)
// or -1 (the polygon would exceed its capacity; it is left unchanged).  Sequential; n <= 64.
template <class PolyX, class RelFn>
DL_HD int poly_clip(PolyX &P, const P2 &C, int ctag, int ptag, RelFn rel)
{
    const double h = 0.5 * (C.x * C.x + C.y * C.y);
    const int n = P.n;
    unsigned long long cut = 0;
    for (int k = 0; k < n; ++k)
        if (vertex_cut(P, k, n, C, ctag, ptag, h, rel)) cut |= 1ull << k;
    if (!cut) return 0;
    const unsigned long long full = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    if (cut == full) return 0;                       // the site itself is inside every half-plane: rounding only
    int a = -1;
    for (int k = 0; k < n; ++k)
        if (((cut >> k) & 1ull) && !((cut >> (k == 0 ? n - 1 : k - 1)) & 1ull)) { a = k; break; }
    int L = 0;
    while ((cut >> ((a + L) % n)) & 1ull) ++L;        // a convex cell loses ONE run of vertices (further runs: rounding, ignored)
    const int b = (a + L - 1) % n, ia = a == 0 ? n - 1 : a - 1, ib = (b + 1) % n;
    const int n2 = n - L + 2;
    if (n2 > P.cap) return -1;
    const int tb = P.T(b);
    const P2 v1 = cut_point(P.T(ia), C, h, rel, P.X(ia), P.Y(ia), P.X(a), P.Y(a));
    const P2 v2 = cut_point(tb, C, h, rel, P.X(b), P.Y(b), P.X(ib), P.Y(ib));
    if (a <= b) {
        const int shift = 2 - L;
        if (shift > 0) { for (int k = n - 1; k > b; --k) { P.X(k + shift) = P.X(k); P.Y(k + shift) = P.Y(k); P.T(k + shift) = P.T(k); } }
        else if (shift < 0) { for (int k = b + 1; k < n; ++k) { P.X(k + shift) = P.X(k); P.Y(k + shift) = P.Y(k); P.T(k + shift) = P.T(k); } }
        P.X(a) = v1.x; P.Y(a) = v1.y; P.T(a) = ctag;
        P.X(a + 1) = v2.x; P.Y(a + 1) = v2.y; P.T(a + 1) = tb;
    } else {
        const int m = a - (b + 1);
        for (int j = 0; j < m; ++j) { P.X(j) = P.X(b + 1 + j); P.Y(j) = P.Y(b + 1 + j); P.T(j) = P.T(b + 1 + j); }
        P.X(m) = v1.x; P.Y(m) = v1.y; P.T(m) = ctag;
        P.X(m + 1) = v2.x; P.Y(m + 1) = v2.y; P.T(m + 1) = tb;
    }
    P.n = n2;
    return 1;
}

// squared distance of the farthest cell vertex from the site
template <class PolyX>
DL_HD double poly_rmax2(const PolyX &P)
{
    double r2 = 0.0;
    for (int k = 0; k < P.n; ++k) r2 = fmax(r2, (double)P.X(k) * (double)P.X(k) + (double)P.Y(k) * (double)P.Y(k));
    return sizeof(P.X(0)) == 8 ? r2 : r2 * 1.00001;
}

// bucket grid over the bounding box of the sites
struct Grid {
    double ox, oy, s, inv_s;     // origin, cell size
    int    gx, gy;
    DL_HD int bx(double x) const { const double f = floor((x - ox) * inv_s); return f < 0.0 ? 0 : (f >= (double)gx ? gx - 1 : (int)f); }
    DL_HD int by(double y) const { const double f = floor((y - oy) * inv_s); return f < 0.0 ? 0 : (f >= (double)gy ? gy - 1 : (int)f); }
};

// All sites of the buckets on the Chebyshev ring r around bucket (bx, by) are applied to the cell of site p (at pp).
// Buckets are stored row-major and `sorted` lists the sites bucket by bucket, so a run of buckets in one row is ONE
// contiguous range of `sorted`.  Returns -1 when the polygon overflows.
// `reach2` = (2 * farthest cell vertex)^2, kept current by the caller's polygon: a site at or beyond that distance cannot
// cut any vertex (|v - c| < |v| implies |c| < 2 |v|), which spares the vertex loop for most sites of the outer rings.
template <class PolyX, class PosFn>
DL_HD int apply_ring(PolyX &P, int p, const P2 &pp, int bx, int by, int r, const Grid &g,
                     const unsigned *bstart, const unsigned *sorted, PosFn pos, double &reach2)
{
    auto rel = [&](int t) { const P2 q = pos(t); return P2{ q.x - pp.x, q.y - pp.y }; };
    auto run = [&](int row, int x0, int x1) -> int {
        if (row < 0 || row >= g.gy) return 0;
        if (x0 < 0) x0 = 0;
        if (x1 > g.gx - 1) x1 = g.gx - 1;
        if (x1 < x0) return 0;
        const unsigned lo = bstart[(size_t)row * g.gx + x0], hi = bstart[(size_t)row * g.gx + x1 + 1];
        for (unsigned j = lo; j < hi; j += 4) {
            // four candidates at a time: their indices, then their positions, are in flight together
            int c[4];
            P2  q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) c[k] = j + k < hi ? (int)sorted[j + k] : -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = c[k] >= 0 ? pos(c[k]) : pp;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (c[k] < 0 || c[k] == p) continue;
                const P2 C = { q[k].x - pp.x, q[k].y - pp.y };
                const double d2 = C.x * C.x + C.y * C.y;
                if (d2 == 0.0 || d2 >= reach2) continue;      // a duplicate of p (same cell), or too far to matter
                const int rc = poly_clip(P, C, c[k], p, rel);
                if (rc < 0) return -1;
                if (rc > 0) reach2 = 4.0 * poly_rmax2(P);
            }
        }
        return 0;
    };
    if (r == 0) return run(by, bx, bx);
    if (run(by - r, bx - r, bx + r) < 0 || run(by + r, bx - r, bx + r) < 0) return -1;
    for (int row = by - r + 1; row <= by + r - 1; ++row) {
        if (bx - r >= 0 && run(row, bx - r, bx - r) < 0) return -1;
        if (bx + r <= g.gx - 1 && run(row, bx + r, bx + r) < 0) return -1;
    }
    return 0;
}

// The star of site p from the sites within `rings` bucket rings.  Returns 1 when the cell is final (every site within
// twice the farthest cell vertex has been applied: sites in unvisited buckets are at least r * s away), 0 when it is
// not (unbounded cells, rims of large holes: the far pass finishes those), -1 when the polygon overflowed.
template <class PolyX, class PosFn>
DL_HD int star_near(PolyX &P, int p, const P2 &pp, const Grid &g, const unsigned *bstart, const unsigned *sorted,
                    PosFn pos, int rings)
{
    poly_init(P);
    const int bx = g.bx(pp.x), by = g.by(pp.y);
    double reach2 = 4.0 * poly_rmax2(P);
    for (int r = 0; r <= rings; ++r) {
        if (apply_ring(P, p, pp, bx, by, r, g, bstart, sorted, pos, reach2) < 0) return -1;
        const double cover = (double)r * g.s;
        if (cover * cover >= reach2) return 1;
    }
    return 0;
}

// Rotation of a triangle's vertex indices that puts the smallest first without changing the cyclic order: every copy
// of a triangle (each of its three sites emits it) then interpolates with identical arithmetic.
DL_HD void canonical3(unsigned &i0, unsigned &i1, unsigned &i2)
{
    if (i1 < i0 && i1 < i2) { const unsigned t = i0; i0 = i1; i1 = i2; i2 = t; }
    else if (i2 < i0 && i2 < i1) { const unsigned t = i2; i2 = i1; i1 = i0; i0 = t; }
}

}  // namespace ofl_dl
