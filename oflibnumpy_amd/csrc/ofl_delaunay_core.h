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

#ifndef DL_DBG
#define DL_DBG(i, v)
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

// The same question for the clip's near-tie branch, with "on the circle as far as float64 can tell" made explicit: 0 when the
// determinant is within 4e-15 of the sum of its absolute terms (Shewchuk's static filter for this expression is 1.1e-15).  The
// four sites of such a question take part in four stars, each asking from its own point of view with coordinates relative to
// ITS site, and the stars only fit together if all four answer alike.  Sites that are co-circular in exact arithmetic (every
// cell of a similarity transform whose float32 rounding repeats from row to row) came out as + 3.6e-15 from one site and as
// - 0.0 from another: one star took the site in, its neighbour's did not, and their triangles overlapped (soak seed 3000265).
// With the filter every site calls such a set a TIE and vertex_cut_ex's index rule -- which is globally consistent: among
// co-circular sites the diagonals through the smallest index exist -- decides for all of them.  (Evaluating the determinant
// once per SET of sites -- indices sorted, coordinates relative to the smallest one, so that everybody gets the same bits --
// was built as well: four more position loads in a branch that similarity fields and lattices take all the time, + 5 - 9 %
// on every Delaunay-path case; what it adds over the filter is the set of determinants within rounding of the THRESHOLD.)
DL_HD double incircle_origin_filtered(const P2 &a, const P2 &b, const P2 &c)
{
    const double a2 = a.x * a.x + a.y * a.y, b2 = b.x * b.x + b.y * b.y, c2 = c.x * c.x + c.y * c.y;
    const double det3 = a.x * (b.y * c2 - b2 * c.y) - a.y * (b.x * c2 - b2 * c.x) + a2 * (b.x * c.y - b.y * c.x);
    const double perm = fabs(a.x) * (fabs(b.y) * c2 + b2 * fabs(c.y)) + fabs(a.y) * (fabs(b.x) * c2 + b2 * fabs(c.x))
                      + a2 * (fabs(b.x * c.y) + fabs(b.y * c.x));
    if (!(fabs(det3) > 4e-15 * perm)) return 0.0;
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
// vertex_cut_ex: 0 = not cut, 2 = cut by the test on the stored vertex (beyond its margin), 1 = cut by the in-circle predicate,
// the index rule for exact ties or the re-derived vertex -- a decision taken vertex by vertex, so that coincident vertices
// (four co-circular sites: a zero-length edge) may come out "cut, not cut, cut" around the polygon; poly_cutmask and
// far_apply keep ONE run of cut vertices and prefer the one that holds a vertex of kind 2.
template <class PolyX, class RelFn>
DL_HD int vertex_cut_ex(const PolyX &P, int k, int n, const P2 &C, int ctag, int ptag, double h, RelFn rel)
{
    const int ta = P.T(k == 0 ? n - 1 : k - 1), tb = P.T(k);
    if (ctag == ta || ctag == tb) return 0;           // the candidate already carries an edge at this vertex
    if (sizeof(P.X(0)) == 4) {
        // float32 cells (the per-thread passes, which are bound by exactly these instructions): the first look in float32
        // as well -- the stored vertex is only good to 6e-8 anyway, and the margin below covers the rounding of three more
        // float32 operations many times over; what it cannot decide goes through the float64 test and the predicate as before
        const float cx = (float)C.x, cy = (float)C.y, hf = (float)h;
        const float txf = (float)P.X(k) * cx, tyf = (float)P.Y(k) * cy, df = txf + tyf - hf;
        const float mf = 8e-6f * (fabsf(txf) + fabsf(tyf) + hf);
        if (df > mf) return 2;
        if (df < -mf) return 0;
    }
    const double tx = (double)P.X(k) * C.x, ty = (double)P.Y(k) * C.y;
    const double d = tx + ty - h;
    const double m = PolyX::decide * (fabs(tx) + fabs(ty) + h);
    if (d > m) return 2;
    if (d < -m) return 0;
    if (ta >= 0 && tb >= 0 && ta != tb) {
        const P2 A = rel(ta), B = rel(tb);
        if ((A.x == C.x && A.y == C.y) || (B.x == C.x && B.y == C.y)) return 0;         // a duplicate of an edge's site
        if (A.x * B.y - A.y * B.x != 0.0) {          // (p, a, b collinear: parallel bisectors, the vertex is a box vertex)
            const double ic = incircle_origin_filtered(A, B, C);
            if (ic != 0.0) return ic > 0.0 ? 1 : 0;
            const int lo_pc = ptag < ctag ? ptag : ctag, lo_ab = ta < tb ? ta : tb;
            return lo_pc < lo_ab ? 1 : 0;
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
        if (isfinite(vx) && isfinite(vy)) return vx * C.x + vy * C.y - h > 0.0 ? 1 : 0;
    }
    return d > 0.0 ? 1 : 0;
}

template <class PolyX, class RelFn>
DL_HD bool vertex_cut(const PolyX &P, int k, int n, const P2 &C, int ctag, int ptag, double h, RelFn rel)
{
    return vertex_cut_ex(P, k, n, C, ctag, ptag, h, rel) != 0;
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

// Which vertices of the polygon does the bisector of candidate `ctag` at relative position C cut off?  (bit k = vertex k;
// n <= 64.)  0 also when EVERY vertex would go: the site itself is inside every half-plane, so that is rounding only.
template <class PolyX, class RelFn>
DL_HD unsigned long long poly_cutmask(const PolyX &P, const P2 &C, int ctag, int ptag, RelFn rel)
{
    const double h = 0.5 * (C.x * C.x + C.y * C.y);
    const int n = P.n;
    unsigned long long cut = 0, sure = 0;
    for (int k = 0; k < n; ++k) {
        const int c = vertex_cut_ex(P, k, n, C, ctag, ptag, h, rel);
        if (c) cut |= 1ull << k;
        if (c == 2) sure |= 1ull << k;
    }
    const unsigned long long full = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    if (cut == 0ull || cut == full) return 0ull;
    if (cut == sure) return cut;                      // no decision by predicate or tie rule: the run is what the plain test says
    // A convex cell loses ONE run of vertices.  Ties decided vertex by vertex can flag a second one (a coincident pair of
    // vertices on the bisector, one "cut", one not, next to the run that really goes): keep the first run that holds a
    // vertex the plain test cut beyond its margin -- the first run if there is none.  (Removing the tie's run instead left
    // the half-plane unapplied: the site was missing from the star, which then disagreed with its neighbours'.)
    const unsigned long long prev = ((cut << 1) | (cut >> (n - 1))) & full;         // bit k = cut[k - 1]
    const unsigned long long starts = cut & ~prev;
    if ((starts & (starts - 1ull)) == 0ull) return cut;                               // one run
    unsigned long long first = 0ull, chosen = 0ull;
    for (int s = 0; s < n && !chosen; ++s) {
        if (!((starts >> s) & 1ull)) continue;
        unsigned long long run = 0ull;
        for (int k = s, m = 0; m < n && ((cut >> k) & 1ull); k = k + 1 == n ? 0 : k + 1, ++m) run |= 1ull << k;
        if (!first) first = run;
        if (run & sure) chosen = run;
    }
    return chosen ? chosen : first;
}

// Removes the vertices of a non-empty `cut` mask (poly_cutmask) and closes the polygon with the candidate's edge.
// Returns 1, or -1 when the polygon would exceed its capacity (it is left unchanged).
template <class PolyX, class RelFn>
DL_HD int poly_apply(PolyX &P, const P2 &C, int ctag, RelFn rel, unsigned long long cut)
{
    const double h = 0.5 * (C.x * C.x + C.y * C.y);
    const int n = P.n;
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

// Clips the polygon with the bisector of candidate `ctag` at relative position C.  Returns 0 (unchanged), 1 (clipped)
// or -1 (the polygon would exceed its capacity; it is left unchanged).  Sequential; n <= 64.
template <class PolyX, class RelFn>
DL_HD int poly_clip(PolyX &P, const P2 &C, int ctag, int ptag, RelFn rel)
{
    const unsigned long long cut = poly_cutmask(P, C, ctag, ptag, rel);
    return cut ? poly_apply(P, C, ctag, rel, cut) : 0;
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

// Dense clusters of DISTINCT sites -- a flow that contracts a block of the image a hundredfold puts ten thousand sites into
// each of a few buckets of the uniform grid (scipy's Qhull triangulates such a set like any other: src/oflibnumpy/utils.py:253).
// Every bucket of more than kHeavy entries gets a grid of its own over the bounding box of its sites, about two sites per
// cell, and its stretch of the sorted list is re-ordered cell by cell (dl_sub_bin_kernel).  A ring search that meets such a
// bucket does not scan it: it walks the rings of the bucket's grid around the point of the box nearest to the site -- sites
// in cells beyond ring r are at least r cell widths from that point, hence from the site -- until the reach of the cell
// under construction is covered.
constexpr unsigned kHeavy = 64;
constexpr unsigned kHeavyBudget = 1024;    // sites of heavy buckets one thread looks at for one star before it hands the star on
struct SubGrid  { Grid g; unsigned off, pad; };        // start[off .. off + gx * gy]: ABSOLUTE positions in the sorted list, cell by cell
struct SubGrids { const unsigned *bucket; const SubGrid *info; const unsigned *start; unsigned n; };     // bucket[]: ascending bucket numbers

DL_HD int find_heavy(const SubGrids &sub, unsigned bucket)
{
    unsigned lo = 0, hi = sub.n;
    while (lo < hi) { const unsigned mid = (lo + hi) >> 1; if (sub.bucket[mid] < bucket) lo = mid + 1; else hi = mid; }
    return lo < sub.n && sub.bucket[lo] == bucket ? (int)lo : -1;
}

// the candidates [lo, hi) of the sorted list, one after the other (the rare, compact form of apply_ring's inner loop)
template <class PolyX, class PosFn>
DL_HD int apply_range(PolyX &P, int p, const P2 &pp, unsigned lo, unsigned hi, const unsigned *sorted, const P2 *sorted_xy, PosFn pos, double &reach2)
{
    auto rel = [&](int t) { const P2 q = pos(t); return P2{ q.x - pp.x, q.y - pp.y }; };
    DL_DBG(1, hi - lo);
    for (unsigned j0 = lo; j0 < hi; j0 += 4) {
        int c4[4];
        P2  q4[4];                                           // four candidates' indices, then their positions, in flight together
        for (int k = 0; k < 4; ++k) {
            if (sorted_xy) {                                 // (as in apply_ring: unconditional loads at a clamped position)
                const unsigned jj = j0 + k < hi ? j0 + k : hi - 1;
                const int ck = (int)sorted[jj];
                q4[k] = sorted_xy[jj];
                c4[k] = j0 + k < hi ? ck : -1;
            } else {
                c4[k] = j0 + k < hi ? (int)sorted[j0 + k] : -1;
                q4[k] = c4[k] >= 0 ? pos(c4[k]) : pp;
            }
        }
        for (int k = 0; k < 4; ++k) {
            const int c = c4[k];
            if (c < 0 || c == p) continue;
            const P2 C = { q4[k].x - pp.x, q4[k].y - pp.y };
            const double d2 = C.x * C.x + C.y * C.y;
            if (d2 == 0.0 || d2 >= reach2) continue;
            const int rc = poly_clip(P, C, c, p, rel);
            if (rc < 0) return -1;
            if (rc > 0) reach2 = 4.0 * poly_rmax2(P);
        }
    }
    return 0;
}

// the buckets x0 .. x1 of row `row`, heavy ones through their own grids; not inlined: the ordinary ring search must not grow by it
template <class PolyX, class PosFn>
#if defined(__HIPCC__)
__host__ __device__ __noinline__
#endif
int apply_heavy_run(PolyX &P, int p, const P2 &pp, int row, int x0, int x1, const Grid &g, const unsigned *bstart,
                    const unsigned *sorted, const P2 *sorted_xy, PosFn pos, double &reach2, const SubGrids &sub, unsigned *budget)
{
    const bool give_up = budget != nullptr;
    DL_DBG(0, 1);
    for (int bx = x0; bx <= x1; ++bx) {
        const size_t b = (size_t)row * g.gx + bx;
        const unsigned lo = bstart[b], hi = bstart[b + 1];
        const int h = hi - lo > kHeavy ? find_heavy(sub, (unsigned)b) : -1;
        if (hi - lo > kHeavy && h < 0) DL_DBG(3, 1);
        if (h < 0) { if (apply_range(P, p, pp, lo, hi, sorted, sorted_xy, pos, reach2) < 0) return -1; continue; }
        const SubGrid sg = sub.info[h];
        const Grid &q = sg.g;
        const unsigned *st = sub.start + sg.off;
        // rings of the bucket's own grid around the point of its box nearest to the site
        const double cx = fmin(fmax(pp.x, q.ox), q.ox + q.s * q.gx), cy = fmin(fmax(pp.y, q.oy), q.oy + q.s * q.gy);
        const double d0 = (cx - pp.x) * (cx - pp.x) + (cy - pp.y) * (cy - pp.y);
        if (d0 >= reach2) continue;                              // the whole box is out of reach
        const int sx = q.bx(cx), sy = q.by(cy);
        const int ax = sx > q.gx - 1 - sx ? sx : q.gx - 1 - sx, ay = sy > q.gy - 1 - sy ? sy : q.gy - 1 - sy;
        const int rmax = ax > ay ? ax : ay;
        for (int r = 0; r <= rmax; ++r) {
            for (int yy = sy - r; yy <= sy + r; ++yy) {
                DL_DBG(5, 1);
                if (yy < 0 || yy >= q.gy) continue;
                const bool full = yy == sy - r || yy == sy + r;
                for (int part = 0; part < (full ? 1 : 2); ++part) {
                    int u0 = full ? sx - r : (part ? sx + r : sx - r), u1 = full ? sx + r : u0;
                    if (!full && (u0 < 0 || u0 >= q.gx)) continue;
                    if (u0 < 0) u0 = 0;
                    if (u1 > q.gx - 1) u1 = q.gx - 1;
                    if (u1 < u0) continue;
                    const unsigned c0 = st[(size_t)yy * q.gx + u0], c1 = st[(size_t)yy * q.gx + u1 + 1];
                    if (apply_range(P, p, pp, c0, c1, sorted, sorted_xy, pos, reach2) < 0) return -1;
                    if (give_up) *budget += c1 - c0;
                }
            }
            const double cover = (double)r * q.s;
            if (cover * cover >= reach2) break;           // (what lies beyond ring r is at least r cells from the box point, hence from the site)
            // A site ON THE RIM of a cluster, or next to one: its cell reaches out into the sparse surroundings, every site of the
            // cluster is within that reach -- ten thousand per bucket, each dragged through the vertex loop by ONE thread (a
            // few hundred such sites made the clip pass of a 4K field with a 400 x 400 block contracted 100 times 0.4 s long) --
            // and a rim cell does not close within the rings anyway.  A cell that has looked at more than kHeavyBudget sites
            // of heavy buckets (an interior site of a cluster needs a few hundred) is given up (1: the caller hands it on
            // unfinished, with the edges found so far as seeds): the cooperative passes scan clusters 64 sites at a time.
            if (give_up && *budget > kHeavyBudget) { DL_DBG(2, 1); return 1; }
        }
    }
    return 0;
}

// All sites of the buckets on the Chebyshev ring r around bucket (bx, by) are applied to the cell of site p (at pp).
// Buckets are stored row-major and `sorted` lists the sites bucket by bucket, so a run of buckets in one row is ONE
// contiguous range of `sorted`.  Returns -1 when the polygon overflows.
// `reach2` = (2 * farthest cell vertex)^2, kept current by the caller's polygon: a site at or beyond that distance cannot
// cut any vertex (|v - c| < |v| implies |c| < 2 |v|), which spares the vertex loop for most sites of the outer rings.
// HEAVY = false (the passes every field runs): a run that may hold a heavy bucket is not touched -- 2 is returned and the caller
// leaves the star to the HEAVY instantiation (a launch of its own, only for such sites): the few hundred lines of the heavy search,
// a call that is not inlined and a polygon whose address escapes cost the ordinary clip pass half its speed when they sat in it.
template <bool HEAVY = false, class PolyX, class PosFn>
#ifndef OFL_RING_W
#define OFL_RING_W 4
#endif
DL_HD int apply_ring(PolyX &P, int p, const P2 &pp, int bx, int by, int r, const Grid &g,
                     const unsigned *bstart, const unsigned *sorted, PosFn pos, double &reach2,
                     const P2 *sorted_xy = nullptr,          // positions in `sorted` order (one contiguous read per run) or null
                     const SubGrids *sub = nullptr,          // the grids of the heavy buckets of g (or null: every bucket is scanned)
                     unsigned *budget = nullptr)             // counts the sites of heavy buckets looked at; non-null: may return 1 (given up, see apply_heavy_run)
{
    auto rel = [&](int t) { const P2 q = pos(t); return P2{ q.x - pp.x, q.y - pp.y }; };
    // The ring as a sequence of runs -- top row, bottom row, then the two end buckets of the rows between -- walked by ONE
    // loop (four separate calls made four copies of the four-times-unrolled clip code, 250 KB per kernel).  The bucket
    // bounds of run k + 1 are requested before the candidates of run k are processed, which takes one of the two dependent
    // loads per run off a thread's chain (measured: no change in the kernels' times -- PMC shows 87 000 vector instructions
    // per wave on the rim of a large hole: the few hundred waves such fields leave to this pass compute, they do not wait).
    const int nseg = r == 0 ? 1 : 2 + 2 * (2 * r - 1);
    int row = 0, x0 = 0, x1 = -1;                        // the buckets of the run `bounds` looked up last
    auto bounds = [&](int seg, unsigned &lo, unsigned &hi) {
        if (r == 0) { row = by; x0 = bx; x1 = bx; }
        else if (seg == 0) { row = by - r; x0 = bx - r; x1 = bx + r; }
        else if (seg == 1) { row = by + r; x0 = bx - r; x1 = bx + r; }
        else {
            const int m = seg - 2;
            row = by - r + 1 + (m >> 1);
            x0 = x1 = (m & 1) ? bx + r : bx - r;
            if (x0 < 0 || x0 > g.gx - 1) { lo = hi = 0; return; }     // (a single bucket outside the grid must not be clamped onto the border one)
        }
        if (x0 < 0) x0 = 0;
        if (x1 > g.gx - 1) x1 = g.gx - 1;
        if (row < 0 || row >= g.gy || x1 < x0) { lo = hi = 0; return; }
        lo = bstart[(size_t)row * g.gx + x0]; hi = bstart[(size_t)row * g.gx + x1 + 1];
    };
    unsigned lo, hi;
    bounds(0, lo, hi);
    for (int seg = 0; seg < nseg; ++seg) {
        if (sub && hi - lo > kHeavy) {
            if constexpr (!HEAVY) return 2;                // (rare) a run that may hold a heavy bucket: for the HEAVY instantiation
            else {
                // bucket by bucket, heavy ones through their own grids
                const int hr = apply_heavy_run(P, p, pp, row, x0, x1, g, bstart, sorted, sorted_xy, pos, reach2, *sub, budget);
                if (hr != 0) return hr;                    // -1: overflow, 1: given up (the rim of a cluster)
                if (seg + 1 < nseg) bounds(seg + 1, lo, hi);
                continue;
            }
        }
        unsigned nlo = 0, nhi = 0;
        if (seg + 1 < nseg) bounds(seg + 1, nlo, nhi);
        DL_DBG(6, hi - lo);
        for (unsigned j = lo; j < hi; j += OFL_RING_W) {
            // a few candidates at a time: their indices, then their positions, are in flight together
            int c[OFL_RING_W];
            P2  q[OFL_RING_W];
            if (sorted_xy) {
                // With the positions stored in list order both loads depend on the list position alone: asked for unconditionally, at a
                // clamped position, ALL of them leave before the first is looked at.  (A load inside a branch is waited for where it
                // stands, and left alone the compiler sinks each candidate's loads to its use behind the previous candidate's clip:
                // a batch of four candidates was five round trips, not one.  The empty asm pins the raw values here.)
                int ck[OFL_RING_W];
#ifdef __HIPCC__
#pragma unroll
#endif
                for (int k = 0; k < OFL_RING_W; ++k) {
                    const unsigned jj = j + k < hi ? j + k : hi - 1;
                    ck[k] = (int)sorted[jj];
                    q[k] = sorted_xy[jj];
                }
#ifdef __HIP_DEVICE_COMPILE__
#pragma unroll
                for (int k = 0; k < OFL_RING_W; ++k) asm volatile("" : "+v"(ck[k]), "+v"(q[k].x), "+v"(q[k].y));
#endif
#ifdef __HIPCC__
#pragma unroll
#endif
                for (int k = 0; k < OFL_RING_W; ++k) c[k] = j + k < hi ? ck[k] : -1;
            } else {
                for (int k = 0; k < OFL_RING_W; ++k) {
                    c[k] = j + k < hi ? (int)sorted[j + k] : -1;
                    q[k] = c[k] >= 0 ? pos(c[k]) : pp;
                }
            }
#ifdef __HIPCC__
#pragma unroll
#endif
            for (int k = 0; k < OFL_RING_W; ++k) {
                if (c[k] < 0 || c[k] == p) continue;
                const P2 C = { q[k].x - pp.x, q[k].y - pp.y };
                const double d2 = C.x * C.x + C.y * C.y;
                if (d2 == 0.0 || d2 >= reach2) continue;      // a duplicate of p (same cell), or too far to matter
                DL_DBG(7, 1);
                const int rc = poly_clip(P, C, c[k], p, rel);
                if (rc < 0) return -1;
                if (rc > 0) reach2 = 4.0 * poly_rmax2(P);
            }
        }
        lo = nlo; hi = nhi;
    }
    return 0;
}

// The star of site p from the sites within `rings` bucket rings.  Returns 1 when the cell is final (every site within
// twice the farthest cell vertex has been applied: sites in unvisited buckets are at least r * s away), 0 when it is
// not (unbounded cells, rims of large holes: the far pass finishes those), -1 when the polygon overflowed.
// A cell that is still unbounded (an edge on the initial box) after `open_rings` rings is given up at once: hull points
// and rims of holes do not close within the ring search, and the cooperative passes pick them up with the sites found
// so far as seeds.
// `rescue`: called once on a cell that is still unbounded after `open_rings` rings, it may clip the cell with sites of its
// own choosing (the GPU pass: the site's grid neighbours -- across a tear of the mesh the neighbour on the other side is
// the site that closes the cell, twenty pixels away where the ring search reaches seven); returns < 0 on overflow.
struct NoRescue { template <class PolyX> DL_HD int operator()(PolyX &) const { return 0; } };

// `open_reach2`: a cell whose reach (twice its farthest vertex, squared) exceeds this counts as unbounded as well.  The rim
// of a straight tear or hole is a row of collinear sites: the bisectors of a rim site's two rim neighbours are parallel
// up to float32 rounding of the flow and meet a million pixels out -- a "closed" cell by its tags, which then dragged every
// candidate of all six rings through the vertex loop (87 000 vector instructions per wave on the rim of a 400 x 800 hole:
// 0.9 ms for 14 000 sites) before being handed on anyway.
// Returns 2 (HEAVY = false only) when the search met a run that may hold a heavy bucket: the polygon is abandoned, the caller
// hands the site to the HEAVY instantiation.
template <bool HEAVY = false, class PolyX, class PosFn, class RescueFn = NoRescue>
DL_HD int star_near(PolyX &P, int p, const P2 &pp, const Grid &g, const unsigned *bstart, const unsigned *sorted,
                    PosFn pos, int rings, const P2 *sorted_xy = nullptr, int open_rings = 1 << 30,
                    int *rings_done = nullptr,               // receives the last ring that was applied completely (-1: none)
                    RescueFn rescue = RescueFn(), double open_reach2 = 1e300,
                    const SubGrids *sub = nullptr)           // the grids of g's heavy buckets (dense clusters), or null
{
    poly_init(P);
    const int bx = g.bx(pp.x), by = g.by(pp.y);
    double reach2 = 4.0 * poly_rmax2(P);
    if (rings_done) *rings_done = -1;
    unsigned heavy_seen = 0;
    for (int r = 0; r <= rings; ++r) {
        const int ar = apply_ring<HEAVY>(P, p, pp, bx, by, r, g, bstart, sorted, pos, reach2, sorted_xy, sub, sub ? &heavy_seen : nullptr);
        if (ar < 0) return -1;
        if (ar == 2) return 2;
        if (ar > 0) return rescue(P) < 0 ? -1 : 0;         // the rim of a dense cluster: unfinished (ring r was not applied completely)
        if (rings_done) *rings_done = r;
        const double cover = (double)r * g.s;
        if (cover * cover >= reach2) return 1;
        if (r >= open_rings) {
            bool open = reach2 > open_reach2;
            for (int k = 0; k < P.n; ++k) open = open || P.T(k) < 0;
            if (open) return rescue(P) < 0 ? -1 : 0;      // (closed by the rescue or not: the later passes, which look at the
                                                          // sparse coarse grid before the dense fine rings, take it from here)
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ mesh fan
// Most sites of a warped grid sit in an intact neighbourhood: their eight grid neighbours are kept and the cell-wise
// mesh around them is locally the Delaunay triangulation.  For such a site the star need not be BUILT, only VERIFIED:
// the mesh proposes a fan (the four axis neighbours, plus the diagonal neighbour of every surrounding cell whose
// Delaunay diagonal passes through the site), and a fan of counter-clockwise triangles that winds around the site once
// and whose circumcircles hold no other site IS the site's Delaunay star (empty-circle property; the triangles of a
// triangulation around a vertex are unique).  Emptiness is tested against the sites of the buckets under the
// circumcircles -- sites of other sheets of a folded field are found there like any other -- with three fused
// multiply-adds per (site, triangle) and the in-circle predicate of the clip path only inside a rounding margin.
// Anything else (a missing neighbour, a site inside a circle, wide circles, a clockwise triangle) returns 0 and the
// site takes the clip path; so the fan never changes WHAT is computed, only how fast.
//
// Slots 0 .. 7 = grid neighbours E, SE, S, SW, W, NW, N, NE (x right, y down: counter-clockwise in the Poly convention).
// Cell c (0 .. 3) lies between the axis slots 2c and 2c + 2 and owns the diagonal slot 2c + 1.  Which diagonal splits
// a cell is decided from the cell's smallest-index corner a (corners a, b, c, d around the cell): a-c exists iff c is
// inside circle(a, b, d), ties to a-c -- every corner of the cell evaluates the same expression on the same numbers, so
// the four stars agree bit for bit, and ties fall the way the certified path and vertex_cut break them.
// index offset of grid-neighbour slot s without a table (dx + 1 / dy + 1 of the eight slots, two bits each)
DL_HD int slot_offset(int s, int W)
{
    const int dx = (int)((0x901Au >> (2 * s)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * s)) & 3u) - 1;
    return dx + dy * W;
}

constexpr double kFanEps  = 1e-13;       // relative margin of the float64 three-term in-circle sum (its rounding error is ~1e-15)
constexpr float  kFanEpsF = 2e-5f;       // ... and of its float32 version (inputs rounded to float32: ~1e-6)

// first set bit of an 8-bit ring read cyclically from position `from`: the next present slot at or after `from`
DL_HD int ring_next(unsigned mask8, int from)
{
    const unsigned r = ((mask8 | (mask8 << 8)) >> from) & 0xFFu;      // bit i = slot (from + i) & 7
    return (from + __builtin_ctz(r | 0x100u)) & 7;
}

// kept8: bit s = grid-neighbour slot s exists (inside the grid, kept by the point mask, not a dropped duplicate).  With
// all eight the proposal is the cell-wise mesh as described above; with some missing, the kept neighbours in their
// cyclic order (a diagonal neighbour of an incomplete cell is simply taken) -- a guess that the verification either
// confirms or sends to the clip path, so speckled point masks still settle most of their stars here.
template <class PosFn, class SlotPosFn>
DL_HD int star_fan(int p, int W, const P2 &pp, unsigned kept8, PosFn pos, SlotPosFn npos, const Grid &g, const unsigned *bstart,
                   const unsigned *sorted, const P2 *sorted_xy, int max_span,
                   P2 *nrel, int nstride,                     // scratch: relative positions of the eight slots
                   unsigned *nbr_out)                         // the star, counter-clockwise (3 .. 8 sites)
{
    const int off[8] = { 1, W + 1, W, W - 1, -1, -W - 1, -W, -W + 1 };
    P2 Q[8];
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int s = 0; s < 8; ++s) Q[s] = ((kept8 >> s) & 1u) ? npos(s) : pp;   // position of grid-neighbour slot s (= pos(p + off[s]))
    // diagonals: cell 0 (a = p, b = E, c = SE, d = S), cell 1 (a = W, b = p, c = S, d = SW),
    //            cell 2 (a = NW, b = N, c = p, d = W), cell 3 (a = N, b = NE, c = E, d = p)
    auto ac = [](const P2 &A, const P2 &B, const P2 &C, const P2 &D) {
        return incircle_origin_filtered(P2{ B.x - A.x, B.y - A.y }, P2{ D.x - A.x, D.y - A.y }, P2{ C.x - A.x, C.y - A.y }) >= 0.0;
    };
    bool whole[4];                                               // cell c has all its corners
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int c = 0; c < 4; ++c) whole[c] = ((kept8 >> (2 * c)) & (kept8 >> (2 * c + 1)) & (kept8 >> ((2 * c + 2) & 7)) & 1u) != 0;
    // A whole cell must be convex and positively oriented for what follows: its diagonal is chosen by ONE in-circle sign, and the
    // corner that diagonal does not lead to is taken on trust (`adj` below) -- in a cell with a reflex corner that corner can lie
    // INSIDE the triangle of the other three (an outlier vector next to a motion boundary: the soak of the composed paths found
    // one such node in 3.6 M), and the fan would be a triangle with a site in it.  Such a site is left to the clip pass, as
    // cell_verify leaves the cell.
    auto convex = [](const P2 &A, const P2 &B, const P2 &C, const P2 &D) {
        auto cr = [](const P2 &o, const P2 &u, const P2 &v) { return (u.x - o.x) * (v.y - o.y) - (u.y - o.y) * (v.x - o.x); };
        return cr(A, B, C) > 0.0 && cr(A, C, D) > 0.0 && cr(B, C, D) > 0.0 && cr(B, D, A) > 0.0;
    };
    if ((whole[0] && !convex(pp, Q[0], Q[1], Q[2])) || (whole[1] && !convex(Q[4], pp, Q[2], Q[3])) ||
        (whole[2] && !convex(Q[5], Q[6], pp, Q[4])) || (whole[3] && !convex(Q[6], Q[7], Q[0], pp))) return 0;
    unsigned pmask8 = kept8 & 0x55u;                             // present slots: the kept axis neighbours ...
    if (((kept8 >> 1) & 1u) && (!whole[0] || ac(pp, Q[0], Q[1], Q[2]))) pmask8 |= 2u;         // ... and the diagonal ones a whole
    if (((kept8 >> 3) & 1u) && (!whole[1] || !ac(Q[4], pp, Q[2], Q[3]))) pmask8 |= 8u;        //     cell's diagonal leads to
    if (((kept8 >> 5) & 1u) && (!whole[2] || ac(Q[5], Q[6], pp, Q[4]))) pmask8 |= 32u;        //     (any kept one of a broken cell)
    if (((kept8 >> 7) & 1u) && (!whole[3] || !ac(Q[6], Q[7], Q[0], pp))) pmask8 |= 128u;
    int npresent = 0;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int s = 0; s < 8; ++s) npresent += (pmask8 >> s) & 1u;
    if (npresent < 3) return 0;
    P2 N[8];
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int s = 0; s < 8; ++s) { N[s] = P2{ Q[s].x - pp.x, Q[s].y - pp.y }; nrel[s * nstride] = N[s]; }
    // triangle k (for a present slot k) = (N[k], N[next present slot]).
    // In-circle sum of a site C against triangle (0, A, B): |C|^2 o + C.x u + C.y w, negative inside; kept in float32
    // for the first look at every (site, triangle) pair.
    float o[8], u[8], w[8];
    float Mmax = 0.0f;
    unsigned long long adj = 0ull;                               // byte s: the triangles whose circle slot s is known to lie ON (or to be settled against)
    bool ok = true;
    int winds = 0;
    float bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int k = 0; k < 8; ++k) {
        const bool valid = (pmask8 >> k) & 1u;
        const int nk = ring_next(pmask8, (k + 1) & 7);
        const P2 A = N[k], B = nrel[nk * nstride];
        const double a2 = A.x * A.x + A.y * A.y, b2 = B.x * B.x + B.y * B.y;
        const double ok_ = A.x * B.y - A.y * B.x;
        const double uk = A.y * b2 - a2 * B.y, wk = a2 * B.x - A.x * b2;
        const double mo = fabs(A.x * B.y) + fabs(A.y * B.x), mu = fabs(A.y) * b2 + a2 * fabs(B.y), mw = a2 * fabs(B.x) + fabs(A.x) * b2;
        o[k] = valid ? (float)ok_ : 0.0f; u[k] = valid ? (float)uk : 0.0f; w[k] = valid ? (float)wk : 0.0f;
        if (valid) {
            adj |= (1ull << (8 * k + k)) | (1ull << (8 * nk + k));      // its own two sites
            // ... and the FOURTH corner of a whole cell the triangle lies in: which diagonal splits that cell has been
            // decided above, once and identically for all four of its corners (for a similarity every cell is co-circular
            // to rounding: re-deciding it from this site's point of view could only disagree with the other three)
            const int c = k >> 1;
            if (whole[c]) {
                if (!(k & 1)) adj |= 1ull << (8 * (nk == k + 1 ? ((k + 2) & 7) : k + 1) + k);
                else          adj |= 1ull << (8 * (k - 1) + k);
            }
            Mmax = fmaxf(Mmax, (float)fmax(mo, fmax(mu, mw)));
            if (!(ok_ > 0.0)) ok = false;                        // counter-clockwise triangles of less than a half turn only
            if (A.y < 0.0 && B.y >= 0.0) ++winds;
            // bounding box of the circumcircle, float32 with a safety margin (wide circles leave the fan path anyway)
            const float inv = 0.5f / (float)ok_;
            const float cx = -(float)uk * inv, cy = -(float)wk * inv;
            const float r = sqrtf(cx * cx + cy * cy);
            const float pad = r * 1.0001f + 1e-5f * (fabsf(cx) + fabsf(cy)) + 1e-30f;
            bx0 = fminf(bx0, cx - pad); bx1 = fmaxf(bx1, cx + pad);
            by0 = fminf(by0, cy - pad); by1 = fmaxf(by1, cy + pad);
        }
    }
    if (!ok || winds != 1) return 0;
    if (!(bx1 - bx0 < 3e30f) || !(by1 - by0 < 3e30f) || !(Mmax < 3e30f)) return 0;          // also catches NaN
    Mmax = Mmax * 1.000001f + 1e-37f;                            // an unused slot (o = u = w = 0) calls every site "outside"
    const int cb0 = g.bx(pp.x + (double)bx0), cb1 = g.bx(pp.x + (double)bx1);
    const int rb0 = g.by(pp.y + (double)by0), rb1 = g.by(pp.y + (double)by1);
    if (cb1 - cb0 >= max_span || rb1 - rb0 >= max_span) return 0;
    for (int row = rb0; row <= rb1; ++row) {
        const unsigned lo = bstart[(size_t)row * g.gx + cb0], hi = bstart[(size_t)row * g.gx + cb1 + 1];
        if (hi - lo > 4 * kHeavy) return 0;                      // a dense cluster under the circles: the clip pass (which searches such buckets by their own grids)
        for (unsigned j = lo; j < hi; ++j) {
            const P2 qa = sorted_xy ? sorted_xy[j] : pos((int)sorted[j]);
            const int qi = (int)sorted[j];
            const P2 C = { qa.x - pp.x, qa.y - pp.y };
            const float cxf = (float)C.x, cyf = (float)C.y, c2f = cxf * cxf + cyf * cyf;
            if (qi < 0 || qi == p || (C.x == 0.0 && C.y == 0.0)) continue;        // a blanked entry, the site itself, or a duplicate of it
            const float marg = kFanEpsF * (c2f + fabsf(cxf) + fabsf(cyf)) * Mmax;
            unsigned need = 0;
#ifdef __HIPCC__
#pragma unroll
#endif
            for (int k = 0; k < 8; ++k) {
                const float sum = c2f * o[k] + (cxf * u[k] + cyf * w[k]);       // < 0: inside
                if (!(sum > marg)) need |= 1u << k;
            }
            if (!need) continue;
            bool member = false;                                 // one of the eight grid neighbours, or a duplicate of one?
#ifdef __HIPCC__
#pragma unroll
#endif
            for (int s = 0; s < 8; ++s)
                if (qi == p + off[s]) { member = true; need &= ~(unsigned)((adj >> (8 * s)) & 0xFFull); }
            if (!need) continue;
            // inside some circle, or within the float32 margin of one: float64, then (within ITS margin) the predicate of
            // the clip path with ties as vertex_cut breaks them.  Rare: a plain loop with computed slot numbers.
            const double c2 = C.x * C.x + C.y * C.y;
#ifdef __HIPCC__
#pragma unroll 1
#endif
            for (int k = 0; k < 8; ++k) {
                if (!((need >> k) & 1u) || !((pmask8 >> k) & 1u)) continue;
                const int sa = k, sb = ring_next(pmask8, (k + 1) & 7);
                const int pa = p + slot_offset(sa, W), pb = p + slot_offset(sb, W);
                if (qi == pa || qi == pb) continue;              // the triangle's own sites lie ON its circle
                const P2 A = nrel[sa * nstride], B = nrel[sb * nstride];
                if ((A.x == C.x && A.y == C.y) || (B.x == C.x && B.y == C.y)) continue;   // a duplicate of a triangle's site
                const double a2 = A.x * A.x + A.y * A.y, b2 = B.x * B.x + B.y * B.y;
                const double t0 = c2 * (A.x * B.y - A.y * B.x), t1 = C.x * (A.y * b2 - a2 * B.y), t2 = C.y * (a2 * B.x - A.x * b2);
                const double sum = t0 + (t1 + t2);
                const double mag = c2 * (fabs(A.x * B.y) + fabs(A.y * B.x)) + fabs(C.x) * (fabs(A.y) * b2 + a2 * fabs(B.y))
                                 + fabs(C.y) * (a2 * fabs(B.x) + fabs(A.x) * b2);
                if (sum > kFanEps * mag) continue;               // outside
                if (sum < -kFanEps * mag) return 0;              // inside: not a Delaunay triangle
                const double ic = incircle_origin(A, B, C);
                if (ic > 0.0) return 0;
                if (ic == 0.0) {
                    if (!member) {
#ifdef __HIPCC__
#pragma unroll 1
#endif
                        for (int s = 0; s < 8; ++s) { const P2 T = nrel[s * nstride]; member = member || (((kept8 >> s) & 1u) && T.x == C.x && T.y == C.y); }
                    }
                    if (!member && (p < qi ? p : qi) < (pa < pb ? pa : pb)) return 0;
                }
            }
        }
    }
    int n = 0;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int s = 0; s < 8; ++s) if ((pmask8 >> s) & 1u) nbr_out[n++] = (unsigned)(p + off[s]);
    return n;
}

// ------------------------------------------------------------------------------------------------ mesh cells
// The fan pass above verifies every triangle of the warped mesh THREE times (once from each of its vertices) after a
// float64 set-up of eight triangles per site.  The same empty-circle property can be established once per CELL: a grid
// cell (corners a = (x, y), b = (x + 1, y), c = (x + 1, y + 1), d = (x, y + 1), all kept) that is convex and positively
// oriented is split along its Delaunay diagonal -- decided by the expression star_fan uses, so a site that has to fall
// back to its fan sees the same diagonal -- and its two triangles are tested against the sites of the buckets under
// their circumcircles (float32 first look, float64 inside its margin, the in-circle predicate inside that one's).  A site
// whose four surrounding cells are verified has its star for free: the axis neighbours plus the diagonal neighbour of
// every cell whose diagonal passes through it -- the angles of four convex cells at a common corner add up to one turn,
// and triangles with empty circumcircles around a vertex ARE its Delaunay star.  Ties (a site exactly ON a circle that
// is not a corner of the cell) and everything else unusual fail the cell; its corners then take the fan / clip path, so
// the cells -- like the fans -- never change WHAT is computed.
// Returns 0 (not verified), 1 (verified, diagonal a - c) or 2 (verified, diagonal b - d).
template <class PosFn>
DL_HD int cell_verify(int ia, int W, const P2 &A, const P2 &B, const P2 &C, const P2 &D, const Grid &g,
                      const unsigned *bstart, const unsigned *sorted, const P2 *sorted_xy, PosFn pos, int max_span)
{
    auto cross = [](const P2 &o, const P2 &u, const P2 &v) { return (u.x - o.x) * (v.y - o.y) - (u.y - o.y) * (v.x - o.x); };
    if (!(cross(A, B, C) > 0.0 && cross(A, C, D) > 0.0 && cross(B, C, D) > 0.0 && cross(B, D, A) > 0.0)) return 0;
    const bool ac = incircle_origin_filtered(P2{ B.x - A.x, B.y - A.y }, P2{ D.x - A.x, D.y - A.y }, P2{ C.x - A.x, C.y - A.y }) >= 0.0;
    const P2 O = ac ? A : B;                                      // the vertex both triangles share
    // triangle t = (O, U[t], V[t]), counter-clockwise: (a, b, c), (a, c, d)  or  (b, c, d), (b, d, a)
    const P2 q0 = ac ? B : C, q1 = ac ? C : D, q2 = ac ? D : A;
    const P2 U[2] = { P2{ q0.x - O.x, q0.y - O.y }, P2{ q1.x - O.x, q1.y - O.y } };
    const P2 V[2] = { P2{ q1.x - O.x, q1.y - O.y }, P2{ q2.x - O.x, q2.y - O.y } };
    float o[2], u[2], w[2];
    float Mmax = 0.0f, bx0 = 3e38f, bx1 = -3e38f, by0 = 3e38f, by1 = -3e38f;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int t = 0; t < 2; ++t) {
        const double a2 = U[t].x * U[t].x + U[t].y * U[t].y, b2 = V[t].x * V[t].x + V[t].y * V[t].y;
        const double ot = U[t].x * V[t].y - U[t].y * V[t].x;
        const double ut = U[t].y * b2 - a2 * V[t].y, wt = a2 * V[t].x - U[t].x * b2;
        const double mo = fabs(U[t].x * V[t].y) + fabs(U[t].y * V[t].x), mu = fabs(U[t].y) * b2 + a2 * fabs(V[t].y), mw = a2 * fabs(V[t].x) + fabs(U[t].x) * b2;
        if (!(ot > 0.0)) return 0;
        o[t] = (float)ot; u[t] = (float)ut; w[t] = (float)wt;
        Mmax = fmaxf(Mmax, (float)fmax(mo, fmax(mu, mw)));
        const float inv = 0.5f / (float)ot;
        const float cx = -(float)ut * inv, cy = -(float)wt * inv;
        const float r = sqrtf(cx * cx + cy * cy);
        const float pad = r * 1.0001f + 1e-5f * (fabsf(cx) + fabsf(cy)) + 1e-30f;
        bx0 = fminf(bx0, cx - pad); bx1 = fmaxf(bx1, cx + pad);
        by0 = fminf(by0, cy - pad); by1 = fmaxf(by1, cy + pad);
    }
    if (!(bx1 - bx0 < 3e30f) || !(by1 - by0 < 3e30f) || !(Mmax < 3e30f)) return 0;          // also catches NaN
    Mmax = Mmax * 1.000001f + 1e-37f;
    const int cb0 = g.bx(O.x + (double)bx0), cb1 = g.bx(O.x + (double)bx1);
    const int rb0 = g.by(O.y + (double)by0), rb1 = g.by(O.y + (double)by1);
    if (cb1 - cb0 >= max_span || rb1 - rb0 >= max_span) return 0;
    for (int row = rb0; row <= rb1; ++row) {
        const unsigned lo = bstart[(size_t)row * g.gx + cb0], hi = bstart[(size_t)row * g.gx + cb1 + 1];
        if (hi - lo > 4 * kHeavy) return 0;                          // a dense cluster under the circles: left to the clip pass
        for (unsigned j = lo; j < hi; ++j) {
            // (asking for a few candidates' positions together, as apply_ring does, changes nothing here: four cost a wave of occupancy, two +- 0)
            const P2 qa = sorted_xy ? sorted_xy[j] : pos((int)sorted[j]);
            const P2 Cq = { qa.x - O.x, qa.y - O.y };
            const float cxf = (float)Cq.x, cyf = (float)Cq.y, c2f = cxf * cxf + cyf * cyf;
            const float marg = kFanEpsF * (c2f + fabsf(cxf) + fabsf(cyf)) * Mmax;
            const bool n0 = !(c2f * o[0] + (cxf * u[0] + cyf * w[0]) > marg), n1 = !(c2f * o[1] + (cxf * u[1] + cyf * w[1]) > marg);
            if (!(n0 || n1)) continue;                               // well outside both circles (a blanked entry holds its old position: harmless)
            const int qi = (int)sorted[j];
            if (qi < 0) continue;                                    // a dropped duplicate
            if (qi == ia || qi == ia + 1 || qi == ia + W || qi == ia + W + 1) continue;      // the cell's own corners: vertices, or settled by the diagonal
            const double c2 = Cq.x * Cq.x + Cq.y * Cq.y;
#ifdef __HIPCC__
#pragma unroll
#endif
            for (int t = 0; t < 2; ++t) {                            // (unrolled: a run-time index would move U / V out of registers)
                if (!(t ? n1 : n0)) continue;
                const P2 Ut = U[t], Vt = V[t];
                const double a2 = Ut.x * Ut.x + Ut.y * Ut.y, b2 = Vt.x * Vt.x + Vt.y * Vt.y;
                const double t0 = c2 * (Ut.x * Vt.y - Ut.y * Vt.x), t1 = Cq.x * (Ut.y * b2 - a2 * Vt.y), t2 = Cq.y * (a2 * Vt.x - Ut.x * b2);
                const double sum = t0 + (t1 + t2);
                const double mag = c2 * (fabs(Ut.x * Vt.y) + fabs(Ut.y * Vt.x)) + fabs(Cq.x) * (fabs(Ut.y) * b2 + a2 * fabs(Vt.y))
                                 + fabs(Cq.y) * (a2 * fabs(Vt.x) + fabs(Ut.x) * b2);
                if (sum > kFanEps * mag) continue;                   // outside
                if (sum < -kFanEps * mag) return 0;                  // inside: not a Delaunay triangle
                if (!(incircle_origin(Ut, Vt, Cq) < 0.0)) return 0;  // inside, or exactly on the circle (a tie: left to the fan / clip path)
            }
        }
    }
    return ac ? 1 : 2;
}

// The star of an interior site from the flags of its four cells (cell_verify; f_se = the cell whose corner a is the site,
// f_sw / f_nw / f_ne = the cells to its lower left / upper left / upper right): slots E, SE, S, SW, W, NW, N, NE as in
// star_fan, the diagonal ones only where the cell's diagonal passes through the site.  Returns the number of neighbours
// (4 .. 8), or 0 when a cell is not verified.
DL_HD int star_from_cells(int p, int W, int f_se, int f_sw, int f_nw, int f_ne, unsigned *nbr_out)
{
    if (!(f_se && f_sw && f_nw && f_ne)) return 0;
    int n = 0;
    nbr_out[n++] = (unsigned)(p + 1);
    if (f_se == 1) nbr_out[n++] = (unsigned)(p + W + 1);            // cell (a = p, b = E, c = SE, d = S): diagonal a - c
    nbr_out[n++] = (unsigned)(p + W);
    if (f_sw == 2) nbr_out[n++] = (unsigned)(p + W - 1);            // cell (a = W, b = p, c = S, d = SW): diagonal b - d
    nbr_out[n++] = (unsigned)(p - 1);
    if (f_nw == 1) nbr_out[n++] = (unsigned)(p - W - 1);            // cell (a = NW, b = N, c = p, d = W): diagonal a - c
    nbr_out[n++] = (unsigned)(p - W);
    if (f_ne == 2) nbr_out[n++] = (unsigned)(p - W + 1);            // cell (a = N, b = NE, c = E, d = p): diagonal b - d
    return n;
}

// The polygon of a cell from the cyclic sequence of its edges' sites (box sides: -1 .. -4) -- what a pass that had to
// leave the cell unfinished hands on: vertex k is where the lines of tags[k - 1] and tags[k] meet.  One step instead of one
// clip per site.  Returns false (polygon untouched) when a vertex does not come out finite: the caller then clips the
// sites one by one.
template <class PolyX, class RelFn>
DL_HD bool poly_from_tags(PolyX &P, const unsigned *tags, int n, RelFn rel)
{
    if (n < 3 || n > P.cap) return false;
    double vx[16], vy[16];
    if (n > 16) return false;
    for (int k = 0; k < n; ++k) {
        double ax, ay, ah, bx, by, bh;
        edge_line((int)tags[k == 0 ? n - 1 : k - 1], rel, ax, ay, ah);
        edge_line((int)tags[k], rel, bx, by, bh);
        const double det = ax * by - bx * ay;
        if (det == 0.0) return false;
        vx[k] = (ah * by - bh * ay) / det; vy[k] = (ax * bh - bx * ah) / det;
        if (!(isfinite(vx[k]) && isfinite(vy[k])) || fabs(vx[k]) > 4.0 * kBox || fabs(vy[k]) > 4.0 * kBox) return false;
    }
    for (int k = 0; k < n; ++k) { P.X(k) = vx[k]; P.Y(k) = vy[k]; P.T(k) = (int)tags[k]; }
    P.n = n;
    return true;
}

// Second per-thread attempt at a star the ring search left unfinished (rims of tears and small holes: cells tens of
// bucket widths across, far too few vertices to be worth a workgroup).  The edges the cell had when the ring search
// stopped (`seeds`: their sites in cyclic order, box sides included) rebuild it; the fine rings the search did not reach are applied; then rings of the COARSE
// grid, which holds only the unfinished sites -- a Delaunay neighbour beyond the fine rings is itself unfinished (a
// finished site has all its neighbours within its own ring search, and a fan-verified one within kFanSpan buckets).
// Returns 1 when the cell is final and bounded, 0 when it is not within `rings1` coarse rings (or still unbounded), -1
// on polygon overflow.
template <class PolyX, class PosFn>
DL_HD int star_near2(PolyX &P, int p, const P2 &pp, const unsigned *seeds, int nseeds, int rings_done, int rings,
                     const Grid &g, const unsigned *bstart, const unsigned *sorted, const P2 *sorted_xy,
                     int rings1, const Grid &g1, const unsigned *b1start, const unsigned *sorted1, const P2 *sorted1_xy,
                     PosFn pos, int open_rings1 = 1 << 30, double open_reach2 = 1e300, const SubGrids *sub = nullptr)
{
    auto rel = [&](int t) { const P2 q = pos(t); return P2{ q.x - pp.x, q.y - pp.y }; };
    if (!poly_from_tags(P, seeds, nseeds, rel)) {
        poly_init(P);
        for (int i = 0; i < nseeds; ++i) {
            const int c = (int)seeds[i];
            if (c >= 0 && poly_clip(P, rel(c), c, p, rel) < 0) return -1;
        }
    }
    double reach2 = 4.0 * poly_rmax2(P);
    // the coarse rings first: they are sparse, and a cell they cannot close (the rim of a large hole) is given up before
    // the dense fine rings are touched; the order in which half-planes are applied does not change their intersection
    {
        const int bx = g1.bx(pp.x), by = g1.by(pp.y);
        const int a = bx > g1.gx - 1 - bx ? bx : g1.gx - 1 - bx, b = by > g1.gy - 1 - by ? by : g1.gy - 1 - by;
        const int rgrid = a > b ? a : b;                  // beyond this ring the coarse grid is exhausted
        bool done = false;
        for (int r = 0; r <= rings1 && !done; ++r) {
            if (apply_ring(P, p, pp, bx, by, r, g1, b1start, sorted1, pos, reach2, sorted1_xy) < 0) return -1;
            const double cover = (double)r * g1.s;
            done = cover * cover >= reach2 || r >= rgrid;
            if (!done && r >= open_rings1) {              // still unbounded this far out: a large hole, not a tear
                bool open = reach2 > open_reach2;         // (or "closed" a million pixels out by two nearly parallel bisectors)
                for (int k = 0; k < P.n; ++k) open = open || P.T(k) < 0;
                if (open) return 0;
            }
        }
        if (!done) return 0;
    }
    {
        const int bx = g.bx(pp.x), by = g.by(pp.y);
        for (int r = rings_done + 1; r <= rings; ++r) {
            const int ar = apply_ring(P, p, pp, bx, by, r, g, bstart, sorted, pos, reach2, sorted_xy, sub);
            if (ar < 0) return -1;
            if (ar != 0) return 0;                          // a dense cluster in the fine rings: left to the cooperative passes
        }
    }
    for (int k = 0; k < P.n; ++k) if (P.T(k) < 0) return 0;
    return 1;
}

// Rotation of a triangle's vertex indices that puts the smallest first without changing the cyclic order: every copy
// of a triangle (each of its three sites emits it) then interpolates with identical arithmetic.
DL_HD void canonical3(unsigned &i0, unsigned &i1, unsigned &i2)
{
    if (i1 < i0 && i1 < i2) { const unsigned t = i0; i0 = i1; i1 = i2; i2 = t; }
    else if (i2 < i0 && i2 < i1) { const unsigned t = i2; i2 = i1; i1 = i0; i0 = t; }
}

}  // namespace ofl_dl
