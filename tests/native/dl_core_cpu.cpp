// CPU build of the exact scatter path's geometry core (oflibnumpy_amd/csrc/ofl_delaunay_core.h) -- TEST infrastructure:
// tests/test_delaunay_core.py compiles this file with g++ and checks the stars it produces against SciPy's Delaunay
// triangulation on the reference fixtures.  The product never loads it (the GPU kernels include the same header).
#include "../../oflibnumpy_amd/csrc/ofl_delaunay_core.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

using namespace ofl_dl;

namespace {

// sequential clip for polygons of any size (the far pass; the GPU version of this step is workgroup-cooperative)
template <class RelFn>
int poly_clip_any(Poly &P, const P2 &C, int ctag, int ptag, RelFn rel, std::vector<char> &cut)
{
    const double h = 0.5 * (C.x * C.x + C.y * C.y);
    const int n = P.n;
    cut.assign(n, 0);
    int ncut = 0;
    for (int k = 0; k < n; ++k) { cut[k] = (char)vertex_cut_ex(P, k, n, C, ctag, ptag, h, rel); ncut += cut[k] != 0; }
    if (ncut == 0 || ncut == n) return 0;
    int a = -1, first = -1;                           // the first run with a vertex cut beyond the margin, else the first run (as far_apply)
    for (int k = 0; k < n && a < 0; ++k)
        if (cut[k] && !cut[k == 0 ? n - 1 : k - 1]) {
            if (first < 0) first = k;
            for (int m = k, c = 0; c < n && cut[m]; m = (m + 1) % n, ++c) if (cut[m] == 2) a = k;
        }
    if (a < 0) a = first;
    int L = 0;
    while (cut[(a + L) % n]) ++L;
    const int b = (a + L - 1) % n, ia = a == 0 ? n - 1 : a - 1, ib = (b + 1) % n;
    if (n - L + 2 > P.cap) return -1;
    const int tb = P.T(b);
    const P2 v1 = cut_point(P.T(ia), C, h, rel, P.X(ia), P.Y(ia), P.X(a), P.Y(a));
    const P2 v2 = cut_point(tb, C, h, rel, P.X(b), P.Y(b), P.X(ib), P.Y(ib));
    std::vector<double> nx, ny; std::vector<int> nt;
    for (int j = 0; j < n - L; ++j) { const int k = (b + 1 + j) % n; nx.push_back(P.X(k)); ny.push_back(P.Y(k)); nt.push_back(P.T(k)); }
    nx.push_back(v1.x); ny.push_back(v1.y); nt.push_back(ctag);
    nx.push_back(v2.x); ny.push_back(v2.y); nt.push_back(tb);
    for (size_t k = 0; k < nx.size(); ++k) { P.X((int)k) = nx[k]; P.Y((int)k) = ny[k]; P.T((int)k) = nt[k]; }
    P.n = (int)nx.size();
    return 1;
}

}  // namespace

// pts [n][2]; tri_out receives (p, a, b) for every pair of consecutive real neighbours of every star ("emit all");
// info[8]: [0] far sites, [1] polygon overflows in the near pass, [2] grid gx, [3] grid gy, [4] largest star, [5] sites settled by
// mesh fans, [6] by the second per-thread pass, [7] by the mesh-cell pass
// kept (or null: all) marks the sites that exist; with W > 0 the sites are the points of a warped H x W grid in row-major
// order and intact neighbourhoods take the mesh-fan shortcut first (info[5] counts them), as the GPU path does
static int stars_impl(const double *pts, int n, const unsigned char *kept, int W, int rings, int near_cap, int *tri_out,
                      long long tri_cap, long long *n_tri, int *info)
{
    if (n <= 0) return -1;
    auto is_kept = [&](int i) { return !kept || kept[i]; };
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    int nk = 0;
    for (int i = 0; i < n; ++i) {
        if (!is_kept(i)) continue;
        ++nk;
        x0 = std::min(x0, pts[2 * i]); x1 = std::max(x1, pts[2 * i]);
        y0 = std::min(y0, pts[2 * i + 1]); y1 = std::max(y1, pts[2 * i + 1]);
    }
    if (nk == 0) return -1;
    Grid g;
    const double bw = x1 - x0, bh = y1 - y0;
    double s = sqrt(std::max(bw * bh, 1e-300) / nk);
    s = std::max(s, (bw + bh) / (double)nk);
    if (!(s > 0)) s = 1.0;
    g.ox = x0; g.oy = y0; g.s = s; g.inv_s = 1.0 / s;
    g.gx = (int)floor(bw / s) + 1; g.gy = (int)floor(bh / s) + 1;
    const size_t nb = (size_t)g.gx * g.gy;
    std::vector<unsigned> bstart(nb + 1, 0), sorted(nk), cursor(nb, 0);
    auto bucket = [&](int i) { return (size_t)g.by(pts[2 * i + 1]) * g.gx + g.bx(pts[2 * i]); };
    for (int i = 0; i < n; ++i) if (is_kept(i)) ++bstart[bucket(i) + 1];
    for (size_t b = 0; b < nb; ++b) bstart[b + 1] += bstart[b];
    for (int i = 0; i < n; ++i) if (is_kept(i)) { const size_t b = bucket(i); sorted[bstart[b] + cursor[b]++] = (unsigned)i; }   // ascending index per bucket
    auto pos = [&](int i) { return P2{ pts[2 * i], pts[2 * i + 1] }; };
    std::vector<double> vx(4096), vy(4096);
    std::vector<int> tag(4096);
    std::vector<int> far;
    std::vector<char> flags;
    long long nt = 0;
    int overflow = 0, maxdeg = 0;
    auto emit = [&](int p, const Poly &P) {
        maxdeg = std::max(maxdeg, P.n);
        for (int k = 0; k < P.n; ++k) {
            const int a = P.T(k), b = P.T((k + 1) % P.n);
            if (a < 0 || b < 0 || a == b) continue;
            if (nt < tri_cap) { tri_out[3 * nt] = p; tri_out[3 * nt + 1] = a; tri_out[3 * nt + 2] = b; }
            ++nt;
        }
    };
    std::vector<float> fx(64), fy(64);
    std::vector<std::vector<unsigned>> seed_of(n);
    std::vector<int> rd_of(n, -1);
    int fans = 0, near2 = 0, by_cells = 0;
    const int H = W > 0 ? n / W : 0;
    // the GPU's mesh-cell pass: every intact cell verified once (cell_verify), sites with four verified cells settled from the flags
    std::vector<unsigned char> cellflag(W > 0 ? n : 0, 0);
    if (W > 0)
        for (int y = 0; y + 1 < H; ++y)
            for (int x = 0; x + 1 < W; ++x) {
                const int ia = y * W + x;
                if (!(is_kept(ia) && is_kept(ia + 1) && is_kept(ia + W) && is_kept(ia + W + 1))) continue;
                cellflag[ia] = (unsigned char)cell_verify(ia, W, pos(ia), pos(ia + 1), pos(ia + W + 1), pos(ia + W), g, bstart.data(), sorted.data(),
                                                          (const P2 *)nullptr, pos, 6);
            }
    for (int p = 0; p < n; ++p) {
        if (!is_kept(p)) continue;
        if (W > 0) {
            const int xs = p % W, ys = p / W;
            if (xs > 0 && ys > 0 && xs < W - 1 && ys < H - 1) {
                unsigned nb8[8];
                const int m = star_from_cells(p, W, cellflag[p], cellflag[p - 1], cellflag[p - W - 1], cellflag[p - W], nb8);
                if (m > 0) {
                    for (int k = 0; k < m; ++k) tag[k] = (int)nb8[k];
                    Poly Q{ vx.data(), vy.data(), tag.data(), 1, 64, m };
                    emit(p, Q);
                    ++by_cells;
                    continue;
                }
            }
        }
        if (W > 0) {
            const int x = p % W, y = p / W;
            unsigned kept8 = 0;
            for (int sl = 0; sl < 8; ++sl) {
                const int dx = (int)((0x901Au >> (2 * sl)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * sl)) & 3u) - 1;
                if (x + dx >= 0 && x + dx < W && y + dy >= 0 && y + dy < H && is_kept(p + dy * W + dx)) kept8 |= 1u << sl;
            }
            {
                P2 nrel[8];
                unsigned nb8[8];
                const int m = star_fan(p, W, pos(p), kept8, pos, [&](int sl) { return pos(p + slot_offset(sl, W)); }, g, bstart.data(),
                                       sorted.data(), (const P2 *)nullptr, 6, nrel, 1, nb8);
                if (m > 0) {
                    for (int k = 0; k < m; ++k) tag[k] = (int)nb8[k];
                    Poly Q{ vx.data(), vy.data(), tag.data(), 1, 64, m };
                    emit(p, Q);
                    ++fans;
                    continue;
                }
            }
        }
        // near_cap < 0: the float32 cell of the GPU's per-thread pass
        if (near_cap < 0) {
            PolyT<float> P{ fx.data(), fy.data(), tag.data(), 1, -near_cap, 0 };
            int rdone = -1;
            // as the GPU's clip pass: a cell still unbounded at the open-ring check is clipped with the site's grid neighbours
            auto rescue = [&](PolyT<float> &Q) -> int {
                if (W <= 0) return 0;
                const P2 pp = pos(p);
                auto rel = [&](int t) { const P2 v = pos(t); return P2{ v.x - pp.x, v.y - pp.y }; };
                const int x = p % W, y = p / W;
                for (int sl = 0; sl < 8; ++sl) {
                    const int dx = (int)((0x901Au >> (2 * sl)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * sl)) & 3u) - 1;
                    if (x + dx < 0 || x + dx >= W || y + dy < 0 || y + dy >= H) continue;
                    const int q = p + dy * W + dx;
                    if (!is_kept(q)) continue;
                    const P2 C = rel(q);
                    if (C.x == 0.0 && C.y == 0.0) continue;
                    if (poly_clip(Q, C, q, p, rel) < 0) return -1;
                }
                return 0;
            };
            const int rc = star_near(P, p, pos(p), g, bstart.data(), sorted.data(), pos, rings, (const P2 *)nullptr, 2, &rdone, rescue);
            bool ok = rc == 1;
            for (int k = 0; ok && k < P.n; ++k) ok = P.T(k) >= 0;
            if (ok) {
                Poly Q{ vx.data(), vy.data(), tag.data(), 1, 64, P.n };
                emit(p, Q);
                continue;
            }
            if (rc < 0) ++overflow;
            for (int k = 0; k < P.n; ++k) seed_of[p].push_back((unsigned)P.T(k));      // every edge, box sides included: the cell is rebuilt from them
            rd_of[p] = rdone;
            far.push_back(p);
            continue;
        }
        Poly P{ vx.data(), vy.data(), tag.data(), 1, near_cap, 0 };
        const int rc = star_near(P, p, pos(p), g, bstart.data(), sorted.data(), pos, rings);
        if (rc == 1) { emit(p, P); continue; }
        if (rc < 0) ++overflow;
        far.push_back(p);
    }
    // the GPU's second per-thread pass: coarse grid (8 fine buckets per cell) of the unfinished sites
    std::vector<char> finished(n, 0);
    if (near_cap < 0 && !far.empty()) {
        Grid g1 = g;
        g1.s = g.s * 8; g1.inv_s = 1.0 / g1.s; g1.gx = (g.gx + 7) / 8; g1.gy = (g.gy + 7) / 8;
        const size_t nb1 = (size_t)g1.gx * g1.gy;
        std::vector<unsigned> b1(nb1 + 1, 0), s1(far.size()), c1(nb1, 0);
        auto bucket1 = [&](int i) { return (size_t)g1.by(pts[2 * i + 1]) * g1.gx + g1.bx(pts[2 * i]); };
        for (int i : far) ++b1[bucket1(i) + 1];
        for (size_t b = 0; b < nb1; ++b) b1[b + 1] += b1[b];
        for (int i : far) { const size_t b = bucket1(i); s1[b1[b] + c1[b]++] = (unsigned)i; }
        for (int p : far) {
            if (W > 0) { const int x = p % W, y = p / W; if (x == 0 || y == 0 || x == W - 1 || y == H - 1) continue; }
            PolyT<float> P{ fx.data(), fy.data(), tag.data(), 1, 16, 0 };
            const int rc = star_near2(P, p, pos(p), seed_of[p].data(), (int)seed_of[p].size(), rd_of[p], rings, g, bstart.data(),
                                      sorted.data(), (const P2 *)nullptr, 6, g1, b1.data(), s1.data(), (const P2 *)nullptr, pos);
            if (rc != 1) continue;
            Poly Q{ vx.data(), vy.data(), tag.data(), 1, 64, P.n };
            emit(p, Q);
            finished[p] = 1;
            ++near2;
        }
    }
    for (int p : far) {
        if (finished[p]) continue;
        Poly P{ vx.data(), vy.data(), tag.data(), 1, 4096, 0 };
        poly_init(P);
        const P2 pp = pos(p);
        auto rel = [&](int t) { const P2 q = pos(t); return P2{ q.x - pp.x, q.y - pp.y }; };
        // the sites of the near rings first (nearest buckets first shrink the cell quickly), then every far site
        const int bx = g.bx(pp.x), by = g.by(pp.y);
        for (int r = 0; r <= rings; ++r)
            for (int row = by - r; row <= by + r; ++row) {
                if (row < 0 || row >= g.gy) continue;
                for (int col = bx - r; col <= bx + r; ++col) {
                    if (col < 0 || col >= g.gx) continue;
                    if (std::max(abs(row - by), abs(col - bx)) != r) continue;
                    for (unsigned j = bstart[(size_t)row * g.gx + col]; j < bstart[(size_t)row * g.gx + col + 1]; ++j) {
                        const int c = (int)sorted[j];
                        const P2 C = rel(c);
                        if (c == p || (C.x == 0.0 && C.y == 0.0)) continue;
                        if (poly_clip_any(P, C, c, p, rel, flags) < 0) return -2;
                    }
                }
            }
        for (int c : far) {
            const P2 C = rel(c);
            if (c == p || (C.x == 0.0 && C.y == 0.0)) continue;
            if (poly_clip_any(P, C, c, p, rel, flags) < 0) return -2;
        }
        emit(p, P);
    }
    *n_tri = nt;
    info[0] = (int)far.size(); info[1] = overflow; info[2] = g.gx; info[3] = g.gy; info[4] = maxdeg; info[5] = fans; info[6] = near2; info[7] = by_cells;
    return 0;
}

extern "C" int dl_stars_cpu(const double *pts, int n, int rings, int near_cap, int *tri_out, long long tri_cap,
                            long long *n_tri, int *info)
{
    return stars_impl(pts, n, nullptr, 0, rings, near_cap, tri_out, tri_cap, n_tri, info);
}

// the sites are the kept points of a warped H x W grid (pts holds all H * W positions): mesh fans first, as on the GPU
extern "C" int dl_stars_grid_cpu(const double *pts, const unsigned char *kept, int H, int W, int rings, int near_cap,
                                 int *tri_out, long long tri_cap, long long *n_tri, int *info)
{
    return stars_impl(pts, H * W, kept, W, rings, near_cap, tri_out, tri_cap, n_tri, info);
}

// debugging aid: the near-pass star of ONE site with float32 or float64 cell storage, every clip printed
extern "C" int dl_star_trace(const double *pts, int n, int p, int use_float, int rings)
{
    double x0 = pts[0], x1 = pts[0], y0 = pts[1], y1 = pts[1];
    for (int i = 1; i < n; ++i) {
        x0 = std::min(x0, pts[2 * i]); x1 = std::max(x1, pts[2 * i]);
        y0 = std::min(y0, pts[2 * i + 1]); y1 = std::max(y1, pts[2 * i + 1]);
    }
    Grid g;
    const double bw = x1 - x0, bh = y1 - y0;
    double s = std::max(sqrt(std::max(bw * bh, 1e-300) / n), (bw + bh) / (double)n);
    g.ox = x0; g.oy = y0; g.s = s; g.inv_s = 1.0 / s;
    g.gx = (int)floor(bw / s) + 1; g.gy = (int)floor(bh / s) + 1;
    const size_t nb = (size_t)g.gx * g.gy;
    std::vector<unsigned> bstart(nb + 1, 0), sorted(n), cursor(nb, 0);
    auto bucket = [&](int i) { return (size_t)g.by(pts[2 * i + 1]) * g.gx + g.bx(pts[2 * i]); };
    for (int i = 0; i < n; ++i) ++bstart[bucket(i) + 1];
    for (size_t b = 0; b < nb; ++b) bstart[b + 1] += bstart[b];
    for (int i = 0; i < n; ++i) { const size_t b = bucket(i); sorted[bstart[b] + cursor[b]++] = (unsigned)i; }
    auto pos = [&](int i) { return P2{ pts[2 * i], pts[2 * i + 1] }; };
    const P2 pp = pos(p);
    auto rel = [&](int t) { const P2 q = pos(t); return P2{ q.x - pp.x, q.y - pp.y }; };
    std::vector<double> vx(64), vy(64); std::vector<float> fx(64), fy(64); std::vector<int> tag(64);
    Poly D{ vx.data(), vy.data(), tag.data(), 1, 64, 0 };
    PolyT<float> F{ fx.data(), fy.data(), tag.data(), 1, 64, 0 };
    if (use_float) poly_init(F); else poly_init(D);
    const int bx = g.bx(pp.x), by = g.by(pp.y);
    for (int r = 0; r <= rings; ++r) {
        for (int row = by - r; row <= by + r; ++row) for (int col = bx - r; col <= bx + r; ++col) {
            if (row < 0 || row >= g.gy || col < 0 || col >= g.gx || std::max(abs(row - by), abs(col - bx)) != r) continue;
            for (unsigned j = bstart[(size_t)row * g.gx + col]; j < bstart[(size_t)row * g.gx + col + 1]; ++j) {
                const int c = (int)sorted[j];
                const P2 C = rel(c);
                if (c == p || (C.x == 0 && C.y == 0)) continue;
                const int rc = use_float ? poly_clip(F, C, c, p, rel) : poly_clip(D, C, c, p, rel);
                if (rc) {
                    const int m = use_float ? F.n : D.n;
                    printf("r=%d clip by %d (%.4f,%.4f) rc=%d n=%d:", r, c, C.x, C.y, rc, m);
                    for (int k = 0; k < m; ++k) printf(" [%d](%.5g,%.5g)", tag[k], use_float ? (double)F.X(k) : D.X(k), use_float ? (double)F.Y(k) : D.Y(k));
                    printf("\n");
                }
            }
        }
    }
    return 0;
}
