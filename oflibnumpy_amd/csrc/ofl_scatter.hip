// ofl_scatter.hip -- K3: scattered -> regular-grid linear interpolation for gfx950.
//
// Replaces scipy.interpolate.griddata(points, values, grid, 'linear') as the reference uses it
// (src/oflibnumpy/utils.py:237-258; flow_class.py:1398-1410): the scattered points are the regular
// grid displaced by a flow, so their connectivity is known -- every source cell (x, y)-(x+1, y+1)
// becomes two triangles, split along the diagonal the Delaunay criterion picks for that cell
// (in-circle test in float64 on exactly the positions SciPy sees).  For flows whose cells stay close
// to similar (rotation / uniform scaling / translation + small deformation) this IS the Delaunay
// triangulation SciPy builds, without Qhull's O(N log N) serial construction.
//
//   pass 1 (raster)   one thread per source cell: both triangles are scan-converted over the integer
//                     nodes of their bounding boxes; a node inside (barycentric >= -eps, eps as in
//                     SciPy's _barycentric_inside) records the triangle id with atomicMin -> a
//                     deterministic owner per output node, no value races.  Triangles whose bounding
//                     box is large go to a list that pass 1b scans with one wave per triangle.
//   pass 2 (resolve)  one thread per output node (or per query point): barycentric coordinates in
//                     float64 from the owner triangle, values and mask interpolated, 0 where no
//                     triangle covers the node (NaN -> 0 of utils.py:254).
//
// Memory traffic: flow 8 B + point mask 1 B per cell corner (neighbouring cells share lines), one
// 4-byte atomic per covered node, then 4 + 3 * (8 + 4 C) gathered bytes and 4 C + 1 written bytes per
// node -- all L2-friendly because owner triangles of neighbouring nodes are neighbours in memory.
#include "ofl_common.h"
#include "ofl_scatter_dev.h"
#include <algorithm>
#include <chrono>
#include <mutex>
#include <stdlib.h>
#include <vector>
#include <type_traits>

using namespace ofl;
using namespace ofl_sc;

namespace {

constexpr uint32_t kNoOwner   = 0xFFFFFFFFu;
constexpr uint32_t kGapOwner  = 0xFFFFFFFEu;                     // owner-map marker: uncovered node inside the hull (pass 2b)
constexpr uint32_t kGapFar    = 0xFFFFFFFDu;                     // ... whose 5 x 5 neighbourhood is uncovered too (pass 2c)
constexpr uint32_t kGapDone   = 0xFFFFFFFCu;                     // ... that a gap pass has filled (not a seed for other gap nodes)
constexpr int      kSmallArea = 1024;                             // bbox nodes scanned inside the raster kernel (lane or wave)
#ifndef OFL_SC_COOP
#define OFL_SC_COOP 1
#endif
constexpr int      kCoopMinArea = 64;                             // smallest bbox a whole wave scans together
// the big-triangle list holds every triangle of the mesh if need be (a field that magnifies 30x makes ALL of them big)
inline long long big_cap_for(int H, int W) { return 2ll * H * W; }      // >= all triangles; doubles as two node-sized maps (deep gap fill)
constexpr int      kCandCap   = 1 << 22;                          // hull candidates kept on the device
constexpr int      kHullCap   = 1 << 16;                          // vertices per hull chain
constexpr int      kFillRadius = 16;                              // how far an uncovered node looks for a covered one
constexpr double   kHullTol   = 1e-12;                            // distance (px) a node may lie outside a hull edge


struct ScatterWs {          // layout of the caller-provided workspace
    uint32_t *owner;        // [H][W]
    uint32_t *big;          // [big_cap] triangle ids
    unsigned long long big_cap;
    D2       *cand;         // [cand_cap] positions of mesh-boundary points (convex-hull candidates)
    D2       *lower;        // [kHullCap] lower hull chain, x ascending
    D2       *upper;        // [kHullCap] upper hull chain, x ascending
    unsigned long long *counters;   // [1] big-list length, [2] big-list work, [3] candidates, [5] gap nodes beyond the ring search, [6..7] their bounding box
    unsigned long long *kept_slots; // [256] partial counts of kept points
    const D2 *guard;        // [4] warped kept points next to the four image corners (header bytes 64..127) ...
    const int *guard_ok;    // ... and whether all four exist (header byte 128)
    int       cand_cap;
    uint8_t  *coarse;       // [ceil(H/8)][ceil(W/32)] "some node of this 32 x 8 block has an owner" (1 = yes or unknown)
    int       coarse_w;
    int       oy0, oy1;     // rows [oy0, oy1) the owner map covers (0, H unless one row band is computed);
                            // `owner` is biased so that owner[y * W + x] addresses row y for y in that range
};

struct HullRef { const D2 *lower, *upper; int n_lower, n_upper; };

// value of `v` in lane `src` (wave-uniform index)
__device__ __forceinline__ double lane_bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// Appends the points of the lanes that `want` to the hull-candidate list: ONE atomicAdd per wave (a counter
// word bumped once per point serialises at ~12 ns per atomic).  Must be called by all lanes of the wave.
__device__ __forceinline__ void push_candidate(const ScatterWs &ws, bool want, const D2 &p)
{
    const unsigned long long m = __ballot(want);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)m) - 1);
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(&ws.counters[3], (unsigned long long)__popcll(m));
    const int lo = __builtin_amdgcn_readlane((int)(base & 0xffffffffull), leader), hi = __builtin_amdgcn_readlane((int)(base >> 32), leader);
    base = ((unsigned long long)(uint32_t)hi << 32) | (uint32_t)lo;
    if (want) {
        const unsigned long long slot = base + (unsigned long long)__popcll(m & ((1ull << lane) - 1ull));
        if (slot < (unsigned long long)ws.cand_cap) ws.cand[slot] = p;
    }
}

// p strictly inside the (non-degenerate) triangle a b c, with a relative safety margin
__device__ __forceinline__ bool strictly_inside(const D2 &a, const D2 &b, const D2 &c, const D2 &p)
{
    const double det = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
    if (det == 0.0) return false;
    double w1 = (p.x - a.x) * (c.y - a.y) - (p.y - a.y) * (c.x - a.x);
    double w2 = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
    double w0 = det - w1 - w2;
    if (det < 0) { w0 = -w0; w1 = -w1; w2 = -w2; }
    const double margin = 1e-9 * fabs(det);
    return w0 > margin && w1 > margin && w2 > margin;
}

// p inside or on the boundary of the (non-degenerate) triangle a b c without being one of its vertices
__device__ __forceinline__ bool inside_or_on(const D2 &a, const D2 &b, const D2 &c, const D2 &p)
{
    if ((p.x == a.x && p.y == a.y) || (p.x == b.x && p.y == b.y) || (p.x == c.x && p.y == c.y)) return false;
    const double det = (b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x);
    if (det == 0.0) return false;
    double w1 = (p.x - a.x) * (c.y - a.y) - (p.y - a.y) * (c.x - a.x);
    double w2 = (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
    double w0 = det - w1 - w2;
    if (det < 0) { w0 = -w0; w1 = -w1; w2 = -w2; }
    return w0 >= 0.0 && w1 >= 0.0 && w2 >= 0.0;
}

// bounding box restricted to the rows of the owner map
__device__ __forceinline__ TriBox tri_box_rows(const D2 &p0, const D2 &p1, const D2 &p2, int W, int H, const ScatterWs &ws)
{
    TriBox b = tri_box(p0, p1, p2, W, H);
    b.y0 = max(b.y0, ws.oy0);
    b.y1 = min(b.y1, ws.oy1 - 1);
    return b;
}

__device__ __forceinline__ uint32_t tri_id(uint32_t cell, int diag, int t) { return (cell << 2) | ((uint32_t)diag << 1) | (uint32_t)t; }

// decode a triangle id into its three source vertices (linear pixel indices) and positions
__device__ __forceinline__ void tri_decode(uint32_t id, const float *flow, int sign, int W,
                                           size_t (&vi)[3], D2 (&vp)[3])
{
    const int cw = W - 1;
    const uint32_t cell = id >> 2;
    const int diag = (id >> 1) & 1, t = id & 1;
    const int y = (int)(cell / (uint32_t)cw), x = (int)(cell - (uint32_t)y * (uint32_t)cw);
    int i0, i1, i2;
    tri_corners(diag, t, i0, i1, i2);
    const int ci[3] = { i0, i1, i2 };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int px = x + (((ci[k] + 1) >> 1) & 1), py = y + (ci[k] >> 1);     // corner a,b,c,d -> (0,0),(1,0),(1,1),(0,1)
        vi[k] = (size_t)py * W + px;
        vp[k] = point_of(flow, sign, W, px, py);
    }
}

// Convex-hull candidates among the kept points: the points on the border of the kept mesh (image border, or
// next to a dropped point); corners of folded cells are added by the raster pass further down.  Every vertex of the
// convex hull of the kept points is among them.  Must be called by all lanes of the wave.
__device__ __forceinline__ void hull_candidate(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask,
                                               int H, int W, int x, int y, const ScatterWs &ws)
{
    {
        bool cand = false, on_border = false;
        if (x < W && y < H && (!pmask || pmask[(size_t)y * W + x])) {
            cand = on_border = (x == 0 || y == 0 || x == W - 1 || y == H - 1);
            if (!cand && pmask) {
                for (int dy = -1; dy <= 1 && !cand; ++dy)
                    for (int dx = -1; dx <= 1 && !cand; ++dx)
                        cand = !pmask[(size_t)(y + dy) * W + (x + dx)];
            }
        }
        D2 p = { 0.0, 0.0 };
        if (cand) {
            // A point inside or on the boundary of a triangle of OTHER kept points is not an extreme point of
            // the set, so it cannot be a hull vertex.  Image-border points are tested against the triangles
            // (P[j-d], P[j+d], inward neighbour) along their border for d = 1, 2, 4, 8, 16: straight or inward-
            // bulging borders (every smooth warp) shed all but a few points here instead of on the host.
            // Border points of mask holes are tested against the two triangles of the warped image corners.
            p = point_of(flow, sign, W, x, y);
            const bool corner = (x == 0 || x == W - 1) && (y == 0 || y == H - 1);
            auto kept = [&](int xx, int yy) { return !pmask || pmask[(size_t)yy * W + xx] != 0; };
#ifndef OFL_SC_FILTER
#define OFL_SC_FILTER 1
#endif
            if (OFL_SC_FILTER && on_border && !corner && W > 2 && H > 2) {
                const bool horiz = (y == 0 || y == H - 1);           // border runs along x
                const int qx = horiz ? x : (x == 0 ? 1 : W - 2), qy = horiz ? (y == 0 ? 1 : H - 2) : y;
                if (kept(qx, qy)) {
                    const D2 q = point_of(flow, sign, W, qx, qy);
                    const int pos = horiz ? x : y, len = horiz ? W : H;
                    for (int d = 1; d <= 16 && cand; d <<= 1) {
                        if (pos - d < 0 || pos + d >= len) break;
                        const int ax = horiz ? x - d : x, ay = horiz ? y : y - d, bx = horiz ? x + d : x, by = horiz ? y : y + d;
                        if (!kept(ax, ay) || !kept(bx, by)) continue;
                        if (inside_or_on(point_of(flow, sign, W, ax, ay), point_of(flow, sign, W, bx, by), q, p)) cand = false;
                    }
                }
            } else if (!on_border) {
                if (*ws.guard_ok) {
                    const D2 ga = ws.guard[0], gb = ws.guard[1], gc = ws.guard[2], gd = ws.guard[3];
                    if (strictly_inside(ga, gb, gc, p) || strictly_inside(ga, gc, gd, p)) cand = false;
                }
            }
        }
        push_candidate(ws, cand, p);
    }
}

// Pass 0 (one workgroup): the four guard points of the candidate filters -- for every image corner the kept point
// closest to it (Chebyshev distance, then row-major order) inside the 16 x 16 block at that corner.  With a random
// point mask an image corner itself is often dropped; its neighbour serves just as well.
__global__ __launch_bounds__(256)
void scatter_guard_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                          D2 *__restrict__ guard, int *__restrict__ guard_ok)
{
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;            // corner 0..3 = (0,0) (W-1,0) (W-1,H-1) (0,H-1)
    const bool right = (k == 1 || k == 2), bottom = (k >= 2);
    uint32_t best = 0xffffffffu;
    for (int j = lane; j < 256; j += 64) {
        const int dx = j & 15, dy = j >> 4;
        if (dx >= W || dy >= H) continue;
        const int x = right ? W - 1 - dx : dx, y = bottom ? H - 1 - dy : dy;
        if (pmask && !pmask[(size_t)y * W + x]) continue;
        best = min(best, ((uint32_t)max(dx, dy) << 16) | (uint32_t)j);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, off));
    if (lane == 0) {
        if (best != 0xffffffffu) {
            const int j = (int)(best & 0xffffu), dx = j & 15, dy = j >> 4;
            guard[k] = point_of(flow, sign, W, right ? W - 1 - dx : dx, bottom ? H - 1 - dy : dy);
            atomicAdd(guard_ok, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
        *guard_ok = (__hip_atomic_load(guard_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 4) ? 1 : 0;
}

__global__ __launch_bounds__(256)
void scatter_raster_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask,
                           int H, int W, ScatterWs ws)
{
    const int cw = W - 1, ch = H - 1;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (pmask) {
        // Count the points SciPy would receive (utils.py:249-251).  One atomic per WORKGROUP, spread over
        // 256 slots: a single counter word would serialise ~1e5 atomics at ~12 ns each (1.5 ms at 4K --
        // measured, it dominated the first version of this kernel).
        const int kept = (x < W && y < H) ? (pmask[(size_t)y * W + x] != 0) : 0;
        const int block_sum = __syncthreads_count(kept);
        if (threadIdx.x == 0 && block_sum)
            atomicAdd(&ws.kept_slots[(blockIdx.y * gridDim.x + blockIdx.x) & 255], (unsigned long long)block_sum);
    }
    // hull candidates exist only in workgroups on the image border (or anywhere with a point mask): a SCALAR condition
    if (pmask || blockIdx.x == 0 || blockIdx.y == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.y == gridDim.y - 1)
        hull_candidate(flow, sign, pmask, H, W, x, y, ws);
    if (x >= cw || y >= ch) return;
    const size_t i00 = (size_t)y * W + x;
    bool k0 = true, k1 = true, k2 = true, k3 = true;
    if (pmask) {
        k0 = pmask[i00] != 0; k1 = pmask[i00 + 1] != 0;
        k2 = pmask[i00 + W + 1] != 0; k3 = pmask[i00 + W] != 0;
    }
    const int n_keep = (int)k0 + (int)k1 + (int)k2 + (int)k3;
    if (n_keep < 3) return;
    const D2 pa = point_of(flow, sign, W, x, y), pb = point_of(flow, sign, W, x + 1, y);
    const D2 pc = point_of(flow, sign, W, x + 1, y + 1), pd = point_of(flow, sign, W, x, y + 1);
    int diag;
    if (n_keep == 4) diag = pick_diagonal(pa, pb, pc, pd);
    else diag = (!k0 || !k2) ? 1 : 0;     // the only diagonal that leaves a fully kept triangle
    // a cell that is not properly oriented (folded mesh) can push interior points onto the convex hull
    // of the point set: its corners become hull candidates (duplicates are harmless)
    // (a point strictly inside a triangle of kept points is never a hull vertex: corners inside the two
    // triangles of the four guard points -- the kept points next to the image corners -- are dropped: motion
    // boundaries fold thousands of cells far away from the hull)
    if (!(cross2(pa, pb, pc) > 0 && cross2(pa, pc, pd) > 0 && cross2(pb, pc, pd) > 0 && cross2(pb, pd, pa) > 0)) {
        const bool guard = *ws.guard_ok != 0;
        D2 ga = { 0.0, 0.0 }, gb = ga, gc = ga, gd = ga;
        if (guard) { ga = ws.guard[0]; gb = ws.guard[1]; gc = ws.guard[2]; gd = ws.guard[3]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const D2 p = pick4(k, pa, pb, pc, pd);
            bool want = pick4(k, k0, k1, k2, k3);
            if (want && guard && (strictly_inside(ga, gb, gc, p) || strictly_inside(ga, gc, gd, p))) want = false;
            push_candidate(ws, want, p);            // one atomic per wave: ballots count the lanes that are here
        }
    }
    const uint32_t cell = (uint32_t)(y * cw + x);
    unsigned coop_bits = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        int i0, i1, i2;
        tri_corners(diag, t, i0, i1, i2);
        if (!(pick4(i0, k0, k1, k2, k3) && pick4(i1, k0, k1, k2, k3) && pick4(i2, k0, k1, k2, k3))) continue;
        const D2 q0 = pick4(i0, pa, pb, pc, pd), q1 = pick4(i1, pa, pb, pc, pd), q2 = pick4(i2, pa, pb, pc, pd);
        const TriBox b = tri_box_rows(q0, q1, q2, W, H, ws);
        if (b.x1 < b.x0 || b.y1 < b.y0) continue;
        const uint32_t id = tri_id(cell, diag, t);
        const long long area = (long long)(b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1);
        if (area > kSmallArea) {
            const unsigned long long slot = atomicAdd(&ws.counters[1], 1ull);
            atomicAdd(&ws.counters[2], (unsigned long long)area);
            if (slot < ws.big_cap) ws.big[slot] = id;
            continue;
        }
        // Load balance inside the wave: a lane whose bounding box is much larger than its neighbours' (motion
        // boundaries, seams of tiled fields) would keep the other lanes waiting.  When few lanes of the wave
        // hold such a triangle it is only FLAGGED here and scanned by the whole wave afterwards.
        if (OFL_SC_COOP && area > kCoopMinArea && __popcll(__ballot(area > kCoopMinArea)) <= 16) {
            coop_bits |= 1u << t;
            continue;
        }
        TriEdge te;
        if (!tri_setup(q0, q1, q2, te)) continue;
        for (int gy = b.y0; gy <= b.y1; ++gy)
            for (int gx = b.x0; gx <= b.x1; ++gx)
                if (tri_inside(te, (double)gx, (double)gy))
                    atomicMin(&ws.owner[(size_t)gy * W + gx], id);
    }
    if (!OFL_SC_COOP || !__any(coop_bits != 0)) return;
    const int lane = threadIdx.x & 63;
    for (int t = 0; t < 2; ++t) {
        unsigned long long todo = __ballot((coop_bits >> t) & 1u);
        while (todo) {
            const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
            todo &= todo - 1;
            // every lane rebuilds the shared triangle from its id (three cached loads)
            const uint32_t bid = tri_id((uint32_t)__builtin_amdgcn_readlane((int)cell, src), __builtin_amdgcn_readlane(diag, src), t);
            size_t vi[3];
            D2 vp[3];
            tri_decode(bid, flow, sign, W, vi, vp);
            const TriBox bb = tri_box_rows(vp[0], vp[1], vp[2], W, H, ws);
            TriEdge tb;
            if (!tri_setup(vp[0], vp[1], vp[2], tb)) continue;
            if (bb.x1 < bb.x0 || bb.y1 < bb.y0) continue;
            const int bw = bb.x1 - bb.x0 + 1, n = bw * (bb.y1 - bb.y0 + 1);
            for (int k = lane; k < n; k += 64) {
                const int ry = k / bw, gx = bb.x0 + (k - ry * bw), gy = bb.y0 + ry;
                if (tri_inside(tb, (double)gx, (double)gy))
                    atomicMin(&ws.owner[(size_t)gy * W + gx], bid);
            }
        }
    }
}

// y-range of a convex polygon at abscissa qx from one of its x-monotone chains (binary search)
__device__ __forceinline__ bool chain_y(const D2 *ch, int n, double qx, double &yq, double &slack)
{
    if (n <= 0 || qx < ch[0].x || qx > ch[n - 1].x) return false;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ch[mid].x <= qx) lo = mid; else hi = mid;
    }
    const D2 a = ch[lo], b = ch[hi];
    const double dx = b.x - a.x, dy = b.y - a.y;
    if (dx <= 0.0) { yq = a.y; slack = kHullTol; return true; }
    yq = a.y + dy * ((qx - a.x) / dx);
    slack = kHullTol * sqrt(1.0 + (dy / dx) * (dy / dx));     // tolerance measured perpendicular to the edge
    return true;
}

__device__ __forceinline__ bool inside_hull(const HullRef &h, double qx, double qy)
{
    double ylo, yhi, s0, s1;
    if (!chain_y(h.lower, h.n_lower, qx, ylo, s0) || !chain_y(h.upper, h.n_upper, qx, yhi, s1)) return false;
    // at the extreme abscissae the chains end in vertical hull edges
    if (qx == h.lower[0].x) { ylo = fmin(ylo, h.lower[0].y); yhi = fmax(yhi, h.upper[0].y); }
    if (qx == h.lower[h.n_lower - 1].x) { ylo = fmin(ylo, h.lower[h.n_lower - 1].y); yhi = fmax(yhi, h.upper[h.n_upper - 1].y); }
    return qy >= ylo - s0 && qy <= yhi + s1;
}

// nearest covered grid node around (cx, cy): rings of growing Chebyshev radius, ties to the smallest
// Euclidean distance then the smallest owner id (deterministic)
// RINGS: 0 = all rings, 1 = rings 0 .. 2 only, 2 = rings 3 .. kFillRadius only (the two halves of the gap pass)
template <int RINGS>
__device__ __forceinline__ uint32_t nearest_owner(const uint32_t *owner, int y_lo, int y_hi, int W, int cx, int cy, double qx, double qy)
{
    // ring r = the 8 r nodes at Chebyshev distance r.  One dependent load per node made the waves at the rim of a hole
    // spend 170 us in this loop at 4K (the slowest lane sets the time of the kernel): the probes are batched.
    uint32_t best = kNoOwner;
    double bestd = 1e300;
    auto probe = [&](int xx, int yy) -> uint32_t {
        return (xx >= 0 && xx < W && yy >= y_lo && yy < y_hi) ? owner[(size_t)yy * W + xx] : kNoOwner;
    };
    auto take = [&](uint32_t id, int xx, int yy) {
        if (id >= kGapDone) return;
        const double d = (xx - qx) * (xx - qx) + (yy - qy) * (yy - qy);
        if (d < bestd || (d == bestd && id < best)) { bestd = d; best = id; }
    };
    if (RINGS != 2) {
        const uint32_t id = probe(cx, cy);
        if (id < kGapDone) return id;
    }
    // the ring as four runs of 2 r nodes -- top (dx = -r .. r-1, dy = -r), bottom (dx = -r+1 .. r, dy = r), left
    // (dx = -r, dy = -r+1 .. r), right (dx = r, dy = -r .. r-1): every node once; B nodes of each run per round trip
    auto ring = [&](int r, auto batch) {
        constexpr int B = decltype(batch)::value;
        for (int t = 0; t < 2 * r; t += B) {
            uint32_t id[4][B];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sx = (k < 2) ? 1 : 0, sy = 1 - sx;
                const int x0 = cx + ((k == 0 || k == 2) ? -r : (k == 1 ? -r + 1 : r));
                const int y0 = cy + ((k == 0 || k == 3) ? -r : (k == 1 ? r : -r + 1));
#pragma unroll
                for (int u = 0; u < B; ++u) id[k][u] = (t + u < 2 * r) ? probe(x0 + sx * (t + u), y0 + sy * (t + u)) : kNoOwner;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sx = (k < 2) ? 1 : 0, sy = 1 - sx;
                const int x0 = cx + ((k == 0 || k == 2) ? -r : (k == 1 ? -r + 1 : r));
                const int y0 = cy + ((k == 0 || k == 3) ? -r : (k == 1 ? r : -r + 1));
#pragma unroll
                for (int u = 0; u < B; ++u) take(id[k][u], x0 + sx * (t + u), y0 + sy * (t + u));
            }
        }
    };
    // isolated dropped points (speckled masks) end at ring 1 or 2: exactly their 8 / 16 probes; wider rings keep 32
    // probes in flight per round trip -- 40 round trips for rings 3 .. 16 instead of one per node
    if (RINGS != 2) {
        ring(1, std::integral_constant<int, 2>());
        if (best != kNoOwner) return best;
        ring(2, std::integral_constant<int, 4>());
        if (best != kNoOwner) return best;
    }
    if (RINGS != 1) {
        for (int r = 3; r <= kFillRadius; ++r) {
            ring(r, std::integral_constant<int, 8>());
            if (best != kNoOwner) return best;
        }
    }
    return kNoOwner;
}

// pass 1b: one wave per large triangle, 64 nodes of the bounding box per step
__global__ __launch_bounds__(256)
void scatter_big_kernel(const float *__restrict__ flow, int sign, int H, int W, ScatterWs ws)
{
    unsigned long long n = ws.counters[1];
    if (n > ws.big_cap) n = ws.big_cap;
    const int lane = threadIdx.x & 63;
    for (unsigned long long k = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6); k < n;
         k += (unsigned long long)gridDim.x * 4) {
        const uint32_t id = ws.big[k];
        size_t vi[3];
        D2 vp[3];
        tri_decode(id, flow, sign, W, vi, vp);
        const TriBox b = tri_box_rows(vp[0], vp[1], vp[2], W, H, ws);
        if (b.x1 < b.x0 || b.y1 < b.y0) continue;
        const long long bw = b.x1 - b.x0 + 1, area = bw * (b.y1 - b.y0 + 1);
        TriEdge te;
        if (!tri_setup(vp[0], vp[1], vp[2], te)) continue;
        for (long long j = lane; j < area; j += 64) {
            const int gy = b.y0 + (int)(j / bw), gx = b.x0 + (int)(j % bw);
            if (tri_inside(te, (double)gx, (double)gy))
                atomicMin(&ws.owner[(size_t)gy * W + gx], id);
        }
    }
}

// Triangle containing an arbitrary query point: seed = owner of the nearest grid node (or a neighbour),
// then the 3 x 3 cells around the seed's cell are tested with SciPy's inclusion rule.
__device__ __forceinline__ bool locate_query(const float *flow, int sign, const uint8_t *pmask, int H, int W,
                                             const ScatterWs &ws, double qx, double qy, uint32_t &id,
                                             size_t (&vi)[3], D2 (&vp)[3], double &c0, double &c1, double &c2)
{
    if (!(qx >= -1.0 && qx <= (double)W && qy >= -1.0 && qy <= (double)H)) return false;
    const int nx = (int)fmin(fmax(rint(qx), 0.0), (double)(W - 1)), ny = (int)fmin(fmax(rint(qy), 0.0), (double)(H - 1));
    uint32_t seed = kNoOwner;
    for (int dy = 0; dy <= 1 && seed == kNoOwner; ++dy)
        for (int dx = 0; dx <= 1 && seed == kNoOwner; ++dx) {
            const int sx = min(max(nx + (dx ? (qx < nx ? -1 : 1) : 0), 0), W - 1);
            const int sy = min(max(ny + (dy ? (qy < ny ? -1 : 1) : 0), 0), H - 1);
            seed = ws.owner[(size_t)sy * W + sx];
        }
    if (seed == kNoOwner) return false;
    const int cw = W - 1, chh = H - 1;
    const uint32_t cell = seed >> 2;
    const int sy = (int)(cell / (uint32_t)cw), sx = (int)(cell - (uint32_t)sy * (uint32_t)cw);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int cx = sx + dx, cy = sy + dy;
            if (cx < 0 || cy < 0 || cx >= cw || cy >= chh) continue;
            const size_t i00 = (size_t)cy * W + cx;
            bool k0 = true, k1 = true, k2 = true, k3 = true;
            if (pmask) {
                k0 = pmask[i00] != 0; k1 = pmask[i00 + 1] != 0;
                k2 = pmask[i00 + W + 1] != 0; k3 = pmask[i00 + W] != 0;
            }
            const int n_keep = (int)k0 + (int)k1 + (int)k2 + (int)k3;
            if (n_keep < 3) continue;
            const D2 pa = point_of(flow, sign, W, cx, cy), pb = point_of(flow, sign, W, cx + 1, cy);
            const D2 pc = point_of(flow, sign, W, cx + 1, cy + 1), pd = point_of(flow, sign, W, cx, cy + 1);
            const int diag = n_keep == 4 ? pick_diagonal(pa, pb, pc, pd) : ((!k0 || !k2) ? 1 : 0);
            for (int t = 0; t < 2; ++t) {
                int i0, i1, i2;
                tri_corners(diag, t, i0, i1, i2);
                if (!(pick4(i0, k0, k1, k2, k3) && pick4(i1, k0, k1, k2, k3) && pick4(i2, k0, k1, k2, k3))) continue;
                if (bary(pick4(i0, pa, pb, pc, pd), pick4(i1, pa, pb, pc, pd), pick4(i2, pa, pb, pc, pd), qx, qy, c0, c1, c2)) {
                    id = tri_id((uint32_t)(cy * cw + cx), diag, t);
                    tri_decode(id, flow, sign, W, vi, vp);
                    return true;
                }
            }
        }
    return false;
}

// Gap fill shared by the dense and the sparse pass (see scatter_resolve_kernel).
template <int RINGS = 0>
__device__ __forceinline__ bool fill_from_nearest(const float *flow, int sign, int H, int W, const ScatterWs &ws,
                                                  const HullRef &hull, double qx, double qy, uint32_t &id,
                                                  size_t (&vi)[3], D2 (&vp)[3], double &c0, double &c1, double &c2)
{
    if (!(hull.n_lower > 0 && inside_hull(hull, qx, qy))) return false;
    const int nx = (int)fmin(fmax(rint(qx), 0.0), (double)(W - 1)), ny = (int)fmin(fmax(rint(qy), 0.0), (double)(H - 1));
    id = nearest_owner<RINGS>(ws.owner, ws.oy0, ws.oy1, W, nx, ny, qx, qy);
    if (id == kNoOwner) return false;
    tri_decode(id, flow, sign, W, vi, vp);
    (void)bary(vp[0], vp[1], vp[2], qx, qy, c0, c1, c2);
    const double e1x = vp[1].x - vp[0].x, e1y = vp[1].y - vp[0].y, e2x = vp[2].x - vp[0].x, e2y = vp[2].y - vp[0].y;
    return (e1x * e2y - e1y * e2x) != 0.0;
}

// pass 2 for sparse query points (one thread per point, float64 in and out)
__global__ __launch_bounds__(256)
void scatter_query_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask,
                          const float *__restrict__ vals, int C, int H, int W,
                          const double *__restrict__ query_xy, size_t n_query,
                          double *__restrict__ out, uint8_t *__restrict__ found_out, ScatterWs ws, HullRef hull)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_query; i += (size_t)gridDim.x * blockDim.x) {
        const double qx = query_xy[2 * i], qy = query_xy[2 * i + 1];
        uint32_t id = kNoOwner;
        size_t vi[3];
        D2 vp[3];
        double c0 = 0, c1 = 0, c2 = 0;
        bool found = locate_query(flow, sign, pmask, H, W, ws, qx, qy, id, vi, vp, c0, c1, c2);
        if (!found) found = fill_from_nearest(flow, sign, H, W, ws, hull, qx, qy, id, vi, vp, c0, c1, c2);
        for (int c = 0; c < C; ++c)
            out[i * C + c] = found ? c0 * (double)vals[vi[0] * C + c] + c1 * (double)vals[vi[1] * C + c] + c2 * (double)vals[vi[2] * C + c] : 0.0;
        found_out[i] = found ? 1 : 0;
    }
}

// pass 2, grid nodes: node (x, y) takes the triangle the raster pass recorded for it.  Nodes without an owner are
// zero / invalid (NaN -> 0, utils.py:254) -- unless they lie inside the convex hull of the kept points: SciPy's
// Delaunay triangulation spans such gaps (ragged / curved mesh borders, holes left by dropped points) with
// triangles between border vertices; those nodes are queued for pass 2b, which continues the linear function of
// the nearest cell triangle into them (identical for data that is affine across the gap).  Keeping the rare search
// out of this kernel keeps it at 8 waves per SIMD.
template <typename VT>
__global__ __launch_bounds__(256)
void scatter_resolve_grid_kernel(const float *__restrict__ flow, int sign,
                                 const VT *__restrict__ vals, int C, const uint8_t *__restrict__ vmask,
                                 int H, int W, VT *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule,
                                 ScatterWs ws, HullRef hull, int row0, int rows, int er0, int erows)
{
    // rows [row0, row0 + rows) of the grid are resolved; out / valid hold rows [er0, er0 + erows) only (the two ranges
    // differ for a row band under a point mask: gap nodes are then marked over the whole field, exactly as the
    // single-GPU call marks them, and only the band is written)
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int yl = blockIdx.y * 8 + (threadIdx.x >> 5), y = row0 + yl;
    const bool act = x < W && yl < rows;
    const bool emit = (unsigned)(y - er0) < (unsigned)erows;
    const size_t o = (size_t)(y - er0) * W + x;
    const uint32_t id = act ? ws.owner[(size_t)y * W + x] : 0u;
    bool gap = false;
    if (act) {
        if (id != kNoOwner) {      // the raster pass decided containment; only the coordinates are needed
            size_t vi[3];
            D2 vp[3];
            double c0, c1, c2;
            tri_decode(id, flow, sign, W, vi, vp);
            (void)bary(vp[0], vp[1], vp[2], (double)x, (double)y, c0, c1, c2);
            if (emit) resolve_emit(vals, C, vmask, vi, c0, c1, c2, valid_rule, out, valid, o);
        } else {
            if (emit) {
                for (int c = 0; c < C; ++c) out[o * C + c] = (VT)0;
                if (valid) valid[o] = 0;
            }
            gap = hull.n_lower > 0 && inside_hull(hull, (double)x, (double)y);
        }
    }
    // coverage of this 32 x 8 block for pass 2b (blocks are aligned with the grid only when row0 is a multiple of 8:
    // otherwise the map keeps its "unknown" default)
    {
        const int covered = __syncthreads_or(act && id != kNoOwner);
        if (threadIdx.x == 0 && (row0 & 7) == 0) ws.coarse[(size_t)((row0 >> 3) + blockIdx.y) * ws.coarse_w + blockIdx.x] = covered ? 1 : 0;
    }
    // gap nodes are MARKED in the owner map (a plain store: a counter bumped by every wave that holds one serialises
    // at ~12 ns per atomic -- 1.5 ms with a speckled mask); pass 2b scans the map for the marks
    if (gap) ws.owner[(size_t)y * W + x] = kGapOwner;
}

__device__ __forceinline__ void note_deep_node(const ScatterWs &ws, int x, int y, int H, int W);

// pass 2b / 2c: the marked gap nodes.  FAR = false looks at rings 0 .. 2 (isolated dropped points: a lean kernel at full
// occupancy) and re-marks what it could not fill; FAR = true searches rings 3 .. 16 for those (rims of holes).
template <bool FAR, typename VT>
__global__ __launch_bounds__(256)
void scatter_gap_kernel(const float *__restrict__ flow, int sign,
                        const VT *__restrict__ vals, int C, const uint8_t *__restrict__ vmask,
                        int H, int W, VT *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule,
                        ScatterWs ws, HullRef hull, int row0, int rows, int er0, int erows)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int yl = blockIdx.y * 8 + (threadIdx.x >> 5), y = row0 + yl;
    if (x >= W || yl >= rows || ws.owner[(size_t)y * W + x] != (FAR ? kGapFar : kGapOwner)) return;
    const bool emit = (unsigned)(y - er0) < (unsigned)erows;
    uint32_t id;
    size_t vi[3];
    D2 vp[3];
    double c0, c1, c2;
    if (fill_from_nearest<FAR ? 2 : 1>(flow, sign, H, W, ws, hull, (double)x, (double)y, id, vi, vp, c0, c1, c2)) {
        if (emit) resolve_emit(vals, C, vmask, vi, c0, c1, c2, valid_rule, out, valid, (size_t)(y - er0) * W + x);
        ws.owner[(size_t)y * W + x] = kGapDone;
        return;
    }
    if (FAR) {
        note_deep_node(ws, x, y, H, W);
        return;
    }
    if (!FAR) {
        // anything covered within the full search radius?  One byte per 32 x 8 block of the window instead of 1 089
        // owner words (nodes deep inside a hole of the point mask give up here)
        const int bx0 = max(x - kFillRadius, 0) >> 5, bx1 = min(x + kFillRadius, W - 1) >> 5;
        const int by0 = max(y - kFillRadius, 0) >> 3, by1 = min(y + kFillRadius, H - 1) >> 3;
        bool any = false;
        for (int by = by0; by <= by1 && !any; ++by)
            for (int bx = bx0; bx <= bx1 && !any; ++bx) any = ws.coarse[(size_t)by * ws.coarse_w + bx] != 0;
        if (any) ws.owner[(size_t)y * W + x] = kGapFar;
        else note_deep_node(ws, x, y, H, W);
    }
}

// a gap node the ring search cannot reach: raise the flag and grow the bounding box of such nodes (four running maxima
// in the header -- W-1-x, H-1-y, x, y -- so that the zeroed header is the empty box; the atomic is only issued when it
// would change the box, which after the first few nodes it rarely does)
__device__ __forceinline__ void note_deep_node(const ScatterWs &ws, int x, int y, int H, int W)
{
    ws.counters[5] = 1ull;
    int *box = reinterpret_cast<int *>(ws.counters + 6);
    const int v[4] = { W - 1 - x, H - 1 - y, x, y };
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (__hip_atomic_load(box + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v[k]) atomicMax(box + k, v[k]);
}

// Pass 2d (only when a gap node lies deeper than kFillRadius inside a hole of the point mask -- SciPy bridges such holes
// with long triangles and reports them valid): the nearest covered node of EVERY node by jump flooding (log2(size)
// passes over two node-sized maps in the large-triangle list), then the remaining gap nodes continue the triangle of
// their nearest covered node like the ring search does.  Ties: smaller squared distance, then smaller node index.
constexpr uint32_t kNoSeed = 0xFFFFFFFFu;

// the maps cover the rectangle [bx0, bx0 + bw) x [by0, by0 + bh) of the grid (the bounding box of the deep nodes plus
// kFillRadius + 2 nodes: the nearest covered node of a node inside a hole lies on the rim of that hole); seeds are GLOBAL node ids
__global__ __launch_bounds__(256)
void scatter_jfa_init_kernel(const uint32_t *__restrict__ owner, uint32_t *__restrict__ seed, int W, int bx0, int by0, int bw, int bh)
{
    const int lx = blockIdx.x * 32 + (threadIdx.x & 31), ly = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (lx >= bw || ly >= bh) return;
    const uint32_t g = (uint32_t)((size_t)(by0 + ly) * W + (bx0 + lx));
    seed[(size_t)ly * bw + lx] = owner[g] < kGapDone ? g : kNoSeed;
}

__global__ __launch_bounds__(256)
void scatter_jfa_step_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ outs, int W, int bx0, int by0, int bw, int bh, int step)
{
    const int lx = blockIdx.x * 32 + (threadIdx.x & 31), ly = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (lx >= bw || ly >= bh) return;
    const int x = bx0 + lx, y = by0 + ly;
    uint32_t best = in[(size_t)ly * bw + lx];
    long long bestd = 0x7fffffffffffffffll;
    if (best != kNoSeed) { const int sy = (int)(best / (uint32_t)W), sx = (int)(best - (uint32_t)sy * (uint32_t)W); bestd = (long long)(sx - x) * (sx - x) + (long long)(sy - y) * (sy - y); }
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if (dx == 0 && dy == 0) continue;
            const int xx = lx + dx * step, yy = ly + dy * step;
            if (xx < 0 || xx >= bw || yy < 0 || yy >= bh) continue;
            const uint32_t c = in[(size_t)yy * bw + xx];
            if (c == kNoSeed) continue;
            const int sy = (int)(c / (uint32_t)W), sx = (int)(c - (uint32_t)sy * (uint32_t)W);
            const long long d = (long long)(sx - x) * (sx - x) + (long long)(sy - y) * (sy - y);
            if (d < bestd || (d == bestd && c < best)) { bestd = d; best = c; }
        }
    outs[(size_t)ly * bw + lx] = best;
}

template <typename VT>
__global__ __launch_bounds__(256)
void scatter_deep_kernel(const float *__restrict__ flow, int sign,
                         const VT *__restrict__ vals, int C, const uint8_t *__restrict__ vmask,
                         int H, int W, VT *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule,
                         ScatterWs ws, const uint32_t *__restrict__ seed, int row0, int rows, int bx0, int by0, int bw, int bh)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int yl = blockIdx.y * 8 + (threadIdx.x >> 5), y = row0 + yl;
    if (x >= W || yl >= rows) return;
    if (x < bx0 || x >= bx0 + bw || y < by0 || y >= by0 + bh) return;
    const uint32_t mark = ws.owner[(size_t)y * W + x];
    if (mark != kGapOwner && mark != kGapFar) return;
    const uint32_t sd = seed[(size_t)(y - by0) * bw + (x - bx0)];
    if (sd == kNoSeed) return;
    const uint32_t id = ws.owner[sd];
    size_t vi[3];
    D2 vp[3];
    double c0, c1, c2;
    tri_decode(id, flow, sign, W, vi, vp);
    (void)bary(vp[0], vp[1], vp[2], (double)x, (double)y, c0, c1, c2);
    const double e1x = vp[1].x - vp[0].x, e1y = vp[1].y - vp[0].y, e2x = vp[2].x - vp[0].x, e2y = vp[2].y - vp[0].y;
    if ((e1x * e2y - e1y * e2x) == 0.0) return;
    resolve_emit(vals, C, vmask, vi, c0, c1, c2, valid_rule, out, valid, (size_t)yl * W + x);
}

// pass 2 for arbitrary sample positions (query != NULL; mode 2 / ref 't', flow_class.py:1398-1410): the triangle
// containing the query point is searched in the 3 x 3 cells around the owner of the nearest node.
__global__ __launch_bounds__(256)
void scatter_resolve_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask,
                            const float *__restrict__ vals, int C, const uint8_t *__restrict__ vmask,
                            int H, int W, const float *__restrict__ query,
                            float *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule, ScatterWs ws,
                            HullRef hull)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const size_t o = (size_t)y * W + x;
    uint32_t id = kNoOwner;
    size_t vi[3];
    D2 vp[3];
    double c0 = 0, c1 = 0, c2 = 0;
    const float2 q = *reinterpret_cast<const float2 *>(query + o * 2);
    const double qx = (double)q.x, qy = (double)q.y;
    bool found = locate_query(flow, sign, pmask, H, W, ws, qx, qy, id, vi, vp, c0, c1, c2);
    if (!found) found = fill_from_nearest(flow, sign, H, W, ws, hull, qx, qy, id, vi, vp, c0, c1, c2);      // gap fill, see above
    if (!found) {
        for (int c = 0; c < C; ++c) out[o * C + c] = 0.0f;          // NaN -> 0, utils.py:254 / fill_value=0
        if (valid) valid[o] = 0;
        return;
    }
    resolve_emit(vals, C, vmask, vi, c0, c1, c2, valid_rule, out, valid, o);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int cand_cap_for(int H, int W)
{
    const long long n = (long long)H * W;
    return (int)(n < kCandCap ? n : kCandCap);
}

constexpr size_t kHeadBytes  = 256 + 256 * sizeof(unsigned long long);   // counters + kept-point slots
constexpr int    kFirstCand  = 16384;                                     // candidates fetched with the header

ScatterWs carve(void *workspace, int H, int W)
{
    ScatterWs ws;
    char *p = (char *)workspace;
    ws.cand_cap = cand_cap_for(H, W);
    ws.counters = (unsigned long long *)p;                              // header, slots and candidates are
    ws.guard = (const D2 *)(p + 64);
    ws.guard_ok = (const int *)(p + 128);
    p += 256;
    ws.kept_slots = (unsigned long long *)p; p += 256 * sizeof(unsigned long long);   // contiguous: ONE read-back
    ws.cand = (D2 *)p;                       p += align_up((size_t)ws.cand_cap * sizeof(D2), 256);
    ws.lower = (D2 *)p;                      p += (size_t)kHullCap * sizeof(D2);      // lower | upper contiguous:
    ws.upper = (D2 *)p;                      p += (size_t)kHullCap * sizeof(D2);      // ONE upload
    ws.big_cap = (unsigned long long)big_cap_for(H, W);
    ws.big = (uint32_t *)p;                  p += align_up((size_t)ws.big_cap * 4, 256);
    ws.owner = (uint32_t *)p;                p += align_up((size_t)H * W * 4, 256);
    ws.coarse = (uint8_t *)p;
    ws.coarse_w = (W + 31) / 32;
    ws.oy0 = 0;
    ws.oy1 = H;
    return ws;
}

// Andrew's monotone chain on the host: lower and upper chains, both with x ascending and both
// containing the two extreme-x end points.
void convex_chains(std::vector<D2> &pts, std::vector<D2> &lower, std::vector<D2> &upper)
{
    // sort by (x, y): counting sort into ~n/4 x-buckets, then tiny std::sorts -- the candidates are a few
    // thousand points spread evenly along the mesh border, a comparison sort of all of them costs ~0.5 ms
    const size_t n = pts.size();
    double xmin = pts[0].x, xmax = pts[0].x;
    for (const D2 &p : pts) { xmin = std::min(xmin, p.x); xmax = std::max(xmax, p.x); }
    const size_t nb = std::max<size_t>(1, n / 4);
    const double scale = xmax > xmin ? (double)(nb - 1) / (xmax - xmin) : 0.0;
    std::vector<uint32_t> start(nb + 1, 0);
    std::vector<uint32_t> key(n);
    for (size_t i = 0; i < n; ++i) {
        size_t b = (size_t)((pts[i].x - xmin) * scale);
        if (b >= nb) b = nb - 1;
        key[i] = (uint32_t)b;
        ++start[b + 1];
    }
    for (size_t b = 0; b < nb; ++b) start[b + 1] += start[b];
    std::vector<D2> sorted(n);
    std::vector<uint32_t> fill(start.begin(), start.end() - 1);
    for (size_t i = 0; i < n; ++i) sorted[fill[key[i]]++] = pts[i];
    auto less = [](const D2 &a, const D2 &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); };
    size_t out = 0;                                   // compacted in place: reduced buckets shrink
    for (size_t b = 0; b < nb; ++b) {
        const size_t s0 = start[b], e0 = start[b + 1];
        if (e0 - s0 > 8) {
            double bx0 = sorted[s0].x, bx1 = bx0;
            D2 lo = sorted[s0], hi = sorted[s0];
            for (size_t i = s0 + 1; i < e0; ++i) {
                bx0 = std::min(bx0, sorted[i].x); bx1 = std::max(bx1, sorted[i].x);
                if (sorted[i].y < lo.y) lo = sorted[i];
                if (sorted[i].y > hi.y) hi = sorted[i];
            }
            if (bx0 == bx1) {                         // one abscissa: only the lowest and the highest point matter
                sorted[out++] = lo;
                if (hi.y > lo.y) sorted[out++] = hi;
                continue;
            }
        }
        if (e0 - s0 > 1) std::sort(sorted.begin() + s0, sorted.begin() + e0, less);
        if (out != s0) std::copy(sorted.begin() + s0, sorted.begin() + e0, sorted.begin() + out);
        out += e0 - s0;
    }
    sorted.resize(out);
    auto cross = [](const D2 &o, const D2 &a, const D2 &b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); };
    lower.clear(); upper.clear();
    for (const D2 &p : sorted) {       // smallest y at every x
        while (lower.size() >= 2 && cross(lower[lower.size() - 2], lower.back(), p) <= 0) lower.pop_back();
        lower.push_back(p);
    }
    for (const D2 &p : sorted) {       // largest y at every x
        while (upper.size() >= 2 && cross(upper[upper.size() - 2], upper.back(), p) >= 0) upper.pop_back();
        upper.push_back(p);
    }
}

// pinned host staging shared by the scatter calls of this process (guarded by a mutex; the event marks
// the completion of the last upload that read from it)
struct HostStage {
    std::mutex lock;
    char      *buf = nullptr;
    hipEvent_t done = nullptr;
    size_t     bytes = 0;
};

HostStage &host_stage()
{
    static HostStage h;
    return h;
}

}  // namespace

namespace {
size_t legacy_workspace_bytes(int H, int W)
{
    return kHeadBytes + align_up((size_t)cand_cap_for(H, W) * sizeof(D2), 256) + 2 * (size_t)kHullCap * sizeof(D2) +
           align_up((size_t)big_cap_for(H, W) * 4, 256) + align_up((size_t)H * W * 4, 256) +
           align_up((size_t)((W + 31) / 32) * ((H + 7) / 8), 256);
}
}  // namespace

extern "C" {

int ofl_scatter_workspace_bytes(int H, int W, int C, size_t *bytes)
{
    (void)C;
    if (!bytes) return fail(OFL_E_INVALID, "ofl_scatter_workspace_bytes: NULL");
    if (H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_scatter_workspace_bytes: bad shape");
    *bytes = legacy_workspace_bytes(H, W);
    if ((long long)H * W < (1ll << 27)) *bytes = std::max(*bytes, exact_workspace_bytes(H, W));     // the exact path carves the same block
    return OFL_OK;
}

}  // extern "C"

namespace {

// Passes 1, 1b and the hull: everything the interpolation passes need.  `sign` arrives with the point
// precision already folded in (+-1 / +-2).
// [oy0, oy1): rows of the grid whose owners are needed (everything, or one row band plus the gap-fill radius).
int scatter_prepare(const float *flow, int sign, const uint8_t *pmask, int H, int W,
                    void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s,
                    ScatterWs &ws, HullRef &hull, int oy0 = 0, int oy1 = -1)
{
    if (!flow || !workspace) return fail(OFL_E_INVALID, "ofl_scatter_linear: NULL pointer");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 29))
        return fail(OFL_E_INVALID, "ofl_scatter_linear: H*W must be in [1, 2^29)");
    const size_t need = legacy_workspace_bytes(H, W);
    if (workspace_bytes < need) return fail(OFL_E_INVALID, "ofl_scatter_linear: workspace too small (%zu < %zu)", workspace_bytes, need);
    ws = carve(workspace, H, W);
    if (oy1 < 0) oy1 = H;
    OFL_HIP(hipMemsetAsync(ws.owner, 0xFF, (size_t)(oy1 - oy0) * W * 4, s));
    ws.oy0 = oy0;
    ws.oy1 = oy1;
    ws.owner -= (size_t)oy0 * W;         // biased base: owner[y * W + x] is row y of the covered range
    OFL_HIP(hipMemsetAsync(ws.counters, 0, kHeadBytes, s));
    OFL_HIP(hipMemsetAsync(ws.coarse, 1, (size_t)ws.coarse_w * ((H + 7) / 8), s));     // "unknown" until the resolve pass fills it in
    const dim3 grid((W + 31) / 32, (H + 7) / 8), block(256);
    hipLaunchKernelGGL(scatter_guard_kernel, dim3(1), block, 0, s, flow, sign, pmask, H, W, (D2 *)ws.guard, (int *)ws.guard_ok);
    hipLaunchKernelGGL(scatter_raster_kernel, grid, block, 0, s, flow, sign, pmask, H, W, ws);
    OFL_HIP(hipGetLastError());
    // the big-triangle list is usually empty; its length lives on the device, so the sweep kernel is
    // always enqueued with a modest grid and returns immediately when there is nothing to do
    hipLaunchKernelGGL(scatter_big_kernel, dim3(rt().n_cu * 4), block, 0, s, flow, sign, H, W, ws);
    OFL_HIP(hipGetLastError());
    // Convex hull of the kept points: candidates from the device, monotone chain on the host.  This is the
    // one place where the scatter path synchronises the stream: ONE read-back (header + first candidates)
    // into pinned memory, asynchronous uploads of the two chains.
    HostStage &hs = host_stage();
    std::lock_guard<std::mutex> guard(hs.lock);
    const size_t first_bytes = kHeadBytes + (size_t)kFirstCand * sizeof(D2);
    const size_t stage_bytes = first_bytes + 2 * (size_t)kHullCap * sizeof(D2);
    if (!hs.buf) {
        OFL_HIP(hipHostMalloc((void **)&hs.buf, stage_bytes, hipHostMallocDefault));
        OFL_HIP(hipEventCreateWithFlags(&hs.done, hipEventDisableTiming));
        hs.bytes = stage_bytes;
    } else {
        OFL_HIP(hipEventSynchronize(hs.done));          // the previous call's upload has left the buffer
    }
    static const bool timing = getenv("OFL_SC_TIMING") != nullptr;      // development knob: host-side split on stderr
    const auto t0 = std::chrono::steady_clock::now();
    const size_t avail = kHeadBytes + (size_t)std::min(ws.cand_cap, kFirstCand) * sizeof(D2);
    OFL_HIP(hipMemcpyAsync(hs.buf, ws.counters, avail, hipMemcpyDeviceToHost, s));
    OFL_HIP(hipStreamSynchronize(s));
    const auto t1 = std::chrono::steady_clock::now();
    const unsigned long long *cbuf = (const unsigned long long *)hs.buf;
    unsigned long long c[4] = { 0, cbuf[1], cbuf[2], cbuf[3] };
    for (int k = 0; k < 256; ++k) c[0] += cbuf[32 + k];
    if (info_host) {
        info_host[0] = pmask ? c[0] : (uint64_t)H * W;
        info_host[1] = c[1];
        info_host[2] = c[2];
    }
    if (pmask && c[0] == 0) return fail(OFL_E_NOPOINTS, "ofl_scatter_linear: no valid source points");
    if (c[1] > ws.big_cap)      // cannot happen: the list holds every triangle of the mesh
        return fail(OFL_E_INVALID, "ofl_scatter_linear: %llu large triangles exceed the list (%llu)", c[1], ws.big_cap);
    hull = HullRef{ ws.lower, ws.upper, 0, 0 };
    if (c[3] >= 3 && c[3] <= (unsigned long long)ws.cand_cap) {
        std::vector<D2> pts((size_t)c[3]), lower, upper;
        const size_t have = std::min<size_t>(pts.size(), (size_t)kFirstCand);
        memcpy(pts.data(), hs.buf + kHeadBytes, have * sizeof(D2));
        if (pts.size() > have)      // rare: more candidates than came with the header
            OFL_HIP(hipMemcpy(pts.data() + have, ws.cand + have, (pts.size() - have) * sizeof(D2), hipMemcpyDeviceToHost));
        convex_chains(pts, lower, upper);
        if (lower.size() >= 2 && upper.size() >= 2 && lower.size() <= (size_t)kHullCap && upper.size() <= (size_t)kHullCap) {
            char *up = hs.buf + first_bytes;
            memcpy(up, lower.data(), lower.size() * sizeof(D2));
            memcpy(up + (size_t)kHullCap * sizeof(D2), upper.data(), upper.size() * sizeof(D2));
            OFL_HIP(hipMemcpyAsync(ws.lower, up, lower.size() * sizeof(D2), hipMemcpyHostToDevice, s));
            OFL_HIP(hipMemcpyAsync(ws.upper, up + (size_t)kHullCap * sizeof(D2), upper.size() * sizeof(D2), hipMemcpyHostToDevice, s));
            hull.n_lower = (int)lower.size();
            hull.n_upper = (int)upper.size();
        }
    }
    OFL_HIP(hipEventRecord(hs.done, s));
    if (timing) {
        const auto t2 = std::chrono::steady_clock::now();
        fprintf(stderr, "[ofl scatter] wait+readback %.1f us, hull+upload %.1f us, candidates %llu, hull %d+%d\n",
                std::chrono::duration<double, std::micro>(t1 - t0).count(),
                std::chrono::duration<double, std::micro>(t2 - t1).count(), c[3], hull.n_lower, hull.n_upper);
    }
    return OFL_OK;
}

}  // namespace

namespace {

// the gap passes after scatter_resolve_grid_kernel (grid mode): near, far, and -- when the device reports nodes deeper
// than the ring search reaches -- the jump-flooding fill.  Only a point mask can produce such nodes, and only then is
// the flag read back (one more host synchronisation).
template <typename VT>
int scatter_fill_gaps(const float *flow, int sign, const uint8_t *pmask, const VT *vals, int C, const uint8_t *vmask,
                             int H, int W, VT *out, uint8_t *valid, int valid_rule, const ScatterWs &ws, const HullRef &hull,
                             int row0, int rows, int er0, int erows, hipStream_t s)
{
    if (hull.n_lower <= 0) return OFL_OK;
    const dim3 pgrid((W + 31) / 32, (rows + 7) / 8), grid((W + 31) / 32, (erows + 7) / 8), block(256);
    hipLaunchKernelGGL((scatter_gap_kernel<false, VT>), pgrid, block, 0, s, flow, sign, vals, C, vmask, H, W,
                       out, valid, valid_rule, ws, hull, row0, rows, er0, erows);
    hipLaunchKernelGGL((scatter_gap_kernel<true, VT>), pgrid, block, 0, s, flow, sign, vals, C, vmask, H, W,
                       out, valid, valid_rule, ws, hull, row0, rows, er0, erows);
    OFL_HIP(hipGetLastError());
    if (!pmask || ws.oy0 != 0 || ws.oy1 != H || ws.big_cap < 2ull * (unsigned long long)H * W) return OFL_OK;
    unsigned long long head[4] = { 0, 0, 0, 0 };           // [0] flag, [1..2] the four box maxima as ints
    OFL_HIP(hipMemcpyAsync(head, ws.counters + 5, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    OFL_HIP(hipStreamSynchronize(s));
    if (!head[0]) return OFL_OK;
    const int *enc = reinterpret_cast<const int *>(head + 1);
    // the deep nodes start kFillRadius + 1 nodes inside their hole: the covered rim is that far outside their box
    const int rim = kFillRadius + 2;
    const int bx0 = std::max(W - 1 - enc[0] - rim, 0), by0 = std::max(H - 1 - enc[1] - rim, 0);
    const int bx1 = std::min(enc[2] + rim, W - 1), by1 = std::min(enc[3] + rim, H - 1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    const size_t n = (size_t)bw * bh;
    uint32_t *a = ws.big, *b = ws.big + n;
    const dim3 sub((bw + 31) / 32, (bh + 7) / 8);
    hipLaunchKernelGGL(scatter_jfa_init_kernel, sub, block, 0, s, ws.owner, a, W, bx0, by0, bw, bh);
    int step = 1;
    while (step * 2 < std::max(bw, bh)) step *= 2;
    for (; step >= 1; step /= 2) {
        hipLaunchKernelGGL(scatter_jfa_step_kernel, sub, block, 0, s, a, b, W, bx0, by0, bw, bh, step);
        std::swap(a, b);
    }
    hipLaunchKernelGGL(scatter_deep_kernel<VT>, grid, block, 0, s, flow, sign, vals, C, vmask, H, W, out, valid, valid_rule, ws, a, er0, erows,
                       bx0, by0, bw, bh);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}


// grid mode (node positions): argument checks, passes 0-1, the hull, then resolve + gap passes on rows [row0, row0 + rows)
template <typename VT>
int scatter_grid_impl(const char *who, const float *flow, int sign, int point_precision, const uint8_t *pmask,
                      const VT *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows, bool band,
                      VT *out, uint8_t *valid, int valid_rule, void *workspace, size_t workspace_bytes,
                      uint64_t *info_host, hipStream_t s)
{
    if (C < 0 || (C > 0 && (!vals || !out))) return fail(OFL_E_INVALID, "%s: C > 0 needs vals and out", who);
    if (C == 0 && !valid) return fail(OFL_E_INVALID, "%s: nothing to compute", who);
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "%s: sign must be +1 or -1", who);
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "%s: bad point_precision", who);
    if (point_precision == 1) sign *= 2;
    if ((valid_rule & ~(3 | OFL_SCATTER_ROUND | OFL_SCATTER_NEGATE)) || (valid_rule & 3) == 3) return fail(OFL_E_INVALID, "%s: bad valid_rule", who);
    if (H <= 0 || W <= 0 || row0 < 0 || rows <= 0 || row0 + rows > H)
        return fail(OFL_E_INVALID, "%s: rows [%d, %d) outside the %d-row grid", who, row0, row0 + rows, H);
    static const bool legacy = getenv("OFL_SC_LEGACY") != nullptr;       // development knob: always take the owner-map path
    if (!legacy && workspace && workspace_bytes >= 256) {
        // certified fast path: when the cell-wise mesh provably IS the Delaunay triangulation (ofl_scatter_walk.hip),
        // one kernel resolves the rows -- no owner map, no hull
        ofl_mesh_cert cert;
        OFL_TRY(certify_mesh(flow, sign, pmask, H, W, workspace, &cert, s));       // a point mask without zeros drops nothing
        if (cert.certified) {
            if (info_host) { info_host[0] = (uint64_t)H * W; info_host[1] = 0; info_host[2] = 0; }
            return walk_launch<VT>(flow, sign, vals, C, vmask, H, W, row0, rows, out, valid, valid_rule, &cert, nullptr, s);
        }
    }
    // everything else -- folds, dropped points, curved borders, sheared cells: a real Delaunay triangulation of the kept
    // points (ofl_delaunay.hip).  The owner-map path below remains for query positions and as a development reference.
    if (!legacy && workspace && workspace_bytes >= exact_workspace_bytes(H, W) && (long long)H * W < (1ll << 27))
        return exact_scatter<VT>(flow, sign, pmask, vals, C, vmask, H, W, row0, rows, out, valid, valid_rule,
                                 workspace, workspace_bytes, info_host, s);
    ScatterWs ws;
    HullRef hull;
    // owners are needed for the rows and, for the gap fill, kFillRadius rows around them -- or everywhere when a point
    // mask may leave holes deeper than that (the deep fill looks for the nearest covered node of the whole field)
    const int oy0 = (pmask || !band) ? 0 : std::max(0, row0 - kFillRadius), oy1 = (pmask || !band) ? H : std::min(H, row0 + rows + kFillRadius);
    OFL_TRY(scatter_prepare(flow, sign, pmask, H, W, workspace, workspace_bytes, info_host, s, ws, hull, oy0, oy1));
    // with a point mask the passes run over the whole field and only WRITE the band (see scatter_resolve_grid_kernel)
    const int pr0 = pmask ? 0 : row0, prows = pmask ? H : rows;
    const dim3 grid((W + 31) / 32, (prows + 7) / 8), block(256);
    hipLaunchKernelGGL(scatter_resolve_grid_kernel<VT>, grid, block, 0, s, flow, sign, vals, C, vmask, H, W,
                       out, valid, valid_rule, ws, hull, pr0, prows, row0, rows);
    OFL_HIP(hipGetLastError());
    return scatter_fill_gaps<VT>(flow, sign, pmask, vals, C, vmask, H, W, out, valid, valid_rule, ws, hull, pr0, prows, row0, rows, s);
}

}  // namespace

extern "C" {

int ofl_scatter_linear_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                           const float *vals, int C, const uint8_t *vmask, int H, int W,
                           const float *query, float *out, uint8_t *valid, int valid_rule,
                           void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream)
{
    OFL_TRY(need_device());
    hipStream_t s = stream_of(stream);
    if (!query)
        return scatter_grid_impl<float>("ofl_scatter_linear", flow, sign, point_precision, pmask, vals, C, vmask, H, W, 0, H, false,
                                        out, valid, valid_rule, workspace, workspace_bytes, info_host, s);
    // arbitrary sample positions (mode 2 / ref 't'): one resolve launch with the triangle search
    if (C < 0 || (C > 0 && (!vals || !out))) return fail(OFL_E_INVALID, "ofl_scatter_linear: C > 0 needs vals and out");
    if (C == 0 && !valid) return fail(OFL_E_INVALID, "ofl_scatter_linear: nothing to compute");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_scatter_linear: sign must be +1 or -1");
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "ofl_scatter_linear: bad point_precision");
    if (point_precision == 1) sign *= 2;
    if ((valid_rule & ~(3 | OFL_SCATTER_ROUND | OFL_SCATTER_NEGATE)) || (valid_rule & 3) == 3) return fail(OFL_E_INVALID, "ofl_scatter_linear: bad valid_rule");
    ScatterWs ws;
    HullRef hull;
    OFL_TRY(scatter_prepare(flow, sign, pmask, H, W, workspace, workspace_bytes, info_host, s, ws, hull));
    const dim3 grid((W + 31) / 32, (H + 7) / 8), block(256);
    hipLaunchKernelGGL(scatter_resolve_kernel, grid, block, 0, s, flow, sign, pmask, vals, C, vmask, H, W, query,
                       out, valid, valid_rule, ws, hull);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_scatter_certify_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask, int H, int W,
                            void *workspace, size_t workspace_bytes, ofl_mesh_cert *cert_host, void *stream)
{
    OFL_TRY(need_device());
    if (!flow || !workspace || !cert_host || workspace_bytes < 256) return fail(OFL_E_INVALID, "ofl_scatter_certify: NULL pointer / workspace < 256 bytes");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 29)) return fail(OFL_E_INVALID, "ofl_scatter_certify: H*W must be in [1, 2^29)");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_scatter_certify: sign must be +1 or -1");
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "ofl_scatter_certify: bad point_precision");
    return certify_mesh(flow, point_precision ? 2 * sign : sign, pmask, H, W, workspace, cert_host, stream_of(stream));
}

int ofl_scatter_certified_dev(const float *flow, int sign, int point_precision, const float *vals, int C,
                              const uint8_t *vmask, int H, int W, int row0, int rows, float *out_rows,
                              uint8_t *valid_rows, int valid_rule, const ofl_mesh_cert *cert,
                              uint32_t *fail_count_dev, void *stream)
{
    OFL_TRY(need_device());
    if (!flow || !cert) return fail(OFL_E_INVALID, "ofl_scatter_certified: NULL pointer");
    if (!cert->certified) return fail(OFL_E_INVALID, "ofl_scatter_certified: the field is not certified (use ofl_scatter_linear_dev)");
    if (C < 0 || (C > 0 && (!vals || !out_rows))) return fail(OFL_E_INVALID, "ofl_scatter_certified: C > 0 needs vals and out");
    if (C == 0 && !valid_rows) return fail(OFL_E_INVALID, "ofl_scatter_certified: nothing to compute");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_scatter_certified: sign must be +1 or -1");
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "ofl_scatter_certified: bad point_precision");
    if ((valid_rule & ~(3 | OFL_SCATTER_ROUND | OFL_SCATTER_NEGATE)) || (valid_rule & 3) == 3) return fail(OFL_E_INVALID, "ofl_scatter_certified: bad valid_rule");
    if (H < 2 || W < 2 || (long long)H * W >= (1ll << 29) || row0 < 0 || rows <= 0 || row0 + rows > H)
        return fail(OFL_E_INVALID, "ofl_scatter_certified: rows [%d, %d) outside the %d-row grid", row0, row0 + rows, H);
    return walk_launch<float>(flow, point_precision ? 2 * sign : sign, vals, C, vmask, H, W, row0, rows, out_rows, valid_rows,
                              valid_rule, cert, fail_count_dev, stream_of(stream));
}

int ofl_scatter_linear_f64_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                               const double *vals, int C, const uint8_t *vmask, int H, int W,
                               double *out, uint8_t *valid, int valid_rule,
                               void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream)
{
    OFL_TRY(need_device());
    return scatter_grid_impl<double>("ofl_scatter_linear_f64", flow, sign, point_precision, pmask, vals, C, vmask, H, W, 0, H, false,
                                     out, valid, valid_rule, workspace, workspace_bytes, info_host, stream_of(stream));
}

int ofl_scatter_rows_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                         const float *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows,
                         float *out_rows, uint8_t *valid_rows, int valid_rule,
                         void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream)
{
    OFL_TRY(need_device());
    return scatter_grid_impl<float>("ofl_scatter_rows", flow, sign, point_precision, pmask, vals, C, vmask, H, W, row0, rows, true,
                                    out_rows, valid_rows, valid_rule, workspace, workspace_bytes, info_host, stream_of(stream));
}

// Sparse queries (point tracking, utils.py:610-615): the triangle containing each of n_query points
// (x, y in float64) is located through the owner map; out[i][0..C) in float64, found[i] = 0 where the
// reference's griddata returns NaN.
int ofl_scatter_query_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                          const float *vals, int C, int H, int W,
                          const double *query_xy, size_t n_query, double *out, uint8_t *found,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    OFL_TRY(need_device());
    if (C <= 0 || !vals || !out || !found || !query_xy) return fail(OFL_E_INVALID, "ofl_scatter_query: NULL pointer / C <= 0");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_scatter_query: sign must be +1 or -1");
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "ofl_scatter_query: bad point_precision");
    if (point_precision == 1) sign *= 2;
    hipStream_t s = stream_of(stream);
    ScatterWs ws;
    HullRef hull;
    OFL_TRY(scatter_prepare(flow, sign, pmask, H, W, workspace, workspace_bytes, nullptr, s, ws, hull));
    if (n_query == 0) return OFL_OK;
    const size_t nb = (n_query + 255) / 256;
    hipLaunchKernelGGL(scatter_query_kernel, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, s,
                       flow, sign, pmask, vals, C, H, W, query_xy, n_query, out, found, ws, hull);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_scatter_linear(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                       const float *vals, int C, const uint8_t *vmask, int H, int W,
                       const float *query, float *out, uint8_t *valid, int valid_rule)
{
    OFL_TRY(need_device());
    if (H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_scatter_linear: bad shape");
    const size_t n = (size_t)H * W;
    hipStream_t s = rt().stream;
    size_t wsb = 0;
    OFL_TRY(ofl_scatter_workspace_bytes(H, W, C, &wsb));
    void *bufs[8] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    const void *src[6] = { flow, pmask, vals, vmask, query, nullptr };
    const size_t sz[8] = { n * 8, pmask ? n : 0, (size_t)C * n * 4, vmask ? n : 0, query ? n * 8 : 0,
                           (size_t)C * n * 4, valid ? n : 0, wsb };
    int rc = OFL_OK;
    uint64_t info[3];
    do {
        hipError_t e = hipSuccess;
        for (int k = 0; k < 8 && e == hipSuccess; ++k)
            if (sz[k]) e = hipMalloc(&bufs[k], sz[k]);
        if (e != hipSuccess) { rc = hip_fail(e, "hipMalloc"); break; }
        for (int k = 0; k < 5 && e == hipSuccess; ++k)
            if (sz[k] && src[k]) e = hipMemcpyAsync(bufs[k], src[k], sz[k], hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { rc = hip_fail(e, "H2D"); break; }
        rc = ofl_scatter_linear_dev((const float *)bufs[0], sign, point_precision, (const uint8_t *)bufs[1], (const float *)bufs[2], C,
                                    (const uint8_t *)bufs[3], H, W, (const float *)bufs[4], (float *)bufs[5],
                                    (uint8_t *)bufs[6], valid_rule, bufs[7], wsb, info, s);
        if (rc != OFL_OK) break;
        if (sz[5] && (e = hipMemcpyAsync(out, bufs[5], sz[5], hipMemcpyDeviceToHost, s)) != hipSuccess) { rc = hip_fail(e, "D2H"); break; }
        if (sz[6] && (e = hipMemcpyAsync(valid, bufs[6], sz[6], hipMemcpyDeviceToHost, s)) != hipSuccess) { rc = hip_fail(e, "D2H"); break; }
        if ((e = hipStreamSynchronize(s)) != hipSuccess) { rc = hip_fail(e, "sync"); break; }
    } while (0);
    (void)hipStreamSynchronize(s);
    for (int k = 0; k < 8; ++k) if (bufs[k]) (void)hipFree(bufs[k]);
    return rc;
}

}  // extern "C"
