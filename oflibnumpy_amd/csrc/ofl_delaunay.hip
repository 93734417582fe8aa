// ofl_delaunay.hip -- K3 exact path: a real Delaunay triangulation of the warped points on the GPU.
//
// Replaces scipy.interpolate.griddata(points, values, grid, 'linear') (src/oflibnumpy/utils.py:253) for every field
// whose cell-wise mesh is NOT provably the Delaunay triangulation (ofl_scatter_walk.hip certifies the others):
// folded meshes (motion boundaries, the tiled Sintel field of BASELINE config 5 as loaded), dropped points (holes and
// speckle of the point mask), curved mesh borders (pockets between the mesh and its convex hull), sheared cells.
//
//   bin      the kept points are counting-sorted into a bucket grid over their bounding box (~1 point per bucket); exact
//            duplicates of a site are dropped (the smallest index of a location stays)
//   stars    four passes, cheapest first (ofl_delaunay_core.h holds the geometry, shared with the CPU test build):
//            mesh fans   one thread per point VERIFIES the star the warped grid proposes (empty circumcircles against the
//                        sites of the buckets under them) -- the interior of every smooth piece of the field;
//            clip        one thread per remaining point builds its Voronoi cell by half-plane clipping in growing bucket
//                        rings (in-circle decisions, security radius);
//            rims        cells that are not final within the rings -- rims of tears and holes -- against a coarse grid of
//                        the unfinished points only: per thread when there are tens of thousands, one wave per point else;
//            hull        what is still left (unbounded cells) by one workgroup per point against all other left-over
//                        points (every Delaunay neighbour of an unfinished point beyond the ring search is itself unfinished)
//   raster   every star's triangles (p, n_k, n_k+1) are scan-converted with SciPy's inclusion rule; atomicMin keeps
//            the smallest triangle id per node -- each triangle is emitted by all three of its sites, so stars that
//            disagree on an exactly co-circular cell (where Qhull itself is arbitrary) still tile the hull
//   resolve  float64 barycentric interpolation from the owner triangle, vertices in canonical order so that all
//            copies of a triangle give the same bits; nodes without an owner are outside the convex hull (NaN -> 0)
//
// Everything is a pure function of the inputs (buckets are sorted by point index, far points are compacted in
// index order, triangle ids derive from point indices), so row bands concatenate to the full result bit for bit.
#include "ofl_scatter_dev.h"
#ifdef OFL_EXPERIMENTS
__device__ unsigned long long dl_dbg_counters[8];      // development counters of the geometry core (experiments build, OFL_DL_DEBUG prints them)
#define DL_DBG(i, v) atomicAdd(&dl_dbg_counters[i], (unsigned long long)(v))
#endif
#include "ofl_delaunay_core.h"
#include <algorithm>
#include <limits>
#include <stddef.h>
#include <stdlib.h>
#include <vector>

using namespace ofl;
using namespace ofl_sc;
using namespace ofl_dl;

namespace {

constexpr int      kRings    = 6;        // bucket rings of the per-thread star pass
constexpr int      kOpenRings = 2;       // ... of which a cell still unbounded after this many is handed on at once
#ifndef OFL_NEAR_CAP
#define OFL_NEAR_CAP 10
#endif
constexpr int      kNearCap  = OFL_NEAR_CAP;       // polygon capacity of the per-thread pass (float32 cell in LDS)
constexpr int      kSlots    = 16;       // neighbour slots per point
constexpr int      kNear2Rings = 6;      // coarse rings of the second per-thread pass ...
constexpr unsigned kNear2MinPoints = 32768;   // unfinished points below which the second per-thread pass is skipped
constexpr int      kNear2Open = 3;       // ... which gives up a cell that is still unbounded after this many
constexpr int      kMidCap   = 256;      // polygon capacity of the wave pass (unfinished points against the coarse grid of unfinished points)
constexpr int      kMidRings = 8;        // rings of that coarse grid every unfinished point is given ...
constexpr int      kMidRingsMax = 160;   // ... and the rings a BOUNDED cell may go on for (rims of large holes) before it is left to the workgroup pass
constexpr double   kFarBuckets = 256.0;  // far threshold of the cooperative cells, in buckets (dl_params_kernel)
constexpr int      kFarList  = 48;       // cell vertices beyond the data (unbounded directions, sliver fans of a straight border) tested one by one
constexpr int      kMidScale = 8;        // ... whose cells are this many fine buckets wide
constexpr int      kFarCap   = 2560;     // polygon capacity of the workgroup pass (LDS: 20 B per vertex)
constexpr unsigned kDegLeft  = 0xFFFFFFFFu;   // far_deg marker: not finished by the wave pass
constexpr unsigned kFarK     = 4096;     // triangle-id stride of a far point (>= kFarCap)
constexpr unsigned kNoOwner  = 0xFFFFFFFFu;
constexpr int      kSmallArea = 1024;
constexpr int      kLocateRings = 2;     // bucket rings dl_locate_brute searches when the visibility walk got stuck
constexpr unsigned char kDegFar = 0xFF;
constexpr unsigned char kDegTodo = 0xFE;  // not settled by the mesh-fan pass: the clip pass builds this star
constexpr unsigned char kDegFan = 0xFD;   // not settled by the mesh-cell pass: the fan pass looks at this site
constexpr unsigned char kDegHeavy = 0xFC; // the clip pass met a dense cluster around this site: the heavy-bucket launch of the clip pass builds this star
constexpr int      kFanSpan  = 6;        // buckets per axis a fan's circumcircles may span (wider: clip pass)
constexpr int      kFanBlock = 128;
constexpr unsigned kDedupeSmall = 64;     // buckets up to this size drop their duplicates by pairwise comparison (quadratic, one thread), larger ones through a hash table
constexpr unsigned kMaxFar   = 1u << 20;  // unfinished stars ...
constexpr unsigned kMaxLeft  = 1u << 18;  // ... and stars for the workgroup pass (quadratic in their number) before the call gives up
constexpr unsigned kErrDegenerate = 16u;  // err bit: one of the three limits above -- Qhull, too, refuses such input ("initial simplex is flat")
constexpr unsigned kErrSlabList = 32u;    // err bit (slab mode): a rank's list of unfinished sites did not fit the buffer the caller gave it, or is not a list of this field
constexpr int      kSlabMargin = 2 * (kRings + 1) + 2;   // buckets around a row band within which the slab mode builds every star (see exact_stars)
constexpr int      kScanChunk = 2048;    // elements per block of the scan kernels (256 threads x 8)

struct DlHead {                           // device header of the exact path (256 bytes)
    unsigned long long kx0, kx1, ky0, ky1;   // ordered keys of the bounding box while it is reduced
    Grid     grid, grid1;                    // fine buckets (all kept points), coarse buckets (unfinished points)
    unsigned kept, n_far, n_left, pool_used, err;    // err bit 0: far polygon overflow, bit 1: pool overflow, bit 2: big list overflow, bit 3: triangle-id space
    unsigned long long big_n;
    unsigned n_todo;                                 // points the mesh-fan pass left to the clip pass (counted in debug runs only)
    unsigned n_fan;                                  // points the mesh-cell pass left to the fan pass
    double   far_t2;                                 // squared distance beyond which a cell vertex counts as "far" (well outside the data)
    unsigned n_big;                                  // sorted entries that live in buckets of more than kDedupeSmall entries
    unsigned any_heavy;                              // some bucket holds more than kHeavy entries (else the heavy-bucket passes return at once)
    double   need_lo, need_hi;                       // slab mode: only sites with need_lo <= y <= need_hi get a star from cells / fans / clip (else -inf, +inf)
    unsigned slab_stamp;                             // slab mode: slab_stamp_of(field, band) step 1 ran for (0: not a slab state)
    unsigned sample_all, sample_ok;                  // dl_cell_sample_kernel: sampled grid cells with four kept corners / of those, verified
    unsigned n_raster;                               // sites the site-wise raster looks at (stars of 1 .. kSlots neighbours outside the clean tiles)
    unsigned n_heavy;                                // buckets of more than kHeavy entries (dense clusters): they get grids of their own
    unsigned sub_used;                               // words of DlWs::sub_start handed out to those grids
    unsigned dbg[8];                                 // experiments build: counters of the left-over pass (sites, chunks swept, steps, clips, seed/near/coarse clips)
};
static_assert(sizeof(DlHead) <= 256, "DlHead");
static_assert(offsetof(DlHead, slab_stamp) == ofl_sc::kSlabStampAt, "ofl_scatter_dev.h: kSlabStampAt");

struct DlWs {
    DlHead   *head;
    unsigned *bstart;      // [bcap + 1] counts -> exclusive starts
    unsigned *scan_tmp;    // block sums of the scan levels
    unsigned *sorted;      // [N] point indices bucket by bucket
    unsigned char *deg;    // [N] 0 .. 16, kDegTodo, kDegFar
    unsigned *todo_idx;    // [N] points for the fan pass (what the cell pass did not settle), ascending
    unsigned char *cellflag; // [N] per grid cell (x, y): 0 = not verified, 1 / 2 = both triangles Delaunay, diagonal a - c / b - d (ofl_dl::cell_verify)
    unsigned char *tileflag; // [ceil(W / 32) * ceil(H / 8)] 1 = every cell of the 32 x 8 tile is verified (a CLEAN tile)
    unsigned char *dup;    // [N] 1 = an exact duplicate of a site with a smaller index (not a site of the triangulation)
    unsigned *nbr;         // [N][kSlots]
    unsigned *far_idx;     // [N]
    unsigned *far_deg;     // [N]
    unsigned *far_off;     // [N]
    unsigned char *far_wide; // [N] 1 = finished by the wave pass BEYOND the coarse rings the workgroup pass searches: stays a candidate of its sweeps
    unsigned *left_idx;    // [N] ranks of the points the wave pass could not finish
    unsigned *b1start;     // [b1cap + 1] coarse buckets of the unfinished points
    unsigned *b1cursor;    // [b1cap]
    unsigned *sorted1;     // [N] their ranks bucket by bucket
    P2       *sorted_xy;   // [N] positions in `sorted` order
    P2       *sorted1_xy;  // [N] positions in `sorted1` order
    unsigned *sorted1_pt;  // [N] point indices in `sorted1` order
    P2       *left_xy;     // [N] positions of the left-over points, in left_idx order
    unsigned *left_pt;     // [N] their point indices
    double   *left_box;    // [N / 256 + 1][2][8] oriented boxes of the two image halves of 256 consecutive left-over points
    int      *pool;        // [pool_cap] neighbour lists of the far points (negative: unbounded gap)
    unsigned *cstate;      // [6][cstride] ticket + tile words of the ordered compactions (dl_compact_kernel)
    unsigned *heavy_bucket; // [heavy_cap] numbers of the buckets of more than kHeavy entries, ascending
    SubGrid  *heavy_info;  // [heavy_cap] their grids
    unsigned *sub_start;   // [sub_cap] cell starts of those grids (absolute positions in `sorted`)
    unsigned *big;         // [big_cap] triangle ids with a large bounding box
    uint32_t *owner;       // [H][W] (biased by the first row of the band)
    size_t    bcap, b1cap, pool_cap, big_cap, cstride, heavy_cap, sub_cap;
    int       oy0, oy1;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t scan_tmp_elems(size_t n)
{
    size_t t = 0;
    while (n > 8192) { n = (n + 8191) / 8192; t += (n + 63) / 64 * 64; }
    return t + 64;
}

// device-side helpers -------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long okey(double d)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double okey_inv(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
    return __longlong_as_double((long long)b);
}

struct PosFn {
    const float *flow; int sign, W; float inv_w;
    __device__ __forceinline__ PosFn(const float *f, int s, int w) : flow(f), sign(s), W(w), inv_w(1.0f / (float)w) {}
    __device__ __forceinline__ P2 operator()(int i) const
    {
        // row of point i without an integer division: float estimate (i < 2^27, error < 16 rows... corrected exactly)
        int y = (int)((float)i * inv_w);
        int r = i - y * W;
        while (r < 0) { --y; r += W; }
        while (r >= W) { ++y; r -= W; }
        const D2 p = point_of(flow, sign, W, r, y);
        return P2{ p.x, p.y };
    }
};

__device__ __forceinline__ bool kept_pt(const uint8_t *pmask, size_t i) { return !pmask || pmask[i] != 0; }
// a point with a NaN / Inf position cannot be a site (the reference's Flow refuses such vectors; behind the bare C ABI they
// are dropped like masked-out points instead of sending bucket indices out of range)
__device__ __forceinline__ bool finite_pt(double x, double y) { return isfinite(x) && isfinite(y); }

// exclusive scan of one value per thread over a 256-thread block; `total` = block sum (valid in all threads)
__device__ __forceinline__ unsigned block_exscan(unsigned v, unsigned &total)
{
    __shared__ unsigned s_w[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += t;
    }
    __syncthreads();
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int k = 0; k < w; ++k) base += s_w[k];
    total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------------ binning
__global__ __launch_bounds__(256)
void dl_bbox_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                    unsigned long long *__restrict__ partial)      // [workgroups][5]: keys of min x, max x, min y, max y; points kept
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    unsigned cnt = 0;
    if (x < W)
        for (int y = blockIdx.y * 8 + (threadIdx.x >> 5); y < H; y += gridDim.y * 8) {
            if (!kept_pt(pmask, (size_t)y * W + x)) continue;
            const D2 p = point_of(flow, sign, W, x, y);
            if (!finite_pt(p.x, p.y)) continue;
            x0 = fmin(x0, p.x); x1 = fmax(x1, p.x); y0 = fmin(y0, p.y); y1 = fmax(y1, p.y);
            ++cnt;
        }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x0 = fmin(x0, __hiloint2double(__shfl_xor(__double2hiint(x0), off), __shfl_xor(__double2loint(x0), off)));
        x1 = fmax(x1, __hiloint2double(__shfl_xor(__double2hiint(x1), off), __shfl_xor(__double2loint(x1), off)));
        y0 = fmin(y0, __hiloint2double(__shfl_xor(__double2hiint(y0), off), __shfl_xor(__double2loint(y0), off)));
        y1 = fmax(y1, __hiloint2double(__shfl_xor(__double2hiint(y1), off), __shfl_xor(__double2loint(y1), off)));
        cnt += (unsigned)__shfl_xor((int)cnt, off);
    }
    // One record per workgroup, reduced by dl_params_kernel.  (Atomics on the header -- five per workgroup, then one set per
    // workgroup -- are executed one after the other at the memory side: a thousand workgroups spent 50 of this kernel's 67 us
    // at 4K queueing for ONE cache line, and twice the workgroups took twice as long.)
    __shared__ double s_b[4][4];
    __shared__ unsigned s_c[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_b[0][w] = x0; s_b[1][w] = x1; s_b[2][w] = y0; s_b[3][w] = y1; s_c[w] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned c = 0;
        for (int k = 0; k < 4; ++k) { x0 = fmin(x0, s_b[0][k]); x1 = fmax(x1, s_b[1][k]); y0 = fmin(y0, s_b[2][k]); y1 = fmax(y1, s_b[3][k]); c += s_c[k]; }
        unsigned long long *rec = partial + 5 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
        rec[0] = c ? okey(x0) : ~0ull; rec[1] = c ? okey(x1) : 0ull; rec[2] = c ? okey(y0) : ~0ull; rec[3] = c ? okey(y1) : 0ull; rec[4] = c;
    }
}

// what step 1 of the slab mode leaves in the header for step 2 to recognise: the band and the field it ran for (never 0)
__host__ __device__ inline unsigned slab_stamp_of(int H, int W, int row0, int rows)
{
    const unsigned v = ((unsigned)row0 * 65536u + (unsigned)rows) ^ ((unsigned)H * 2654435761u) ^ ((unsigned)W * 40503u);
    return v ? v : 1u;
}

__global__ __launch_bounds__(256)
void dl_params_kernel(DlHead *head, unsigned long long bcap, double bucket_scale, int H, int W, int slab, int row0, int rows,
                      const unsigned long long *__restrict__ partial, unsigned n_partial)
{
    {   // the bounding box and the number of kept points from dl_bbox_kernel's records
        unsigned long long k0 = ~0ull, k1 = 0ull, k2 = ~0ull, k3 = 0ull, c = 0ull;
        for (unsigned i = threadIdx.x; i < n_partial; i += 256) {
            const unsigned long long *rec = partial + 5 * (size_t)i;
            k0 = rec[0] < k0 ? rec[0] : k0; k1 = rec[1] > k1 ? rec[1] : k1;
            k2 = rec[2] < k2 ? rec[2] : k2; k3 = rec[3] > k3 ? rec[3] : k3; c += rec[4];
        }
        __shared__ unsigned long long s_k[5][256];
        s_k[0][threadIdx.x] = k0; s_k[1][threadIdx.x] = k1; s_k[2][threadIdx.x] = k2; s_k[3][threadIdx.x] = k3; s_k[4][threadIdx.x] = c;
        __syncthreads();
        for (int half = 128; half > 0; half >>= 1) {
            if ((int)threadIdx.x < half) {
                const int t = threadIdx.x, u = t + half;
                s_k[0][t] = s_k[0][u] < s_k[0][t] ? s_k[0][u] : s_k[0][t]; s_k[1][t] = s_k[1][u] > s_k[1][t] ? s_k[1][u] : s_k[1][t];
                s_k[2][t] = s_k[2][u] < s_k[2][t] ? s_k[2][u] : s_k[2][t]; s_k[3][t] = s_k[3][u] > s_k[3][t] ? s_k[3][u] : s_k[3][t];
                s_k[4][t] += s_k[4][u];
            }
            __syncthreads();
        }
        if (threadIdx.x != 0) return;
        k0 = s_k[0][0]; k1 = s_k[1][0]; k2 = s_k[2][0]; k3 = s_k[3][0]; c = s_k[4][0];
        head->kx0 = k0; head->kx1 = k1; head->ky0 = k2; head->ky1 = k3; head->kept = (unsigned)c;
    }
    Grid g;
    g.ox = 0.0; g.oy = 0.0; g.s = 1.0; g.inv_s = 1.0; g.gx = 1; g.gy = 1;
    const unsigned n = head->kept;
    if (n > 0) {
        double x0 = okey_inv(head->kx0), x1 = okey_inv(head->kx1), y0 = okey_inv(head->ky0), y1 = okey_inv(head->ky1);
        // The grid covers at most the image and a margin of W + H around it: a handful of garbage vectors (1e9 marks
        // "unknown" in some flow files) must not blow the buckets up to thousands of sites each.  Sites beyond it fall
        // into the border buckets (Grid::bx / by clamp) -- seen from any site inside, such a site is at least as far away
        // as its bucket, which is all the ring searches rely on; the outliers themselves never close within the rings and
        // end in the passes that run until the candidates are exhausted.
        const double m = (double)W + (double)H;
        x0 = fmax(x0, fmin(-m, x1 - 1.0)); x1 = fmin(x1, fmax((double)W + m, x0 + 1.0));
        y0 = fmax(y0, fmin(-m, y1 - 1.0)); y1 = fmin(y1, fmax((double)H + m, y0 + 1.0));
        const double bw = x1 - x0, bh = y1 - y0;
        // every point on one spot or on one axis-parallel line: nothing to triangulate (Qhull: "initial simplex is flat")
        if (n >= 3 && (!(okey_inv(head->kx1) > okey_inv(head->kx0)) || !(okey_inv(head->ky1) > okey_inv(head->ky0)))) atomicOr(&head->err, kErrDegenerate);
        double s = bucket_scale * sqrt(fmax(bw * bh, 1e-300) / (double)n);             // ~bucket_scale^2 points per bucket
        s = fmax(s, (bw + bh) / (double)n);
        if (!(s > 0.0) || !isfinite(s)) s = 1.0;
        for (int it = 0; it < 64; ++it) {
            const double fx = floor(bw / s) + 1.0, fy = floor(bh / s) + 1.0;
            if (fx * fy <= (double)bcap && fx < 2e9 && fy < 2e9) break;
            s *= 1.5;
        }
        g.ox = x0; g.oy = y0; g.s = s; g.inv_s = 1.0 / s;
        g.gx = (int)(floor(bw / s) + 1.0); g.gy = (int)(floor(bh / s) + 1.0);
        // Cell vertices farther than this from their site count as FAR: they are excluded by the cone they lie in rather than by
        // the reach of the near ones.  A straight border that float32 leaves 1e-5 px rough has sliver triangles with
        // circumcentres 1e4 px out, a wavy one (3 px over 100 px) 400 - 4 000 px out: as near vertices their reach covers every
        // candidate of the left-over sweep.  kFarBuckets point spacings (never more than the data's own extent) is where the
        // rim of a hole -- whose cells do reach that far, in all directions -- still pays nothing for it.
        const double t = fmin(kFarBuckets * s, 1.5 * (bw + bh) + 8.0 * s);
        head->far_t2 = t * t;
    } else head->far_t2 = 1.0;
    head->grid = g;
    // slab mode: stars from cells / fans / clip only within kSlabMargin buckets of the band's rows (the first and the last
    // band of a field are open-ended: sites warp beyond the frame)
    const double inf = __longlong_as_double(0x7FF0000000000000ll), m = (double)kSlabMargin * g.s;
    head->need_lo = (!slab || row0 <= 0) ? -inf : (double)row0 - m;
    head->need_hi = (!slab || row0 + rows >= H) ? inf : (double)(row0 + rows - 1) + m;
    // (sites beyond the grid are clamped into its border buckets: a star next to those may hold a neighbour that is "within
    // the rings" by bucket number only -- a slab that comes this close to the border rows takes everything beyond them, too)
    if (head->need_lo < g.oy + (double)(kRings + 2) * g.s) head->need_lo = -inf;
    if (head->need_hi > g.oy + (double)(g.gy - kRings - 2) * g.s) head->need_hi = inf;
    head->slab_stamp = slab ? slab_stamp_of(H, W, row0, rows) : 0u;
    Grid g1 = g;                                     // coarse grid of the unfinished points: kMidScale fine buckets per cell
    g1.s = g.s * kMidScale; g1.inv_s = 1.0 / g1.s;
    g1.gx = (g.gx + kMidScale - 1) / kMidScale; g1.gy = (g.gy + kMidScale - 1) / kMidScale;
    head->grid1 = g1;
}

// the same three binning steps for the unfinished points (ranks into far_idx) on the coarse grid
__global__ __launch_bounds__(256)
void dl_count1_kernel(const float *__restrict__ flow, int sign, int W, const DlHead *__restrict__ head,
                      const unsigned *__restrict__ far_idx, unsigned *__restrict__ bcount)
{
    const Grid g = head->grid1;
    for (unsigned r = blockIdx.x * 256 + threadIdx.x; r < head->n_far; r += gridDim.x * 256) {      // (sized on the device: no read-back of the count)
        const P2 p = PosFn(flow, sign, W)((int)far_idx[r]);
        atomicAdd(&bcount[(size_t)g.by(p.y) * g.gx + g.bx(p.x)], 1u);
    }
}

__global__ __launch_bounds__(256)
void dl_fill1_kernel(const float *__restrict__ flow, int sign, int W, const DlHead *__restrict__ head,
                     const unsigned *__restrict__ far_idx, const unsigned *__restrict__ bstart,
                     unsigned *__restrict__ cursor, unsigned *__restrict__ sorted)
{
    const Grid g = head->grid1;
    for (unsigned r = blockIdx.x * 256 + threadIdx.x; r < head->n_far; r += gridDim.x * 256) {
        const P2 p = PosFn(flow, sign, W)((int)far_idx[r]);
        const size_t b = (size_t)g.by(p.y) * g.gx + g.bx(p.x);
        sorted[bstart[b] + atomicAdd(&cursor[b], 1u)] = r;
    }
}

__global__ __launch_bounds__(256)
void dl_count_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                     const DlHead *__restrict__ head, unsigned *__restrict__ bcount, unsigned char *__restrict__ dup,
                     unsigned *__restrict__ slot)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * W || !kept_pt(pmask, i)) return;
    const Grid g = head->grid;
    const P2 p = PosFn(flow, sign, W)((int)i);
    if (!finite_pt(p.x, p.y)) { dup[i] = 1; return; }      // not a site (flagged like a dropped duplicate)
    // the count a site draws is its slot in the bucket (any order will do: the buckets are sorted by index afterwards), so the
    // fill pass needs neither a second atomic per site nor a zeroed array of bucket cursors (265 MB at 8K)
    slot[i] = atomicAdd(&bcount[(size_t)g.by(p.y) * g.gx + g.bx(p.x)], 1u);
}

// exclusive scan in three launches: sums of 8192-element chunks, ONE workgroup scans up to 8192 of those sums in place,
// then every chunk is scanned with its offset (round 2 walked 2048-element levels: seven launches for the bucket counts of
// a 4K field, five for every compaction -- a dozen scans per call made 42 tiny launches)
constexpr int kScanBig = 8192;             // elements per block: 256 threads x 32

__global__ __launch_bounds__(256)
void dl_scan_reduce_kernel(const unsigned *__restrict__ in, size_t n, unsigned *__restrict__ sums)
{
    const size_t base = (size_t)blockIdx.x * kScanBig + (size_t)threadIdx.x * 32;
    unsigned v = 0;
    if (base + 32 <= n) {
        const uint4 *q = reinterpret_cast<const uint4 *>(in + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const uint4 t = q[k]; v += t.x + t.y + t.z + t.w; }
    } else {
        for (int k = 0; k < 32; ++k) if (base + k < n) v += in[base + k];
    }
    unsigned total;
    (void)block_exscan(v, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256)
void dl_scan_apply_kernel(unsigned *__restrict__ data, size_t n, const unsigned *__restrict__ offsets)
{
    const size_t base = (size_t)blockIdx.x * kScanBig + (size_t)threadIdx.x * 32;
    unsigned e[32], v = 0;
    const bool whole = base + 32 <= n;
    if (whole) {
        const uint4 *q = reinterpret_cast<const uint4 *>(data + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const uint4 t = q[k]; e[4 * k] = t.x; e[4 * k + 1] = t.y; e[4 * k + 2] = t.z; e[4 * k + 3] = t.w; }
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) e[k] = base + k < n ? data[base + k] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) v += e[k];
    unsigned total;
    unsigned run = block_exscan(v, total) + (offsets ? offsets[blockIdx.x] : 0u);
    if (whole) {
        uint4 *q = reinterpret_cast<uint4 *>(data + base);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint4 t;
            t.x = run; run += e[4 * k]; t.y = run; run += e[4 * k + 1]; t.z = run; run += e[4 * k + 2]; t.w = run; run += e[4 * k + 3];
            q[k] = t;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) { if (base + k < n) data[base + k] = run; run += e[k]; }
    }
}

// one workgroup of 1024: in-place exclusive scan of up to 8192 values
__global__ __launch_bounds__(1024)
void dl_scan_small_kernel(unsigned *__restrict__ data, unsigned n)
{
    __shared__ unsigned s_w[16];
    const unsigned base = threadIdx.x * 8;
    unsigned e[8], v = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { e[k] = base + k < n ? data[base + k] : 0u; v += e[k]; }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned t = (unsigned)__shfl_up((int)inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    unsigned run = inc - v;
    for (int k = 0; k < w; ++k) run += s_w[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) { if (base + k < n) data[base + k] = run; run += e[k]; }
}

__global__ __launch_bounds__(256)
void dl_fill_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                    const DlHead *__restrict__ head, const unsigned *__restrict__ bstart, const unsigned *__restrict__ slot,
                    unsigned *__restrict__ sorted, const unsigned char *__restrict__ dup)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)H * W || !kept_pt(pmask, i) || dup[i]) return;
    const Grid g = head->grid;
    const P2 p = PosFn(flow, sign, W)((int)i);
    const size_t b = (size_t)g.by(p.y) * g.gx + g.bx(p.x);
    sorted[bstart[b] + slot[i]] = (unsigned)i;
}

// ascending point index inside every bucket (the fill order is not deterministic)
__global__ __launch_bounds__(256)
void dl_sort_kernel(const DlHead *__restrict__ head, int coarse, const unsigned *__restrict__ bstart, unsigned *__restrict__ sorted)
{
    const size_t nb = coarse ? (size_t)head->grid1.gx * head->grid1.gy : (size_t)head->grid.gx * head->grid.gy;
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nb; b += (size_t)gridDim.x * 256) {
        const unsigned lo = bstart[b], hi = bstart[b + 1];
        // (a heavy bucket of the fine grid keeps its fill order here: dl_sub_bin_kernel re-orders it cell by cell, every cell by
        // index, whatever order it finds; a coarse bucket of more than 256 unfinished sites keeps its fill order)
        if (hi - lo < 2 || hi - lo > (coarse ? 256u : kHeavy)) continue;
        for (unsigned i = lo + 1; i < hi; ++i) {
            const unsigned v = sorted[i];
            unsigned j = i;
            while (j > lo && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; --j; }
            sorted[j] = v;
        }
    }
}

// positions (and point indices) in the order of a list: MODE 0 `list` holds point indices, 1 ranks into far_idx, 2 indices
// into left_idx (ranks of ranks)
template <int MODE>
__global__ __launch_bounds__(256)
void dl_list_xy_kernel(const float *__restrict__ flow, int sign, int W, const DlHead *__restrict__ head,
                       const unsigned *__restrict__ list, const unsigned *__restrict__ far_idx,
                       P2 *__restrict__ xy, unsigned *__restrict__ pt)
{
    const unsigned n = MODE == 0 ? head->kept : (MODE == 1 ? head->n_far : head->n_left);
    const PosFn pos(flow, sign, W);
    for (unsigned j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const unsigned i = MODE == 0 ? list[j] : far_idx[list[j]];
        xy[j] = pos((int)i);
        if (pt) pt[j] = i;
    }
}

// Bounds of every 256 consecutive left-over points (one workgroup each): the sweeps of the workgroup pass skip the chunks
// that cannot hold a cutting site.  Left-over points are mostly image-border points in index order -- the top row, then the
// left and right columns in turns, then the bottom row -- so a chunk is split by image half (x < W / 2 or not) and each
// half gets an ORIENTED box along the line from its first to its last point: a piece of one border side is a thin sliver
// whatever the rotation of the field (an axis-aligned box of a slanted side holds the whole hull).
// box[chunk][half] = { ox, oy, ax, ay, t0, t1, s0, s1 }: origin, unit axis, ranges along the axis and along (-ay, ax);
// t0 > t1: the half is empty.
__global__ __launch_bounds__(256)
void dl_left_box_kernel(const DlHead *__restrict__ head, int W, const unsigned *__restrict__ left_pt,
                        const P2 *__restrict__ left_xy, double *__restrict__ box)
{
    __shared__ unsigned s_first[2], s_last[2];
    __shared__ unsigned long long s_k[2][4];               // ordered keys of t0 (min), t1 (max), s0 (min), s1 (max)
    const unsigned n = head->n_left;
    const int t = threadIdx.x;
    for (unsigned chunk = blockIdx.x; chunk * 256u < n; chunk += gridDim.x) {
        __syncthreads();
        if (t < 2) { s_first[t] = 0xFFFFFFFFu; s_last[t] = 0u; s_k[t][0] = ~0ull; s_k[t][1] = 0ull; s_k[t][2] = ~0ull; s_k[t][3] = 0ull; }
        __syncthreads();
        const unsigned j = chunk * 256 + t;
        const bool in = j < n;
        P2 q{ 0.0, 0.0 };
        int half = 0;
        if (in) {
            q = left_xy[j];
            half = (int)(left_pt[j] % (unsigned)W) * 2 >= W ? 1 : 0;
            atomicMin(&s_first[half], j);
            atomicMax(&s_last[half], j);
        }
        __syncthreads();
        double ax = 1.0, ay = 0.0, ox = 0.0, oy = 0.0;
        if (in) {
            const P2 o = left_xy[s_first[half]], e = left_xy[s_last[half]];
            const double dx = e.x - o.x, dy = e.y - o.y, len = sqrt(dx * dx + dy * dy);
            ox = o.x; oy = o.y;
            if (len > 0.0 && isfinite(len)) { ax = dx / len; ay = dy / len; }
            const double rx = q.x - ox, ry = q.y - oy, tt = rx * ax + ry * ay, ss = ry * ax - rx * ay;
            atomicMin(&s_k[half][0], okey(tt)); atomicMax(&s_k[half][1], okey(tt));
            atomicMin(&s_k[half][2], okey(ss)); atomicMax(&s_k[half][3], okey(ss));
        }
        __syncthreads();
        if (t < 2) {
            double *b = box + ((size_t)chunk * 2 + t) * 8;
            if (s_first[t] == 0xFFFFFFFFu) { b[4] = 1.0; b[5] = 0.0; }          // empty half
            else {
                const P2 o = left_xy[s_first[t]], e = left_xy[s_last[t]];
                const double dx = e.x - o.x, dy = e.y - o.y, len = sqrt(dx * dx + dy * dy);
                double axx = 1.0, ayy = 0.0;
                if (len > 0.0 && isfinite(len)) { axx = dx / len; ayy = dy / len; }       // (the same expressions as above: the same axis)
                b[0] = o.x; b[1] = o.y; b[2] = axx; b[3] = ayy;
                b[4] = okey_inv(s_k[t][0]); b[5] = okey_inv(s_k[t][1]); b[6] = okey_inv(s_k[t][2]); b[7] = okey_inv(s_k[t][3]);
            }
        }
    }
}

// Exact duplicates (folded integer-valued fields: BASELINE config 5 puts several sites on most lattice nodes): Qhull keeps
// ONE vertex per location and which one is its own business, so only the smallest index of a location stays a site here.
// The others are flagged, and their entries in the bucket list are blanked (index 0xFFFFFFFF: every consumer skips
// negative candidates) -- a field with five sheets builds a fifth of the stars.
__global__ __launch_bounds__(256)
void dl_dedupe_kernel(DlHead *__restrict__ head, const unsigned *__restrict__ bstart, unsigned *__restrict__ sorted,
                      const P2 *__restrict__ sorted_xy, unsigned char *__restrict__ dup)
{
    const size_t nb = (size_t)head->grid.gx * head->grid.gy;
    for (size_t b = (size_t)blockIdx.x * 256 + threadIdx.x; b < nb; b += (size_t)gridDim.x * 256) {
        const unsigned lo = bstart[b], hi = bstart[b + 1];
        if (hi - lo > kDedupeSmall) atomicAdd(&head->n_big, hi - lo);     // rare: dl_big_* below take these buckets
        if (hi - lo > kHeavy) head->any_heavy = 1u;                       // rare: dl_sub_bin_kernel gives these buckets grids of their own
        if (hi - lo < 2 || hi - lo > kDedupeSmall) continue;
        for (unsigned j = lo + 1; j < hi; ++j) {
            const P2 q = sorted_xy[j];
            bool same = false;
            for (unsigned i = lo; i < j && !same; ++i) same = sorted[i] != 0xFFFFFFFFu && sorted_xy[i].x == q.x && sorted_xy[i].y == q.y;
            if (same) { dup[sorted[j]] = 1; sorted[j] = 0xFFFFFFFFu; }
        }
    }
}

// Buckets of more than kDedupeSmall entries -- a flow that collapses a whole block of the image onto one pixel is legal
// input to Flow.apply, and Qhull (option Qc) treats coincident points as ONE vertex -- drop their duplicates through an
// open-addressing table of point indices keyed by the position's bits (`table`: T = 2^k >= 2 n_big entries of the
// neighbour pool, which nothing uses yet): insertion keeps the SMALLEST index of every location, a second pass over the
// entries blanks the others.  All three kernels return at once when no such bucket exists (n_big == 0).
__device__ __forceinline__ unsigned dl_big_size(const DlHead *head, unsigned long long cap)
{
    unsigned long long t = 1024;
    while (t < 2ull * head->n_big && 2 * t <= cap) t <<= 1;   // a power of two (cap = pool entries >= 8 n > 4 n_big: never binding)
    return (unsigned)t;
}

__device__ __forceinline__ unsigned dl_pos_hash(const P2 &q)
{
    unsigned long long a = (unsigned long long)__double_as_longlong(q.x), b = (unsigned long long)__double_as_longlong(q.y);
    a ^= a >> 33; a *= 0xff51afd7ed558ccdull; a ^= a >> 33;
    b ^= b >> 29; b *= 0xc4ceb9fe1a85ec53ull; b ^= b >> 32;
    a ^= b + 0x9e3779b97f4a7c15ull + (a << 6) + (a >> 2);
    return (unsigned)(a ^ (a >> 32));
}

__global__ __launch_bounds__(256)
void dl_big_clear_kernel(const DlHead *__restrict__ head, unsigned *__restrict__ table, unsigned long long cap)
{
    if (head->n_big == 0) return;
    const unsigned T = dl_big_size(head, cap);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) table[i] = 0xFFFFFFFFu;
}

// PASS 0: insert (smallest index of a location wins), PASS 1: blank every entry that is not its location's smallest index
template <int PASS>
__global__ __launch_bounds__(256)
void dl_big_kernel(DlHead *__restrict__ head, const unsigned *__restrict__ bstart, unsigned *__restrict__ sorted,
                   const P2 *__restrict__ sorted_xy, unsigned char *__restrict__ dup, unsigned *__restrict__ table,
                   unsigned long long cap, size_t n_entries, const float *__restrict__ flow, int sign, int W)
{
    if (head->n_big == 0) return;
    const unsigned T = dl_big_size(head, cap), mask = T - 1;
    const Grid g = head->grid;
    const PosFn pos(flow, sign, W);
    n_entries = bstart[(size_t)g.gx * g.gy];                   // the filled part of `sorted`
    for (size_t j = (size_t)blockIdx.x * 256 + threadIdx.x; j < n_entries; j += (size_t)gridDim.x * 256) {
        const unsigned c = sorted[j];
        if (c == 0xFFFFFFFFu) continue;
        const P2 q = sorted_xy[j];
        const size_t b = (size_t)g.by(q.y) * g.gx + g.bx(q.x);
        const unsigned cnt = bstart[b + 1] - bstart[b];
        if (cnt <= kDedupeSmall) continue;
        unsigned h = dl_pos_hash(q) & mask;
        for (unsigned probe = 0; probe < T; ++probe, h = (h + 1) & mask) {
            if (PASS == 0) {
                const unsigned cur = atomicCAS(&table[h], 0xFFFFFFFFu, c);
                if (cur == 0xFFFFFFFFu) break;                           // a new location
                const P2 o = pos((int)cur);                              // (whatever index the slot holds now or later: the same location)
                if (o.x == q.x && o.y == q.y) { atomicMin(&table[h], c); break; }
            } else {
                const unsigned cur = table[h];
                if (cur == 0xFFFFFFFFu) break;                           // (cannot happen: every entry was inserted)
                const P2 o = pos((int)cur);
                if (o.x == q.x && o.y == q.y) {
                    if (cur != c) { dup[c] = 1; sorted[j] = 0xFFFFFFFFu; }
                    break;
                }
            }
        }
    }
}

// A grid of its own for every heavy bucket (ofl_dl::SubGrid; the list of their numbers comes from dl_compact_kernel<5>): one
// workgroup per bucket measures the bounding box of the bucket's entries, lays K x K cells over it -- K = ceil(sqrt(m / 2)),
// at most 256 -- counts, scans and re-orders the bucket's stretch of `sorted` / `sorted_xy` cell by cell (through the scratch
// arrays: entries move within [lo, hi) only) and sorts every cell by point index, so that the order of a cluster's sites --
// and with it the order in which every star meets its neighbours -- does not depend on the order the fill pass happened to
// leave (buckets of more than 256 entries keep their fill order in dl_sort_kernel).
__global__ __launch_bounds__(256)
void dl_sub_bin_kernel(DlHead *head, const unsigned *__restrict__ bstart, unsigned *__restrict__ sorted, P2 *__restrict__ sorted_xy,
                       const unsigned *__restrict__ heavy_bucket, SubGrid *__restrict__ info, unsigned *__restrict__ sub_start,
                       unsigned long long sub_cap, unsigned *__restrict__ tmp_idx, P2 *__restrict__ tmp_xy, unsigned *__restrict__ tmp_slot)
{
    __shared__ unsigned long long s_k[4];
    __shared__ unsigned s_off, s_carry;
    const int t = threadIdx.x;
    const unsigned n_heavy = head->n_heavy;
    for (unsigned h = blockIdx.x; h < n_heavy; h += gridDim.x) {
        __syncthreads();
        const size_t b = heavy_bucket[h];
        const unsigned lo = bstart[b], hi = bstart[b + 1], m = hi - lo;
        if (t < 4) s_k[t] = (t & 1) ? 0ull : ~0ull;            // ordered keys of xmin, xmax, ymin, ymax
        __syncthreads();
        double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
        for (unsigned j = lo + t; j < hi; j += 256) {
            const P2 q = sorted_xy[j];
            x0 = fmin(x0, q.x); x1 = fmax(x1, q.x); y0 = fmin(y0, q.y); y1 = fmax(y1, q.y);
        }
        if (x0 <= x1) { atomicMin(&s_k[0], okey(x0)); atomicMax(&s_k[1], okey(x1)); atomicMin(&s_k[2], okey(y0)); atomicMax(&s_k[3], okey(y1)); }
        __syncthreads();
        x0 = okey_inv(s_k[0]); x1 = okey_inv(s_k[1]); y0 = okey_inv(s_k[2]); y1 = okey_inv(s_k[3]);
        int K = (int)ceil(sqrt(0.5 * (double)m));
        K = K < 1 ? 1 : (K > 256 ? 256 : K);
        double cs = fmax(x1 - x0, y1 - y0) / (double)K * 1.0000001;
        if (!(cs > 0.0) || !isfinite(cs)) { K = 1; cs = 1.0; }   // (every entry on one spot: duplicates that were blanked, one live site)
        Grid g;
        g.ox = x0; g.oy = y0; g.s = cs; g.inv_s = 1.0 / cs; g.gx = K; g.gy = K;
        const unsigned cells = (unsigned)(K * K);
        if (t == 0) s_off = atomicAdd(&head->sub_used, cells + 1u);
        __syncthreads();
        const unsigned off = s_off;
        if ((unsigned long long)off + cells + 1u > sub_cap) { if (t == 0) atomicOr(&head->err, kErrDegenerate); continue; }   // (cannot happen: DlWs::sub_cap)
        unsigned *st = sub_start + off;
        for (unsigned c = t; c <= cells; c += 256) st[c] = 0u;
        __syncthreads();
        for (unsigned j = lo + t; j < hi; j += 256) {
            const P2 q = sorted_xy[j];
            tmp_slot[j] = atomicAdd(&st[(unsigned)g.by(q.y) * (unsigned)K + (unsigned)g.bx(q.x)], 1u);
        }
        __syncthreads();
        if (t == 0) s_carry = lo;
        __syncthreads();
        for (unsigned base = 0; base <= cells; base += 1024u) {       // counts -> absolute starts, in place
            unsigned e[4], v = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const unsigned i = base + 4u * t + k; e[k] = i <= cells ? st[i] : 0u; v += e[k]; }
            unsigned total;
            unsigned run = block_exscan(v, total) + s_carry;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const unsigned i = base + 4u * t + k; if (i <= cells) st[i] = run; run += e[k]; }
            __syncthreads();
            if (t == 0) s_carry += total;
            __syncthreads();
        }
        for (unsigned j = lo + t; j < hi; j += 256) {
            const P2 q = sorted_xy[j];
            const unsigned dst = st[(unsigned)g.by(q.y) * (unsigned)K + (unsigned)g.bx(q.x)] + tmp_slot[j];
            tmp_idx[dst] = sorted[j]; tmp_xy[dst] = q;
        }
        __syncthreads();
        for (unsigned j = lo + t; j < hi; j += 256) { sorted[j] = tmp_idx[j]; sorted_xy[j] = tmp_xy[j]; }
        __syncthreads();
        for (unsigned c = t; c < cells; c += 256) {                   // every cell ascending by point index (blanked entries last)
            const unsigned a = st[c], e = st[c + 1];
            for (unsigned i = a + 1; i < e; ++i) {
                const unsigned v = sorted[i];
                const P2 q = sorted_xy[i];
                unsigned j = i;
                while (j > a && sorted[j - 1] > v) { sorted[j] = sorted[j - 1]; sorted_xy[j] = sorted_xy[j - 1]; --j; }
                sorted[j] = v; sorted_xy[j] = q;
            }
        }
        if (t == 0) { SubGrid sg; sg.g = g; sg.off = off; sg.pad = 0u; info[h] = sg; }
    }
}

// every how-manieth grid cell dl_cell_sample_kernel looks at (about 4 096 of them)
__host__ __device__ inline size_t fan_sample_stride(size_t n) { return n / 4096 + 1; }

// Is the mesh worth proposing stars from?  On a strongly sheared field (BASELINE config 5: every source row slides 1 .. 19 px
// against the next, 42 % of the sites are duplicates) the Delaunay neighbours of a site have nothing to do with its grid
// neighbours: no cell verifies, and the cell, cell-flag and fan passes (0.6 ms at 8K) verify nothing.  dl_cell_sample_kernel
// runs the cell test on every fan_sample_stride-th grid cell BEFORE those passes (a few thousand cells, whatever the slab: the
// ranks of a slab-wise call and a whole-field call must decide alike -- which pass settles a site shows in the order of its
// neighbours, hence in triangle ids and ties); when at least 64 samples have four kept corners and fewer than 1 in 128 of those
// verify, all three passes step aside and every site goes to the clip pass.  A speed decision only: the clip pass computes the
// same stars the cells and fans would have verified.
__device__ __forceinline__ bool dl_skip_mesh(const DlHead *head) { return head->sample_all >= 64u && head->sample_ok * 128u < head->sample_all; }

// ------------------------------------------------------------------------------------------------ stars, mesh-cell pass
// One thread per grid cell: both triangles of an intact, convex, positively oriented cell are verified ONCE against the
// sites under their circumcircles (ofl_dl::cell_verify) instead of three times from their three vertices; then one thread
// per site reads the flags of its four cells -- all verified: the star is written without touching a candidate.
__global__ __launch_bounds__(256)
void dl_cell_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                    const DlHead *__restrict__ head, const unsigned *__restrict__ bstart,
                    const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy, const unsigned char *__restrict__ dup,
                    unsigned char *__restrict__ cellflag, unsigned char *__restrict__ tileflag, int tiles_x, int ntiles)
{
    const unsigned per = gridDim.x >> 3;                        // (grid padded to a multiple of 8) one contiguous eighth per XCD
    const int tile = (int)((blockIdx.x & 7u) * per + (blockIdx.x >> 3));
    if (tile >= ntiles) return;
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 32 + (threadIdx.x & 31), y = ty * 8 + (threadIdx.x >> 5);
    const bool in_grid = x < W && y < H;
    const size_t ia = in_grid ? (size_t)y * W + x : 0;
    int flag = 0;
    if (in_grid && x < W - 1 && y < H - 1 && !(head->err & kErrDegenerate) && !dl_skip_mesh(head) &&
        kept_pt(pmask, ia) && kept_pt(pmask, ia + 1) && kept_pt(pmask, ia + W) && kept_pt(pmask, ia + W + 1) &&
        !(dup[ia] | dup[ia + 1] | dup[ia + W] | dup[ia + W + 1])) {
        const D2 a = point_of(flow, sign, W, x, y), b = point_of(flow, sign, W, x + 1, y);
        const D2 c = point_of(flow, sign, W, x + 1, y + 1), d = point_of(flow, sign, W, x, y + 1);
        // (slab mode: a cell none of whose corners gets a star here is not looked at -- every site that does get one still
        // finds all four of its cells verified, so its star comes out of the same pass as in a whole-field run)
        const double lo = head->need_lo, hi = head->need_hi;
        if (fmax(fmax(a.y, b.y), fmax(c.y, d.y)) >= lo && fmin(fmin(a.y, b.y), fmin(c.y, d.y)) <= hi)
        flag = cell_verify((int)ia, W, P2{ a.x, a.y }, P2{ b.x, b.y }, P2{ c.x, c.y }, P2{ d.x, d.y }, head->grid, bstart, sorted, sorted_xy,
                           PosFn(flow, sign, W), kFanSpan);
    }
    if (in_grid) cellflag[ia] = (unsigned char)flag;
    // a CLEAN tile: every one of its 256 cells verified (tiles that reach the last row or column of sites never are)
    const int clean = __syncthreads_and(flag != 0);
    if (threadIdx.x == 0) tileflag[tile] = clean ? 1 : 0;
}

// the sample of grid cells that decides whether the mesh is a guide to this triangulation at all (dl_skip_mesh)
__global__ __launch_bounds__(256)
void dl_cell_sample_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                           DlHead *__restrict__ head, const unsigned *__restrict__ bstart,
                           const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy, const unsigned char *__restrict__ dup)
{
    const size_t n = (size_t)H * W, stride = fan_sample_stride(n);
    const size_t ia = ((size_t)blockIdx.x * 256 + threadIdx.x) * stride;
    unsigned all = 0, ok = 0;
    if (ia < n && !(head->err & kErrDegenerate)) {
        const int y = (int)(ia / (unsigned)W), x = (int)(ia - (size_t)y * W);
        if (x < W - 1 && y < H - 1 &&
            kept_pt(pmask, ia) && kept_pt(pmask, ia + 1) && kept_pt(pmask, ia + W) && kept_pt(pmask, ia + W + 1) &&
            !(dup[ia] | dup[ia + 1] | dup[ia + W] | dup[ia + W + 1])) {
            const D2 a = point_of(flow, sign, W, x, y), b = point_of(flow, sign, W, x + 1, y);
            const D2 c = point_of(flow, sign, W, x + 1, y + 1), d = point_of(flow, sign, W, x, y + 1);
            all = 1;
            ok = cell_verify((int)ia, W, P2{ a.x, a.y }, P2{ b.x, b.y }, P2{ c.x, c.y }, P2{ d.x, d.y }, head->grid, bstart, sorted, sorted_xy,
                             PosFn(flow, sign, W), kFanSpan) != 0 ? 1u : 0u;
        }
    }
    const unsigned long long ball = __ballot(all != 0), bok = __ballot(ok != 0);
    if ((threadIdx.x & 63) == 0) {
        if (ball) atomicAdd(&head->sample_all, (unsigned)__popcll(ball));
        if (bok) atomicAdd(&head->sample_ok, (unsigned)__popcll(bok));
    }
}

__global__ __launch_bounds__(256)
void dl_site_cells_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W, const DlHead *__restrict__ head,
                          const unsigned char *__restrict__ dup, const unsigned char *__restrict__ cellflag,
                          unsigned char *__restrict__ deg, unsigned *__restrict__ nbr)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= (size_t)H * W) return;
    if (!kept_pt(pmask, p) || dup[p]) { deg[p] = 0; return; }
    if (head->err & kErrDegenerate) { deg[p] = kDegTodo; return; }           // (no star pass runs on a refused point set)
    const int y = (int)(p / (unsigned)W), x = (int)(p - (size_t)y * W);
    if (head->need_lo > -1e300 || head->need_hi < 1e300) {   // slab mode: no star for a site outside the slab (degree 0: nothing is drawn from it, nothing looks it up)
        const D2 q = point_of(flow, sign, W, x, y);
        if (!(q.y >= head->need_lo && q.y <= head->need_hi)) { deg[p] = 0; return; }
    }
    int n = 0;
    if (x > 0 && y > 0 && x < W - 1 && y < H - 1) {
        unsigned out[8];
        n = star_from_cells((int)p, W, cellflag[p], cellflag[p - 1], cellflag[p - W - 1], cellflag[p - W], out);
        if (n > 0) {
            uint4 *row = reinterpret_cast<uint4 *>(nbr + p * kSlots);
            row[0] = make_uint4(out[0], out[1], out[2], out[3]);
            row[1] = make_uint4(out[4], n > 5 ? out[5] : 0u, n > 6 ? out[6] : 0u, n > 7 ? out[7] : 0u);
        }
    }
    deg[p] = n > 0 ? (unsigned char)n : (dl_skip_mesh(head) ? kDegTodo : kDegFan);      // (no fans either on a mesh that is no guide)
}

// ------------------------------------------------------------------------------------------------ stars, mesh-fan pass
// One thread per point: a point whose eight grid neighbours are kept proposes the star of the cell-wise mesh and verifies
// it against the sites under its circumcircles (ofl_dl::star_fan).  Verified stars are final; everything else is marked
// for the clip pass.
__global__ __launch_bounds__(kFanBlock, 5)                    // (five waves per SIMD: the pass waits on its candidate loads)
void dl_star_fan_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                        const DlHead *__restrict__ head, const unsigned *__restrict__ todo, const unsigned *__restrict__ bstart,
                        const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy, const unsigned char *__restrict__ dup,
                        unsigned char *__restrict__ deg, unsigned *__restrict__ nbr)
{
    __shared__ P2 s_rel[8][kFanBlock];
    const unsigned n_fan = head->n_fan;                         // the sites the cell pass did not settle, in index order (none when dl_skip_mesh)
    for (unsigned i = blockIdx.x * kFanBlock + threadIdx.x; i < n_fan; i += gridDim.x * kFanBlock) {
        const size_t p = todo[i];
        const int y = (int)(p / (unsigned)W), x = (int)(p - (size_t)y * W);
        // which of the eight grid neighbours exist: inside the grid, kept by the point mask, not a dropped duplicate
        unsigned kept8 = 0;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) {
            const int dx = (int)((0x901Au >> (2 * sl)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * sl)) & 3u) - 1;
            const bool in = (unsigned)(x + dx) < (unsigned)W && (unsigned)(y + dy) < (unsigned)H;
            const size_t q = in ? (size_t)((long long)p + dx + (long long)dy * W) : p;
            if (in && kept_pt(pmask, q) && !dup[q]) kept8 |= 1u << sl;
        }
        int n = 0;
        {
            const Grid g = head->grid;
            const PosFn pos(flow, sign, W);
            auto npos = [&](int sl) {                               // slot -> grid offset at compile time (the loops over slots are unrolled)
                const int dx = (int)((0x901Au >> (2 * sl)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * sl)) & 3u) - 1;
                const D2 q = point_of(flow, sign, W, x + dx, y + dy);
                return P2{ q.x, q.y };
            };
            const D2 c = point_of(flow, sign, W, x, y);
            n = star_fan((int)p, W, P2{ c.x, c.y }, kept8, pos, npos, g, bstart, sorted, sorted_xy, kFanSpan, &s_rel[0][threadIdx.x], kFanBlock,
                         nbr + p * kSlots);
        }
        deg[p] = n > 0 ? (unsigned char)n : kDegTodo;
    }
}

// ------------------------------------------------------------------------------------------------ stars, clip pass (near)
// One thread per point the cells and fans did not settle, in index order.  (Round 3 measured two spatially coherent
// schedules for BASELINE config 5, whose source rows scatter over hundreds of buckets: one wave per 8 x 8 bucket tile -- half
// empty waves, 24 -> 46 ms -- and a list compacted in bucket order -- no change: the pass is bound by the divergent clip
// code of its 64 lanes, not by where the candidates come from.)
#ifndef OFL_NEAR_WAVES
#define OFL_NEAR_WAVES 5      // measured (product flags, same box): 4 waves per SIMD (116 VGPRs, cells of 12) config 5 22.88 ms, 5 waves (96 VGPRs + 9 spilled,
#endif                        // cells of 10: 7 680 B of LDS per wave) 21.86 ms, 6 waves (80 + 18 spilled, cells of 8) 22.69 ms; cells of 10 at 4 waves: 22.90 ms
template <bool HEAVY>       // false: every site of the list; true (a second launch, at once over when no bucket is heavy): the sites the first one marked kDegHeavy
__global__ __launch_bounds__(64, HEAVY ? 1 : OFL_NEAR_WAVES)
void dl_star_near_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, const unsigned char *__restrict__ dup, int H, int W,
                         const DlHead *__restrict__ head, const unsigned *__restrict__ todo,
                         const unsigned *__restrict__ bstart,
                         const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy,
                         unsigned char *__restrict__ deg, unsigned *__restrict__ nbr,
                         const unsigned *__restrict__ heavy_bucket, const SubGrid *__restrict__ heavy_info, const unsigned *__restrict__ sub_start)
{
    __shared__ float s_vx[kNearCap][64], s_vy[kNearCap][64];
    __shared__ int   s_tag[kNearCap][64];
    const unsigned n_todo = head->n_todo;
    const Grid g = head->grid;
    const PosFn pos(flow, sign, W);
    const SubGrids subs{ heavy_bucket, heavy_info, sub_start, head->n_heavy };       // (dense clusters: ofl_dl::apply_heavy_run)
    const SubGrids *sub = subs.n ? &subs : nullptr;
    if (HEAVY && !sub) return;
    for (unsigned base = blockIdx.x * 64; base < n_todo; base += gridDim.x * 64) {
        if (base + threadIdx.x >= n_todo) return;
        const size_t p = todo[base + threadIdx.x];
        if (HEAVY && deg[p] != kDegHeavy) continue;
        PolyT<float> P{ &s_vx[0][threadIdx.x], &s_vy[0][threadIdx.x], &s_tag[0][threadIdx.x], 64, kNearCap, 0 };
        int rings_done;
        const P2 pp = pos((int)p);
        // A cell that is still unbounded after kOpenRings rings is clipped with the site's GRID neighbours before it is given
        // up: across a tear of the mesh (a motion boundary) the neighbour on the other side closes the cell, and a closed cell
        // finishes in the second per-thread pass within a few coarse rings instead of waiting for the far side to turn up
        // (64-px stripes at 4K: that pass 1.09 -> 0.35 ms; hull points stay open, as they must).
        auto rescue = [&](PolyT<float> &Q) -> int {
            const int y = (int)(p / (unsigned)W), x = (int)(p - (size_t)y * W);
            auto rel = [&](int t) { const P2 v = pos(t); return P2{ v.x - pp.x, v.y - pp.y }; };
#pragma unroll 1
            for (int sl = 0; sl < 8; ++sl) {
                const int dx = (int)((0x901Au >> (2 * sl)) & 3u) - 1, dy = (int)((0x01A9u >> (2 * sl)) & 3u) - 1;
                if ((unsigned)(x + dx) >= (unsigned)W || (unsigned)(y + dy) >= (unsigned)H) continue;
                const size_t q = (size_t)((long long)p + dx + (long long)dy * W);
                if (!kept_pt(pmask, q) || dup[q]) continue;
                const P2 C = rel((int)q);
                if (C.x == 0.0 && C.y == 0.0) continue;
                if (poly_clip(Q, C, (int)q, (int)p, rel) < 0) return -1;
            }
            return 0;
        };
        // (a tighter bound -- reach beyond 1 .. 16 times the ring search's own radius -- was measured: no further gain)
        const int rc = star_near<HEAVY>(P, (int)p, pp, g, bstart, sorted, pos, kRings, sorted_xy, kOpenRings, &rings_done, rescue, 4.0 * head->far_t2, sub);
        if (!HEAVY && rc == 2) { deg[p] = kDegHeavy; continue; }
        bool ok = rc == 1;
        for (int k = 0; ok && k < P.n; ++k) ok = P.T(k) >= 0;
        if (!ok) {
            // unfinished: the edges of the cell so far -- their sites in cyclic order, box sides (negative) included -- seed
            // the later passes, which rebuild the cell from them in one step (slot 0 will hold the point's rank, slot 1 the
            // number of seeds, slot 14 the last fine ring that was applied completely)
            deg[p] = kDegFar;
            for (int k = 0; k < P.n; ++k) nbr[p * kSlots + 2 + k] = (unsigned)P.T(k);
            nbr[p * kSlots + 1] = (unsigned)P.n;
            nbr[p * kSlots + 14] = (unsigned)rings_done;
            continue;
        }
        deg[p] = (unsigned char)P.n;
        for (int k = 0; k < P.n; ++k) nbr[p * kSlots + k] = (unsigned)P.T(k);
    }
}

// ------------------------------------------------------------------------------------------------ stars, second per-thread pass
// One thread per unfinished point that is not on the image border (those are hull points more often than not): the cell
// is rebuilt from its seeds and finished against the coarse grid of unfinished points (ofl_dl::star_near2).  What closes
// here -- the rims of tears at motion boundaries, of small holes -- becomes an ordinary small star; the rest stays
// marked for the cooperative passes.
__global__ __launch_bounds__(64)
void dl_star_near2_kernel(const float *__restrict__ flow, int sign, int H, int W, const DlHead *__restrict__ head,
                          const unsigned *__restrict__ far_idx, const unsigned *__restrict__ bstart,
                          const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy,
                          const unsigned *__restrict__ b1start, const unsigned *__restrict__ sorted1_pt,
                          const P2 *__restrict__ sorted1_xy, unsigned char *__restrict__ deg, unsigned *__restrict__ nbr,
                          unsigned *__restrict__ far_deg, unsigned char *__restrict__ far_wide, unsigned min_points,
                          const unsigned *__restrict__ heavy_bucket, const SubGrid *__restrict__ heavy_info, const unsigned *__restrict__ sub_start)
{
    __shared__ float s_vx[kSlots][64], s_vy[kSlots][64];
    __shared__ int   s_tag[kSlots][64];
    const unsigned n_far = head->n_far;
    if (n_far < min_points) return;                        // (a per-thread pass needs tens of thousands of points to fill the chip)
    for (unsigned rank = blockIdx.x * 64 + threadIdx.x; rank < n_far; rank += gridDim.x * 64) {
    const int p = (int)far_idx[rank];
    const int py = (int)((unsigned)p / (unsigned)W), px = p - py * W;
    if (px == 0 || py == 0 || px == W - 1 || py == H - 1) continue;
    const Grid g = head->grid, g1 = head->grid1;
    const PosFn pos(flow, sign, W);
    unsigned *row = nbr + (size_t)p * kSlots;
    const int ns = min((int)row[1], kSlots - 4);
    PolyT<float> P{ &s_vx[0][threadIdx.x], &s_vy[0][threadIdx.x], &s_tag[0][threadIdx.x], 64, kSlots, 0 };
    const SubGrids subs{ heavy_bucket, heavy_info, sub_start, head->n_heavy };
    const int rc = star_near2(P, p, pos(p), row + 2, ns, (int)row[14], kRings, g, bstart, sorted, sorted_xy,
                              kNear2Rings, g1, b1start, sorted1_pt, sorted1_xy, pos, kNear2Open, 4.0 * head->far_t2, subs.n ? &subs : nullptr);
    if (rc != 1) continue;
    for (int k = 0; k < P.n; ++k) row[k] = (unsigned)P.T(k);
    deg[p] = (unsigned char)P.n;
    far_deg[rank] = 0;                                     // not a cooperative-pass star: nothing in the pool
    far_wide[rank] = 0;                                    // (its reach is within the coarse rings every later pass searches)
    }
}

// Is grid site (x, y) CLEAN: do the four cells around it, and the four cells around each of its grid neighbours NW, N, NE and W
// (the only ones with a smaller index), all lie in clean tiles (dl_cell_kernel: every cell of the 32 x 8 tile verified)?  Then
// its star and theirs were written from cell flags (dl_site_cells_kernel; ofl_dl::star_from_cells: E, SE?, S, SW?, W, NW?, N,
// NE?), every triangle it lists is a triangle of a verified cell and is owned by a site whose star lists it: the triangles a
// clean site OWNS are drawn cell by cell (dl_raster_cells_kernel) with the ids the site-wise raster would give them, and that
// raster has nothing to do for a clean site.  The cells in question are [x - 2, x + 1] x [y - 2, y]: at most four tiles.
__device__ __forceinline__ bool site_clean(const unsigned char *__restrict__ tileflag, int x, int y, int W, int H, int tiles_x)
{
    if (x < 2 || y < 2 || x > W - 3 || y > H - 2) return false;
    const int tx0 = (x - 2) >> 5, tx1 = (x + 1) >> 5, ty0 = (y - 2) >> 3, ty1 = y >> 3;
    return tileflag[ty0 * tiles_x + tx0] && tileflag[ty0 * tiles_x + tx1] && tileflag[ty1 * tiles_x + tx0] && tileflag[ty1 * tiles_x + tx1];
}

// compaction in ascending order: MODE 0 = points with deg == kDegFar -> far_idx, MODE 1 = ranks with far_deg == kDegLeft -> left_idx,
// MODE 2 = points with deg == kDegFan -> todo_idx (what the cell pass left to the fan pass),
// MODE 3 = points with deg == kDegTodo -> the clip pass's list (a list in BUCKET order -- entries of `sorted` -- was measured:
// no faster on any field, BASELINE config 5 included, and its compaction costs 0.2 ms more at 4K: the gather of deg[sorted[j]])
// MODE 1 also lists the ranks the wave pass finished beyond kMidRings coarse rings (aux = far_wide): they are not computed
// again, but the workgroup pass searches the coarse grid only that far around a point, and a neighbour whose own cell
// reaches farther would otherwise be missing from its candidates.
// the flags of the eight elements [base, base + 8) as bits (base a multiple of 8; elements at or beyond n are not flagged):
// one 8-byte load of the degree bytes -- or two 16-byte loads of the ranks' words and one 8-byte load of `aux`
template <int MODE>
__device__ __forceinline__ unsigned flagged8(const void *src, const unsigned char *aux, size_t base, size_t n, int W, int H);

template <int MODE>
__device__ __forceinline__ bool flagged(const void *src, const unsigned char *aux, size_t i)
{
    if (MODE == 5) return ((const unsigned *)src)[i + 1] - ((const unsigned *)src)[i] > kHeavy;      // (src = bstart: a heavy bucket)
    if (MODE == 3) return ((const unsigned char *)src)[i] == kDegTodo;
    return MODE == 0 ? ((const unsigned char *)src)[i] == kDegFar
         : MODE == 2 ? ((const unsigned char *)src)[i] == kDegFan : (((const unsigned *)src)[i] == kDegLeft || aux[i] != 0);
}

// Ordered compaction in ONE launch (decoupled look-back).  A workgroup draws a SUPER TILE of kCompactTiles x kScanChunk elements by
// ticket -- so a tile's predecessors are resident or done when it starts waiting for them -- counts it, publishes the count
// as (flag | value) in one word (1 << 30: the tile's own aggregate, 1 << 31: the inclusive prefix) with agent-scope (sc1)
// stores, and its first wave looks back 64 tiles at a time (sc1 loads: the per-XCD L2s are not coherent) until it meets an
// inclusive word; then the super tile is walked a second time (its flags come from L2) and the list is written.  A 4K field is
// 254 super tiles -- one per CU, one look-back window deep: with one tile per 2048 elements the thousands of tiles that start
// together each walked back over all the others (65 us per list; the three-kernel count / scan / write it replaced: 27 us).
// state[0] = the ticket counter, state[4 + t] = tile t's word; zeroed by the caller.
template <int MODE>
__device__ __forceinline__ unsigned flagged8(const void *src, const unsigned char *aux, size_t base, size_t n, int W, int H)
{
    unsigned bits = 0;
    if (MODE == 4) {
        // MODE 4: the sites the site-wise raster has to look at -- a star of 1 .. kSlots neighbours, and not clean (aux = the
        // tile flags): the clean ones are drawn cell by cell
        if (base >= n) return 0u;
        const uint2 w = base + 8 <= n ? *reinterpret_cast<const uint2 *>((const unsigned char *)src + base) : make_uint2(0u, 0u);
        int y = (int)((float)base / (float)W);
        int x = (int)((long long)base - (long long)y * W);
        while (x < 0) { --y; x += W; }
        while (x >= W) { ++y; x -= W; }
        const int tiles_x = (W + 31) / 32;
        // (eight sites of one row whose cells all lie in clean tiles: nothing to list -- four bytes instead of thirty-two)
        if (x + 7 < W && x >= 2 && y >= 2 && x + 7 <= W - 3 && y <= H - 2) {
            const int tx0 = (x - 2) >> 5, tx1 = (x + 8) >> 5, ty0 = (y - 2) >> 3, ty1 = y >> 3;
            if (aux[ty0 * tiles_x + tx0] && aux[ty0 * tiles_x + tx1] && aux[ty1 * tiles_x + tx0] && aux[ty1 * tiles_x + tx1]) return 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned d = base + 8 <= n ? (((k < 4 ? w.x : w.y) >> (8 * (k & 3))) & 0xFFu) : (base + k < n ? ((const unsigned char *)src)[base + k] : 0u);
            if (d >= 1u && d <= (unsigned)kSlots && !site_clean(aux, x, y, W, H, tiles_x)) bits |= 1u << k;
            if (++x == W) { x = 0; ++y; }
        }
        return bits;
    }
    if (MODE == 5) {
        if (base + 8 <= n) {
            const uint4 a = *reinterpret_cast<const uint4 *>((const unsigned *)src + base), b = *reinterpret_cast<const uint4 *>((const unsigned *)src + base + 4);
            const unsigned e[9] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, ((const unsigned *)src)[base + 8] };
#pragma unroll
            for (int k = 0; k < 8; ++k) if (e[k + 1] - e[k] > kHeavy) bits |= 1u << k;
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (base + k < n && flagged<5>(src, aux, base + k)) bits |= 1u << k;
        }
        return bits;
    }
    if (base + 8 <= n) {
        if (MODE != 1) {
            const unsigned char want = MODE == 3 ? kDegTodo : (MODE == 0 ? kDegFar : kDegFan);
            const uint2 w = *reinterpret_cast<const uint2 *>((const unsigned char *)src + base);
#pragma unroll
            for (int k = 0; k < 8; ++k) if ((((k < 4 ? w.x : w.y) >> (8 * (k & 3))) & 0xFFu) == want) bits |= 1u << k;
        } else {
            const uint4 a = *reinterpret_cast<const uint4 *>((const unsigned *)src + base), b = *reinterpret_cast<const uint4 *>((const unsigned *)src + base + 4);
            const uint2 x = *reinterpret_cast<const uint2 *>(aux + base);
            const unsigned d[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
            for (int k = 0; k < 8; ++k) if (d[k] == kDegLeft || (((k < 4 ? x.x : x.y) >> (8 * (k & 3))) & 0xFFu) != 0u) bits |= 1u << k;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) if (base + k < n && flagged<MODE>(src, aux, base + k)) bits |= 1u << k;
    }
    return bits;
}

constexpr int kCompactTiles = 16;
template <int MODE>
__global__ __launch_bounds__(256)
void dl_compact_kernel(const void *__restrict__ src, const unsigned char *__restrict__ aux, DlHead *head, size_t n_fixed,
                       unsigned *__restrict__ state, unsigned *__restrict__ list, unsigned *__restrict__ rank_of, int W = 0, int H = 0)
{
    __shared__ unsigned s_tile, s_prefix;
    if (threadIdx.x == 0) s_tile = atomicAdd(&state[0], 1u);
    __syncthreads();
    const unsigned tile = s_tile;
    // (MODE 5, the heavy buckets: nothing to look at unless the dedupe pass met one -- every tile then publishes an empty prefix)
    const size_t n = MODE == 5 ? (head->any_heavy ? (size_t)head->grid.gx * head->grid.gy : 0) : (MODE != 1 ? n_fixed : head->n_far);
    const size_t base0 = (size_t)tile * kCompactTiles * kScanChunk + (size_t)threadIdx.x * 8;
    unsigned v = 0;
    unsigned keep[kCompactTiles / 4];                     // the flag bits of the thread's 16 x 8 elements: the second walk does not evaluate them again
#pragma unroll
    for (int c = 0; c < kCompactTiles; ++c) {
        const unsigned b8 = flagged8<MODE>(src, aux, base0 + (size_t)c * kScanChunk, n, W, H);
        v += (unsigned)__popc(b8);
        if ((c & 3) == 0) keep[c >> 2] = b8; else keep[c >> 2] |= b8 << (8 * (c & 3));
    }
    unsigned total;
    (void)block_exscan(v, total);
    if (threadIdx.x < 64) {
        unsigned *st = state + 4;
        const int lane = threadIdx.x;
        if (tile > 0 && lane == 0) __hip_atomic_store(&st[tile], 0x40000000u | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned prefix = 0;
        for (long long hi = (long long)tile - 1; hi >= 0;) {
            const long long j = hi - lane;
            unsigned w;
            for (;;) {
                w = j >= 0 ? __hip_atomic_load(&st[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0x80000000u;     // (before tile 0: an inclusive 0)
                if (!__ballot(w == 0u)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            const unsigned long long inc = __ballot((w & 0x80000000u) != 0u);
            const int stop = inc ? __ffsll((long long)inc) - 1 : 63;          // the nearest inclusive word of the window (lane 0 = nearest tile)
            unsigned v2 = lane <= stop ? (w & 0x3FFFFFFFu) : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v2 += (unsigned)__shfl_xor((int)v2, off);
            prefix += v2;
            if (inc) break;
            hi -= 64;
        }
        if (lane == 0) {
            __hip_atomic_store(&st[tile], 0x80000000u | (prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = prefix;
        }
    }
    __syncthreads();
    unsigned run = s_prefix;
#pragma unroll
    for (int c = 0; c < kCompactTiles; ++c) {
        const size_t base = base0 + (size_t)c * kScanChunk;
        if (base - (size_t)threadIdx.x * 8 >= n) break;        // (uniform: the whole chunk lies beyond the end)
        const unsigned bits = (keep[c >> 2] >> (8 * (c & 3))) & 0xFFu, cv = (unsigned)__popc(bits);
        unsigned ctotal;
        unsigned at = run + block_exscan(cv, ctotal);
        run += ctotal;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if ((bits >> k) & 1u) {
                if (MODE == 0 && rank_of) rank_of[(base + k) * kSlots] = at;     // an unfinished point's first neighbour slot holds its rank
                if (MODE != 5 || at < n_fixed) list[at] = (unsigned)(base + k);
                ++at;
            }
    }
    if (tile == gridDim.x - 1 && threadIdx.x == 0) {
        unsigned cnt = s_prefix + total;
        if (MODE == 0 && (unsigned long long)n_fixed * kSlots + (unsigned long long)cnt * kFarK >= 0xFFFFFFF0ull) {
            atomicOr(&head->err, 8u);                      // the unfinished stars exceed the triangle-id space: none is built, the caller is told
            cnt = 0;
        }
        if ((MODE == 0 && cnt > kMaxFar) || (MODE == 1 && cnt > kMaxLeft)) { atomicOr(&head->err, kErrDegenerate); cnt = 0; }
        if (head->err & kErrDegenerate) cnt = 0;           // a degenerate point set: no star pass runs (their loops are sized for ordinary buckets)
        if (MODE == 5 && cnt > n_fixed) { atomicOr(&head->err, kErrDegenerate); cnt = 0; }       // (n_fixed: the capacity of the list; cannot happen by its bound)
        if (MODE == 0) head->n_far = cnt; else if (MODE == 1) head->n_left = cnt; else if (MODE == 2) head->n_fan = cnt;
        else if (MODE == 3) head->n_todo = cnt; else if (MODE == 4) head->n_raster = cnt; else head->n_heavy = cnt;
    }
}

// ------------------------------------------------------------------------------------------------ stars, cooperative passes
// One workgroup of NT threads per unfinished point.  The cell lives in LDS; candidates are tested NT at a time against
// the current cell (one candidate per thread), and the few that cut it are applied one after the other, nearest first,
// by the whole workgroup (vertex flags and the shift of the surviving vertices in parallel).
//   wave pass (CAP 256, NT 64): fine buckets around the point, then rings of the COARSE grid of unfinished points with
//     the security-radius test -- rims of holes, motion boundaries, anything whose cell spans tens of pixels;
//   left-over pass (CAP 384, NT 64; then CAP 2560, NT 256 for the few cells that overflow): what is left -- hull points
//     (unbounded cells: the points of the image border come here directly), rims of very large holes, fan apexes of
//     border pockets -- against the same candidates plus ALL other points left (a Delaunay neighbour of a left-over point
//     beyond the coarse rings is itself left over).  A hull cell has half a dozen edges and most chunks of the sweep cut
//     nothing: the pass waits -- on the chain of loads that finds the point and its seeds, on votes -- far more than it
//     computes, so it runs as single waves, which puts four times as many points in flight per CU as workgroups of 256
//     did (121 VGPRs, 9.4 KB of LDS: 16 per CU), and since a vote within one wave costs next to nothing, every step of 64
//     candidates is tested against the cell as it stands (4K, axis-parallel border: 0.77 -> 0.38 ms; a rotated border:
//     1.25 -> 1.17 ms).
template <int CAP, int NT>
struct FarLds {
    double vx[CAP], vy[CAP];
    int    tag[CAP];
    unsigned char cut[CAP];
    int    cidx[1];            // the candidate being applied (far_chunks)
    unsigned clist[NT];        // chunk numbers the sweep of the workgroup pass keeps (of a group of NT chunks)
    double ccx[1], ccy[1];
    unsigned long long hit[NT / 64];
    double wmax[NT / 64], wfar[NT / 64];
    unsigned run_lo[64];       // candidate runs of the current step (ranges of a sorted list) ...
    int    run_pre[65];        // ... and the exclusive prefix of their lengths
    int    n, a, ncut, nstart, status;
    int    napply;             // cooperative clips applied to this cell so far (counted for the experiments build's debug line)
    // Candidate rejection: a site can only cut a vertex v if it is closer than 2 |v| to the cell's site.  Vertices well
    // outside the data (|v|^2 > t2: the box vertices of an unbounded cell, the circumcentres of sliver triangles along a
    // straight border) would make that radius useless, so they are listed and tested one by one; reach2 covers the rest.
    double reach2;             // (2 * farthest NEAR vertex)^2; 1e300 when the far list overflowed (every vertex is tested)
    double t2;
    int    nfar;
    int    farlist[kFarList];
    // The far vertices of a convex cell are ONE run of the polygon, seen from the site inside a cone of directions (the
    // outward normal cone of a hull point).  When the run [f0, f1] spans less than a half turn, v . c over the run is
    // bounded by its two ends, and a site whose direction lies outside the cone with
    // max(v0 . c / |v0|, v1 . c / |v1|) * vmax < |c|^2 / 2 (minus the margin of vertex_cut) cuts none of them: two dot
    // products instead of the list.  cone = 0 when the run is not unique or too wide.
    int    cone, nrun, f0, f1;
    double c0x, c0y, c1x, c1y, kcone;    // unit directions of the run's ends, 0.999 / (2 * largest |v| in it)
};

template <int CAP, int NT>
__device__ void far_shift(FarLds<CAP, NT> &L, int s0, int s1, int d0)
{
    // moves vertices [s0, s1) to [d0, d0 + s1 - s0); ranges may overlap; all threads of the workgroup take part
    if (d0 == s0 || s1 <= s0) return;
    const int t = threadIdx.x;
    if (d0 < s0) {
        for (int base = s0; base < s1; base += NT) {
            const int k = base + t;
            double x = 0, y = 0; int g = 0;
            if (k < s1) { x = L.vx[k]; y = L.vy[k]; g = L.tag[k]; }
            __syncthreads();
            if (k < s1) { L.vx[k - s0 + d0] = x; L.vy[k - s0 + d0] = y; L.tag[k - s0 + d0] = g; }
            __syncthreads();
        }
    } else {
        for (int end = s1; end > s0; end -= NT) {
            const int k = end - 1 - t;
            double x = 0, y = 0; int g = 0;
            if (k >= s0) { x = L.vx[k]; y = L.vy[k]; g = L.tag[k]; }
            __syncthreads();
            if (k >= s0) { L.vx[k - s0 + d0] = x; L.vy[k - s0 + d0] = y; L.tag[k - s0 + d0] = g; }
            __syncthreads();
        }
    }
}

// reach, far list and far cone of the current cell; all threads, ends with a barrier
template <int CAP, int NT>
__device__ __noinline__ void far_refresh(FarLds<CAP, NT> &L)
{
    const int t = threadIdx.x, n = L.n;
    const double t2 = L.t2;
    if (t == 0) { L.nfar = 0; L.nrun = 0; L.f0 = 0; L.f1 = 0; }
    __syncthreads();
    double r2 = 0.0, f2 = 0.0;
    for (int k = t; k < n; k += NT) {
        const double v2 = L.vx[k] * L.vx[k] + L.vy[k] * L.vy[k];
        const bool far = v2 > t2;
        L.cut[k] = far ? 1 : 0;                           // (far_apply rewrites the flags before it reads them)
        if (far) { const int slot = atomicAdd(&L.nfar, 1); if (slot < kFarList) L.farlist[slot] = k; f2 = fmax(f2, v2); }
        else r2 = fmax(r2, v2);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r2 = fmax(r2, __hiloint2double(__shfl_xor(__double2hiint(r2), off), __shfl_xor(__double2loint(r2), off)));
        f2 = fmax(f2, __hiloint2double(__shfl_xor(__double2hiint(f2), off), __shfl_xor(__double2loint(f2), off)));
    }
    if (NT > 64) { if ((t & 63) == 0) { L.wmax[t >> 6] = r2; L.wfar[t >> 6] = f2; } }
    __syncthreads();
    for (int k = t; k < n; k += NT)
        if (L.cut[k]) {
            if (!L.cut[k == 0 ? n - 1 : k - 1]) { atomicAdd(&L.nrun, 1); L.f0 = k; }
            if (!L.cut[k + 1 == n ? 0 : k + 1]) L.f1 = k;
        }
    __syncthreads();
    if (t == 0) {
        double m = r2, fm = f2;
        if (NT > 64) { m = 0.0; fm = 0.0; for (int w = 0; w < NT / 64; ++w) { m = fmax(m, L.wmax[w]); fm = fmax(fm, L.wfar[w]); } }
        const bool listed = L.nfar <= kFarList;
        L.reach2 = 4.0 * m;
        L.cone = 0;
        if (L.nfar > 0 && L.nrun == 1) {
            const double ax = L.vx[L.f0], ay = L.vy[L.f0], bx = L.vx[L.f1], by = L.vy[L.f1];
            const double na = sqrt(ax * ax + ay * ay), nb = sqrt(bx * bx + by * by);
            const double c0x = ax / na, c0y = ay / na, c1x = bx / nb, c1y = by / nb;
            // counter-clockwise from f0 to f1 by less than a half turn, with room to spare (or a single far vertex)
            if (L.f0 == L.f1 || (c0x * c1y - c0y * c1x > 1e-6 || (c0x * c1x + c0y * c1y > 0.5 && c0x * c1y - c0y * c1x >= 0.0))) {
                const double vmax = sqrt(fm);
                L.c0x = c0x; L.c0y = c0y; L.c1x = c1x; L.c1y = c1y; L.kcone = 0.999 / (2.0 * vmax);
                L.cone = isfinite(vmax) && isfinite(c0x + c0y + c1x + c1y) ? 1 : 0;
            }
        }
        if (!listed && !L.cone) L.reach2 = 1e300;         // neither a list nor a cone: every vertex is tested
        else if (!listed) L.nfar = -L.nfar;              // a cone without a list: sites the cone cannot reject test every vertex
    }
    __syncthreads();
}

// Ties on coincident vertices can leave more than one run of cut vertices (ofl_dl::poly_cutmask): the first run that holds a
// vertex cut beyond the margin is the one that goes -- the first run if none does.  L.cut holds vertex_cut_ex's 0 / 1 / 2,
// L.a receives the start of the run; all threads, ends with a barrier.  (Not inlined: the workgroup pass sits at 121 VGPRs --
// four waves per SIMD -- and this rare step must not take a register from its sweeps.)
template <int CAP, int NT>
__device__ __noinline__ void far_pick_run(FarLds<CAP, NT> &L)
{
    const int t = threadIdx.x, n = L.n;
    __syncthreads();
    if (t == 0) L.a = 0x7fffffff;
    __syncthreads();
    for (int k = t; k < n; k += NT)
        if (L.cut[k] && !L.cut[k == 0 ? n - 1 : k - 1]) {
            bool sure = false;
            for (int m = k, c = 0; c < n && L.cut[m]; m = m + 1 == n ? 0 : m + 1, ++c) sure = sure || L.cut[m] == 2;
            atomicMin(&L.a, (sure ? 0 : 1 << 24) | k);
        }
    __syncthreads();
    if (t == 0) L.a &= 0xFFFFFF;
    __syncthreads();
}

template <int CAP, int NT, class RelFn>
__device__ void far_apply(FarLds<CAP, NT> &L, const P2 &C, int ctag, int ptag, RelFn rel)
{
    // cooperative version of ofl_dl::poly_clip
    __syncthreads();                                   // the previous application has been read by every thread
    const int t = threadIdx.x, n = L.n;
    const double h = 0.5 * (C.x * C.x + C.y * C.y);
    Poly P{ L.vx, L.vy, L.tag, 1, CAP, n };
    if (t == 0) { L.a = 0x7fffffff; L.ncut = 0; L.nstart = 0; }
    __syncthreads();
    int mine = 0;
    bool weak = false;
    for (int k = t; k < n; k += NT) {
        const int c = vertex_cut_ex(P, k, n, C, ctag, ptag, h, rel);
        L.cut[k] = (unsigned char)c; mine += c != 0; weak = weak || c == 1;
    }
    if (mine) atomicAdd(&L.ncut, mine);
    if (weak) L.nstart = 1;                            // (any vertex decided by the predicate or a tie rule: look at the runs below)
    __syncthreads();
    const int ncut = L.ncut;
    if (ncut == 0 || ncut == n) return;
    for (int k = t; k < n; k += NT)
        if (L.cut[k] && !L.cut[k == 0 ? n - 1 : k - 1]) atomicMin(&L.a, k);
    __syncthreads();
    if (L.nstart) far_pick_run(L);                    // (rare: a vertex decided by the predicate or a tie rule)
    const int a = L.a;
    int len = 0;
    while (L.cut[(a + len) % n]) ++len;               // every thread walks the (short) run: uniform result
    const int b = (a + len - 1) % n, ia = a == 0 ? n - 1 : a - 1, ib = (b + 1) % n;
    const int n2 = n - len + 2;
    if (n2 > CAP) { if (t == 0) L.status |= 1; return; }
    const int tb = L.tag[b];
    const P2 v1 = cut_point(L.tag[ia], C, h, rel, L.vx[ia], L.vy[ia], L.vx[a], L.vy[a]);
    const P2 v2 = cut_point(tb, C, h, rel, L.vx[b], L.vy[b], L.vx[ib], L.vy[ib]);
    __syncthreads();
    if (a <= b) {
        far_shift(L, b + 1, n, a + 2);
        if (t == 0) {
            L.vx[a] = v1.x; L.vy[a] = v1.y; L.tag[a] = ctag;
            L.vx[a + 1] = v2.x; L.vy[a + 1] = v2.y; L.tag[a + 1] = tb;
        }
    } else {
        const int m = a - (b + 1);
        far_shift(L, b + 1, a, 0);
        if (t == 0) {
            L.vx[m] = v1.x; L.vy[m] = v1.y; L.tag[m] = ctag;
            L.vx[m + 1] = v2.x; L.vy[m + 1] = v2.y; L.tag[m + 1] = tb;
        }
    }
    if (t == 0) { L.n = n2; ++L.napply; }
    __syncthreads();
    far_refresh(L);
}

// does candidate `cand` at q cut the cell as it stands?  (per thread; C receives its relative position)
template <int CAP, int NT, class RelFn>
__device__ __forceinline__ bool far_test(const FarLds<CAP, NT> &L, int p, const P2 &pp, int cand, const P2 &q, RelFn rel, P2 &C)
{
    bool hit = false;
    C = P2{ 0.0, 0.0 };
    if (cand >= 0 && cand != p) {
        C.x = q.x - pp.x; C.y = q.y - pp.y;
        const double d2 = C.x * C.x + C.y * C.y;
        if (d2 != 0.0) {
            const int n = L.n;
            const double h = 0.5 * d2;
            Poly P{ const_cast<double *>(L.vx), const_cast<double *>(L.vy), const_cast<int *>(L.tag), 1, CAP, n };
            bool all = d2 < L.reach2, some = !all && L.nfar != 0;
            if (some && L.cone) {
                // vmax max(d0 . c, d1 . c, 0) - |c|^2 / 2 < -(margin of vertex_cut) = -1e-9 (|vx cx| + |vy cy| + |c|^2 / 2)
                const double m0 = L.c0x * C.x + L.c0y * C.y, m1 = L.c1x * C.x + L.c1y * C.y;
                const bool inside = L.c0x * C.y - L.c0y * C.x >= 0.0 && C.x * L.c1y - C.y * L.c1x >= 0.0;
                if (!inside && fmax(fmax(m0, m1), 0.0) + 1e-9 * (fabs(C.x) + fabs(C.y)) < d2 * L.kcone) some = false;
            }
            if (some && L.nfar < 0) { all = true; some = false; }
            // Every vertex, or the far list: four at a time.  The first look of vertex_cut -- is v . c - h beyond the margin,
            // either way? -- needs a vertex and the two tags beside it; a loop that asks vertex_cut one vertex after the other
            // (and stops at the first cut) is a chain of dependent LDS reads, 100+ cycles each, which is what the sweeps of
            // these passes spent their time on.  Here the reads of four vertices are in flight together, the look is taken on
            // all four, and only a vertex inside the margin goes through vertex_cut itself (same decision: its own first look).
            const int cnt = all ? n : (some ? L.nfar : 0);
            for (int i0 = 0; i0 < cnt && !hit; i0 += 4) {
                int kk[4], ta[4], tb[4];
                double x[4], y[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) kk[j] = i0 + j < cnt ? (all ? i0 + j : L.farlist[i0 + j]) : -1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = kk[j] < 0 ? 0 : kk[j];
                    ta[j] = L.tag[k == 0 ? n - 1 : k - 1]; tb[j] = L.tag[k]; x[j] = L.vx[k]; y[j] = L.vy[k];
                }
                unsigned unsure = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kk[j] < 0 || cand == ta[j] || cand == tb[j]) continue;      // (the candidate already carries an edge at this vertex)
                    const double tx = x[j] * C.x, ty = y[j] * C.y, d = tx + ty - h, m = kDecide * (fabs(tx) + fabs(ty) + h);
                    if (d > m) hit = true;
                    else if (!(d < -m)) unsure |= 1u << j;
                }
                if (!hit && unsure) {
#pragma unroll 1
                    for (int j = 0; j < 4 && !hit; ++j)
                        if ((unsure >> j) & 1u) hit = vertex_cut(P, kk[j], n, C, cand, p, h, rel);
                }
            }
        }
    }
    return hit;
}

// K steps of up to NT candidates each (thread t holds candidates cand[0 .. K), or -1) under ONE vote per round.  Most steps
// cut nothing: they cost one barrier (the vote); the cell is only touched -- and the workgroup only synchronises further --
// when some candidate cuts it.  The candidates that do are applied NEAREST FIRST: they arrive in bucket or index order,
// and applied in that order a row of sites that approaches the cell's site clips it once per site (stripes, folds, a wavy
// image border: a step of 64 sites meant up to 64 cooperative clips of which the last few survive), whereas after the
// nearest ones most of the others no longer cut anything, which far_apply finds out with one cooperative pass over the
// vertices (a lane per vertex) and no clip.  A candidate that does not cut the cell now cannot cut the smaller cell
// later, so the flags taken before the first vote stay valid.
template <int K, int CAP, int NT, class RelFn>
__device__ void far_chunks(FarLds<CAP, NT> &L, int p, const P2 &pp, const int (&cand)[K], const P2 (&q)[K], RelFn rel)
{
    const int t = threadIdx.x;
    unsigned pend = 0;                                     // bit k: candidate k of this lane cuts the cell as it stands now
    double d2[K];
#pragma unroll
    for (int k = 0; k < K; ++k) { P2 C; if (far_test(L, p, pp, cand[k], q[k], rel, C)) pend |= 1u << k; d2[k] = C.x * C.x + C.y * C.y; }
    for (;;) {
        double best = 1e300;
        int bestk = -1;
#pragma unroll
        for (int k = 0; k < K; ++k) if (((pend >> k) & 1u) && d2[k] < best) { best = d2[k]; bestk = k; }
        // (every application below ends with a barrier: the cell is stable here.  A single wave votes with a ballot:
        // __syncthreads_or is a workgroup reduction through LDS, a hundred instructions for every step of every sweep)
        if (NT == 64 ? (__ballot(bestk >= 0) == 0ull) : !__syncthreads_or(bestk >= 0)) return;
        // the nearest cutting candidate of the workgroup: lowest lane of the wave minimum, lowest wave of equal minima
        double wmin = best;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            wmin = fmin(wmin, __hiloint2double(__shfl_xor(__double2hiint(wmin), off), __shfl_xor(__double2loint(wmin), off)));
        const unsigned long long eq = __ballot(bestk >= 0 && best == wmin);
        const int lead = (t & ~63) + (eq ? __ffsll((long long)eq) - 1 : 0);
        bool mine = t == lead && eq != 0;
        if (NT > 64) {
            if (mine) L.wmax[t >> 6] = wmin;
            if ((t & 63) == 0 && eq == 0) L.wmax[t >> 6] = 1e300;
            __syncthreads();
            int wbest = 0;
            for (int w = 1; w < NT / 64; ++w) if (L.wmax[w] < L.wmax[wbest]) wbest = w;
            mine = mine && (t >> 6) == wbest;
        }
        if (mine) {
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (k == bestk) { L.cidx[0] = cand[k]; L.ccx[0] = q[k].x - pp.x; L.ccy[0] = q[k].y - pp.y; }
            pend &= ~(1u << bestk);
        }
        __syncthreads();
        far_apply(L, P2{ L.ccx[0], L.ccy[0] }, L.cidx[0], p, rel);
    }
}

template <int CAP, int NT, class RelFn>
__device__ void far_chunk(FarLds<CAP, NT> &L, int p, const P2 &pp, int cand, const P2 &q, RelFn rel)
{
    const int c1[1] = { cand };
    const P2  q1[1] = { q };
    far_chunks<1>(L, p, pp, c1, q1, rel);
}

// The runs in L.run_lo / L.run_pre (lengths, turned into a prefix here) are walked as ONE dense list, NT candidates per
// step; `map` turns an entry of the sorted list into a point index.
template <int CAP, int NT, class RelFn>
__device__ void far_dense(FarLds<CAP, NT> &L, int p, const P2 &pp, int nruns, RelFn rel,
                          const unsigned *__restrict__ list_pt, const P2 *__restrict__ list_xy)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int r = 0; r < nruns; ++r) { const int c = L.run_pre[r]; L.run_pre[r] = acc; acc += c; }
        L.run_pre[nruns] = acc;
    }
    __syncthreads();
    const int total = L.run_pre[nruns];
    for (int base = 0; base < total; base += NT) {
        const int gidx = base + (int)threadIdx.x;
        int cand = -1;
        P2 q{ 0.0, 0.0 };
        if (gidx < total) {
            int r = 0;
            while (L.run_pre[r + 1] <= gidx) ++r;
            const unsigned j = L.run_lo[r] + (unsigned)(gidx - L.run_pre[r]);
            cand = (int)list_pt[j];
            q = list_xy[j];
        }
        far_chunk(L, p, pp, cand, q, rel);
    }
}

// one run = the buckets [x0, x1] of row `row` of grid g (empty when outside)
template <int CAP, int NT>
__device__ __forceinline__ void far_set_run(FarLds<CAP, NT> &L, int slot, const Grid &g, const unsigned *__restrict__ bstart,
                                            int row, int x0, int x1)
{
    unsigned lo = 0, cnt = 0;
    x0 = max(x0, 0); x1 = min(x1, g.gx - 1);
    if (row >= 0 && row < g.gy && x1 >= x0) {
        lo = bstart[(size_t)row * g.gx + x0];
        cnt = bstart[(size_t)row * g.gx + x1 + 1] - lo;
    }
    L.run_lo[slot] = lo; L.run_pre[slot] = (int)cnt;
}

// candidates of the fine buckets within kRings of the point's bucket
template <int CAP, int NT, class RelFn>
__device__ void far_near_rows(FarLds<CAP, NT> &L, int p, const P2 &pp, const Grid &g, const unsigned *__restrict__ bstart,
                              const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy, RelFn rel)
{
    const int t = threadIdx.x, bx = g.bx(pp.x), by = g.by(pp.y);
    __syncthreads();
    if (t <= 2 * kRings) far_set_run(L, t, g, bstart, (t & 1) ? by - ((t + 1) >> 1) : by + (t >> 1), bx - kRings, bx + kRings);      // (rows from the point's own outwards)
    far_dense(L, p, pp, 2 * kRings + 1, rel, sorted, sorted_xy);
}

// The unfinished points in the coarse cells at Chebyshev distance (ra, rb] from the point's cell (ra = -1: the whole square
// of radius rb): full rows above and below, the two side stretches of the rows in between; runs go through the run list
// 64 at a time.
template <int CAP, int NT, class RelFn>
__device__ void far_coarse_annulus(FarLds<CAP, NT> &L, int p, const P2 &pp, int ra, int rb, const Grid &g1,
                                   const unsigned *__restrict__ b1start, const unsigned *__restrict__ sorted1_pt,
                                   const P2 *__restrict__ sorted1_xy, RelFn rel)
{
    const int t = threadIdx.x, bx = g1.bx(pp.x), by = g1.by(pp.y);
    const int nt = rb - ra, nm = ra >= 0 ? 2 * ra + 1 : 0, total = 2 * nt + 2 * nm;
    for (int b0 = 0; b0 < total; b0 += 64) {
        const int cnt = min(64, total - b0);
        __syncthreads();
        if (t < cnt) {
            const int i = b0 + t;
            if (ra < 0) {
                // the whole square: rows from the point's own outwards, nearest first (from the top row down, the sites of a
                // stretch of border arrive APPROACHING the nearest one, and each of them clips the cell)
                const int k = i >> 1;
                far_set_run(L, t, g1, b1start, (i & 1) ? by - k : by + k, bx - rb, bx + rb);
            }
            else if (i < nt) far_set_run(L, t, g1, b1start, by - rb + i, bx - rb, bx + rb);
            else if (i < 2 * nt) far_set_run(L, t, g1, b1start, by + ra + 1 + (i - nt), bx - rb, bx + rb);
            else {
                const int m = i - 2 * nt, row = by - ra + (m >> 1);
                if (m & 1) far_set_run(L, t, g1, b1start, row, bx + ra + 1, bx + rb);
                else       far_set_run(L, t, g1, b1start, row, bx - rb, bx - ra - 1);
            }
        }
        far_dense(L, p, pp, cnt, rel, sorted1_pt, sorted1_xy);
    }
}

// The cell the per-thread pass left in the point's neighbour slots (the sites of its edges in cyclic order, box sides
// included): rebuilt in one step -- vertex k is where the lines of edges k - 1 and k meet, one lane each -- instead of one
// cooperative clip per site (eight clips of a dozen barriers each were a third of the workgroup pass).  Should a vertex
// not come out finite, the sites are applied as candidates the old way.
template <int CAP, int NT, class RelFn>
__device__ void far_seeds(FarLds<CAP, NT> &L, int p, const P2 &pp, const unsigned *__restrict__ nbr, const PosFn &pos, RelFn rel)
{
    const unsigned *sd = nbr + (size_t)p * kSlots;
    const int ns = min((int)sd[1], kSlots - 4), t = threadIdx.x;
    bool bad = ns < 3;
    double vx = 0.0, vy = 0.0;
    int tag = 0;
    if (t < ns && !bad) {
        double ax, ay, ah, bx, by, bh;
        tag = (int)sd[2 + t];
        edge_line((int)sd[2 + (t == 0 ? ns - 1 : t - 1)], rel, ax, ay, ah);
        edge_line(tag, rel, bx, by, bh);
        const double det = ax * by - bx * ay;
        vx = (ah * by - bh * ay) / det; vy = (ax * bh - bx * ah) / det;
        bad = det == 0.0 || !(isfinite(vx) && isfinite(vy)) || fabs(vx) > 4.0 * kBox || fabs(vy) > 4.0 * kBox;
    }
    if (NT == 64 ? (__ballot(bad) == 0ull) : !__syncthreads_or(bad)) {
        if (t < ns) { L.vx[t] = vx; L.vy[t] = vy; L.tag[t] = tag; }
        if (t == 0) L.n = ns;
        __syncthreads();
        far_refresh(L);
        return;
    }
    const int cand = t < ns ? (int)sd[2 + t] : -1;
    far_chunk(L, p, pp, cand, cand >= 0 ? pos(cand) : pp, rel);
}

template <int CAP, int NT>
__device__ void far_store(FarLds<CAP, NT> &L, unsigned rank, DlHead *head, unsigned *__restrict__ far_deg,
                          unsigned *__restrict__ far_off, int *__restrict__ pool, unsigned long long pool_cap, unsigned *s_off)
{
    const int t = threadIdx.x, n = L.n;
    if (t == 0) {
        *s_off = atomicAdd(&head->pool_used, (unsigned)n);
        if (L.status) atomicOr(&head->err, 1u);
    }
    __syncthreads();
    const unsigned off = *s_off;
    if ((unsigned long long)off + n > pool_cap) {
        if (t == 0) { atomicOr(&head->err, 2u); far_deg[rank] = 0; far_off[rank] = 0; }
        return;
    }
    for (int k = t; k < n; k += NT) pool[off + k] = L.tag[k];
    if (t == 0) { far_deg[rank] = (unsigned)n; far_off[rank] = off; }
}

__device__ void mid_point(FarLds<kMidCap, 64> &L, unsigned &s_off, unsigned rank,
                          const float *__restrict__ flow, int sign, int H, int W, DlHead *head,
                        const unsigned *__restrict__ bstart, const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy,
                        const unsigned *__restrict__ b1start, const unsigned *__restrict__ sorted1_pt, const P2 *__restrict__ sorted1_xy,
                        const unsigned *__restrict__ far_idx, const unsigned char *__restrict__ deg, const unsigned *__restrict__ nbr,
                        unsigned *__restrict__ far_deg, unsigned *__restrict__ far_off, int *__restrict__ pool, unsigned long long pool_cap,
                        unsigned char *__restrict__ far_wide)
{
    const int t = threadIdx.x;
    const int p = (int)far_idx[rank];
    if (deg[p] != kDegFar) return;                         // finished by the second per-thread pass
    // Points of the image border are hull points more often than not: their cells never close, whatever this pass applies
    // (it used to give them kMidRings rings, which the next pass then applied again).  They go straight to the left-over list.
    const int py = (int)((unsigned)p / (unsigned)W), px = p - py * W;
    if (px == 0 || py == 0 || px == W - 1 || py == H - 1) {
        if (t == 0) { far_deg[rank] = kDegLeft; far_wide[rank] = 0; }
        return;
    }
    const Grid g = head->grid, g1 = head->grid1;
    const PosFn pos(flow, sign, W);
    const P2 pp = pos(p);
    auto rel = [&](int q) { const P2 v = pos(q); return P2{ v.x - pp.x, v.y - pp.y }; };
    if (t == 0) {
        Poly P{ L.vx, L.vy, L.tag, 1, kMidCap, 0 };
        poly_init(P);
        L.n = P.n; L.status = 0; L.t2 = head->far_t2;
    }
    __syncthreads();
    far_refresh(L);
    far_seeds(L, p, pp, nbr, pos, rel);
    far_near_rows(L, p, pp, g, bstart, sorted, sorted_xy, rel);
    // Rings of the coarse grid until the cell is final: every unfinished point within twice its farthest vertex has been
    // applied -- points in unvisited coarse cells are at least r * s1 away (finished points farther than the fine rings
    // cannot be neighbours: they would not have finished) -- or the rings have left the grid (rims of holes and tears: their
    // cells close once the far side has been applied).
    const int rmax = kMidRingsMax;
    const int cbx = g1.bx(pp.x), cby = g1.by(pp.y);
    const int rgrid = max(max(cbx, g1.gx - 1 - cbx), max(cby, g1.gy - 1 - cby));
    bool done = false;
    int ra = -1;                                           // the last ring applied
    for (; ra < rmax && !done;) {
        // ring by ring while the cell is small; then annuli that grow by half their radius (one pass over their rows
        // instead of one per ring: the rim of a large hole needs a hundred rings)
        const int rb = ra < kMidRings ? ra + 1 : min(ra + max(ra / 2, 1), rmax);
        far_coarse_annulus(L, p, pp, ra, rb, g1, b1start, sorted1_pt, sorted1_xy, rel);
        double r2 = 0.0;
        const int n = L.n;
        for (int k = t; k < n; k += 64) r2 = fmax(r2, L.vx[k] * L.vx[k] + L.vy[k] * L.vy[k]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            r2 = fmax(r2, __hiloint2double(__shfl_xor(__double2hiint(r2), off), __shfl_xor(__double2loint(r2), off)));
        const double cover = (double)rb * g1.s;
        done = cover * cover >= 4.0 * r2 || rb >= rgrid;
        if (L.status) break;
        ra = rb;
    }
    __syncthreads();
    if (t == 0) far_wide[rank] = (done && !L.status && ra > kMidRings) ? 1 : 0;
    if (!done || L.status) { if (t == 0) far_deg[rank] = kDegLeft; return; }
    far_store(L, rank, head, far_deg, far_off, pool, pool_cap, &s_off);
}

// one wave per unfinished point; the grid is fixed and walks the ranks (their number stays on the device)
__global__ __launch_bounds__(64, 4)                        // (the pass waits on loads and votes: four waves per SIMD)
void dl_star_mid_kernel(const float *__restrict__ flow, int sign, int H, int W, DlHead *head,
                        const unsigned *__restrict__ bstart, const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy,
                        const unsigned *__restrict__ b1start, const unsigned *__restrict__ sorted1_pt, const P2 *__restrict__ sorted1_xy,
                        const unsigned *__restrict__ far_idx, const unsigned char *__restrict__ deg, const unsigned *__restrict__ nbr,
                        unsigned *__restrict__ far_deg, unsigned *__restrict__ far_off, int *__restrict__ pool, unsigned long long pool_cap,
                        unsigned char *__restrict__ far_wide)
{
    __shared__ FarLds<kMidCap, 64> L;
    __shared__ unsigned s_off;
    const unsigned n_far = head->n_far;
    for (unsigned rank = blockIdx.x; rank < n_far; rank += gridDim.x) {
        __syncthreads();                                   // the previous point's cell has been stored by every thread
        mid_point(L, s_off, rank, flow, sign, H, W, head, bstart, sorted, sorted_xy, b1start, sorted1_pt, sorted1_xy, far_idx, deg, nbr,
                  far_deg, far_off, pool, pool_cap, far_wide);
    }
}

template <int CAP, int NT>
__device__ void far_point(FarLds<CAP, NT> &L, unsigned &s_off, unsigned li, unsigned n_left,
                          const float *__restrict__ flow, int sign, int H, int W, DlHead *head,
                        const unsigned *__restrict__ bstart, const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy,
                        const unsigned *__restrict__ b1start, const unsigned *__restrict__ sorted1_pt, const P2 *__restrict__ sorted1_xy,
                        const unsigned *__restrict__ far_idx, const unsigned *__restrict__ left_idx,
                        const unsigned *__restrict__ left_pt, const P2 *__restrict__ left_xy, const double *__restrict__ left_box,
                        const unsigned *__restrict__ nbr, unsigned *__restrict__ far_deg, unsigned *__restrict__ far_off, int *__restrict__ pool,
                        unsigned long long pool_cap)
{
    const int t = threadIdx.x;
    const unsigned rank = left_idx[li];
    const int p = (int)left_pt[li];                      // (= far_idx[rank], and its position: one round trip instead of three)
    const P2 pp = left_xy[li];
    if (far_deg[rank] != kDegLeft) return;               // finished by the pass with the smaller capacity
    const Grid g = head->grid, g1 = head->grid1;
    const PosFn pos(flow, sign, W);
    auto rel = [&](int q) { const P2 v = pos(q); return P2{ v.x - pp.x, v.y - pp.y }; };
    if (t == 0) {
        Poly P{ L.vx, L.vy, L.tag, 1, CAP, 0 };
        poly_init(P);
        L.n = P.n; L.status = 0; L.t2 = head->far_t2; L.napply = 0;
    }
    __syncthreads();
    far_refresh(L);
    far_seeds(L, p, pp, nbr, pos, rel);
    far_near_rows(L, p, pp, g, bstart, sorted, sorted_xy, rel);
    far_coarse_annulus(L, p, pp, -1, kMidRings, g1, b1start, sorted1_pt, sorted1_xy, rel);
#ifdef OFL_EXPERIMENTS
    __syncthreads();
    const int dbg_before = L.napply;
    unsigned dbg_chunks = 0;
#endif
    // (Round 4 measured five more forms of this pass, each bit-identical in its results, none faster -- profiles/HISTORY.md: the
    // chunks nearest first with the box test repeated before every chunk (+ 2 .. 7 %); a chunk's 256 candidates in one round trip
    // under one vote (+ 15 .. 20 %); 5 and 6 waves per SIMD instead of 4 (no change); the sites in another order (no change);
    // scan first, clip afterwards -- every candidate only TESTED against the cell as it stands, loads of four to eight steps in
    // flight, the few cutters noted in LDS, sorted by distance and then clipped (+ 20 .. 60 %; with far_test / far_apply /
    // far_chunks out of line, a third of the code size: + 30 .. 100 %).  PMC: 7 000 vector + 5 000 scalar + 970 LDS + 120 memory
    // instructions per site, 81 % of the wave cycles waiting, 1 500 of 4 096 wave slots occupied on average.)
    // Every other left-over point -- but only the chunks of 256 whose bounding box could hold a site that cuts the cell AS IT
    // STANDS (later the cell only shrinks: its reach falls, its far vertices stay inside the present cone and below the present
    // largest distance, so what cannot cut now cannot cut later).  The test is far_test's, applied to a box: nearer than the
    // reach of the near vertices, or not excluded by the far cone.
    const unsigned n_chunks = (n_left + 255) / 256;
    // could chunk ck hold a site that cuts the cell as it stands now?
    auto chunk_keep = [&](unsigned ck) -> bool {
        bool keep = false;
#pragma unroll 1
        for (int half = 0; half < 2 && !keep; ++half) {
            const double *b = left_box + ((size_t)ck * 2 + half) * 8;
            const double ax = b[2], ay = b[3], t0 = b[4], t1 = b[5], s0 = b[6], s1 = b[7];
            if (!(t0 <= t1)) continue;                       // empty half
            const double rx = b[0] - pp.x, ry = b[1] - pp.y;              // the box origin seen from the site
            const double tp = -(rx * ax + ry * ay), sp = -(ry * ax - rx * ay);        // the site in the box frame
            const double dt = fmax(fmax(t0 - tp, tp - t1), 0.0), ds = fmax(fmax(s0 - sp, sp - s1), 0.0), dmin2 = dt * dt + ds * ds;
            if (dmin2 < L.reach2) { keep = true; break; }
            if (L.nfar == 0) continue;
            if (!L.cone) { keep = true; break; }
            double mmax = 0.0, amax = 0.0;
            bool right0 = true, left1 = true;                // every corner strictly outside one side of the cone?
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double tt = (k & 1) ? t1 : t0, ss = (k & 2) ? s1 : s0;
                const double cx = rx + tt * ax - ss * ay, cy = ry + tt * ay + ss * ax;
                mmax = fmax(mmax, fmax(L.c0x * cx + L.c0y * cy, L.c1x * cx + L.c1y * cy));
                amax = fmax(amax, fabs(cx) + fabs(cy));
                right0 = right0 && (L.c0x * cy - L.c0y * cx < 0.0);
                left1 = left1 && (cx * L.c1y - cy * L.c1x < 0.0);
            }
            // (the corners are rounded: a margin of 1e-9 of their size on the dot products, as on the sites themselves)
            if (!((right0 || left1) && mmax + 2e-9 * amax < dmin2 * L.kcone)) keep = true;
        }
        return keep;
    };
    for (unsigned cbase = 0; cbase < n_chunks; cbase += NT) {
        __syncthreads();
        const unsigned ck = cbase + t;
        const bool keep = ck < n_chunks && chunk_keep(ck);
        // ordered compaction of the kept chunk numbers of this group
        const unsigned long long bal = __ballot(keep);
        if ((t & 63) == 0) L.hit[t >> 6] = bal;
        __syncthreads();
        unsigned before = 0, total = 0;
        for (int w = 0; w < NT / 64; ++w) { const unsigned c = (unsigned)__popcll(L.hit[w]); if (w < (t >> 6)) before += c; total += c; }
        if (keep) L.clist[before + (unsigned)__popcll(bal & ((1ull << (t & 63)) - 1ull))] = ck;
        __syncthreads();
#ifdef OFL_EXPERIMENTS
        dbg_chunks += total;
#endif
        constexpr int kVote = NT == 64 ? 1 : 4;            // steps under one vote (a wave's own vote is cheap)
        constexpr int kPer = 256 / NT;                     // steps of NT candidates per chunk of 256
        for (unsigned k0 = 0; k0 < total * kPer; k0 += kVote) {
            int cand[kVote];
            P2  q[kVote];
            unsigned cks[kVote];
#pragma unroll
            for (int k = 0; k < kVote; ++k) cks[k] = (k0 + k) / kPer < total ? L.clist[(k0 + k) / kPer] : 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < kVote; ++k) {
                const unsigned j = cks[k] * 256u + ((k0 + k) % kPer) * NT + t;
                const bool in = cks[k] != 0xFFFFFFFFu && j < n_left;
                cand[k] = in ? (int)left_pt[j] : -1;
                q[k] = in ? left_xy[j] : pp;
            }
            far_chunks<kVote>(L, p, pp, cand, q, rel);
        }
    }
    __syncthreads();
#ifdef OFL_EXPERIMENTS
    if (t == 0 && NT == 64) {
        atomicAdd(&head->dbg[0], 1u); atomicAdd(&head->dbg[1], dbg_chunks); atomicAdd(&head->dbg[2], (unsigned)dbg_before);
        atomicAdd(&head->dbg[3], (unsigned)(L.napply - dbg_before)); atomicAdd(&head->dbg[4], (unsigned)L.n);
        atomicMax(&head->dbg[5], dbg_chunks); if (dbg_chunks > 12) atomicAdd(&head->dbg[6], 1u); atomicMax(&head->dbg[7], (unsigned)L.napply);
    }
#endif
    if (CAP < kFarCap && L.status) return;               // overflow of the small capacity: far_deg stays kDegLeft for the next pass
    far_store(L, rank, head, far_deg, far_off, pool, pool_cap, &s_off);
}

#ifndef OFL_FAR_WAVES
#define OFL_FAR_WAVES 4
#endif
#ifndef OFL_FAR_CAP
#define OFL_FAR_CAP 384
#endif
template <int CAP, int NT>          // first single waves with a small cell capacity, then -- for the few fans that overflowed it -- workgroups with the large one
__global__ __launch_bounds__(NT, NT == 64 ? OFL_FAR_WAVES : 1)
void dl_star_far_kernel(const float *__restrict__ flow, int sign, int H, int W, DlHead *head,
                        const unsigned *__restrict__ bstart, const unsigned *__restrict__ sorted, const P2 *__restrict__ sorted_xy,
                        const unsigned *__restrict__ b1start, const unsigned *__restrict__ sorted1_pt, const P2 *__restrict__ sorted1_xy,
                        const unsigned *__restrict__ far_idx, const unsigned *__restrict__ left_idx,
                        const unsigned *__restrict__ left_pt, const P2 *__restrict__ left_xy, const double *__restrict__ left_box,
                        const unsigned *__restrict__ nbr, unsigned *__restrict__ far_deg, unsigned *__restrict__ far_off, int *__restrict__ pool,
                        unsigned long long pool_cap)
{
    __shared__ FarLds<CAP, NT> L;
    __shared__ unsigned s_off;
    const unsigned n_left = head->n_left;
    for (unsigned lb = blockIdx.x; lb < n_left; lb += gridDim.x) {
        __syncthreads();
        const unsigned li = lb;
        far_point<CAP, NT>(L, s_off, li, n_left, flow, sign, H, W, head, bstart, sorted, sorted_xy, b1start, sorted1_pt, sorted1_xy, far_idx,
                       left_idx, left_pt, left_xy, left_box, nbr, far_deg, far_off, pool, pool_cap);
    }
}

// ------------------------------------------------------------------------------------------------ raster
struct TriRef { unsigned i0, i1, i2; bool ok; };

__device__ __forceinline__ TriRef dl_decode(unsigned id, unsigned far_base, const DlWs &ws)
{
    TriRef r;
    r.ok = false;
    if (id < far_base) {
        const unsigned p = id / kSlots, k = id % kSlots, d = ws.deg[p];
        if (d > kSlots || k >= d) return r;
        r.i0 = p; r.i1 = ws.nbr[(size_t)p * kSlots + k]; r.i2 = ws.nbr[(size_t)p * kSlots + (k + 1 == d ? 0 : k + 1)];
        r.ok = true;
    } else {
        const unsigned rank = (id - far_base) / kFarK, k = (id - far_base) % kFarK, d = ws.far_deg[rank];
        if (d == kDegLeft || k >= d) return r;
        const unsigned off = ws.far_off[rank];
        const int a = ws.pool[off + k], b = ws.pool[off + (k + 1 == d ? 0 : k + 1)];
        if (a < 0 || b < 0 || a == b) return r;
        r.i0 = ws.far_idx[rank]; r.i1 = (unsigned)a; r.i2 = (unsigned)b;
        r.ok = true;
    }
    canonical3(r.i0, r.i1, r.i2);
    return r;
}

__device__ __forceinline__ TriBox box_rows(const D2 &p0, const D2 &p1, const D2 &p2, int W, int H, const DlWs &ws)
{
    TriBox b = tri_box(p0, p1, p2, W, H);
    b.y0 = max(b.y0, ws.oy0);
    b.y1 = min(b.y1, ws.oy1 - 1);
    return b;
}

__device__ __forceinline__ D2 pt(const PosFn &pos, unsigned i) { const P2 p = pos((int)i); return D2{ p.x, p.y }; }

// does the star of site s list b right after a (cyclically)?  Stars of more than 64 neighbours are not searched.
__device__ __forceinline__ bool star_has(const DlWs &ws, unsigned s, unsigned a, unsigned b)
{
    const unsigned d = ws.deg[s];
    if (d <= kSlots) {
        // the whole row in four 16-byte loads that are in flight together (a loop that stops at the match is a chain of
        // dependent 4-byte loads: this look-up is most of the small-triangle raster's time)
        const uint4 *row = reinterpret_cast<const uint4 *>(ws.nbr + (size_t)s * kSlots);
        const uint4 r0 = row[0], r1 = row[1], r2 = row[2], r3 = row[3];
        const unsigned v[kSlots] = { r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y, r3.z, r3.w };
        bool has = false;
#pragma unroll
        for (int k = 0; k < kSlots; ++k) {
            const unsigned nxt = (unsigned)(k + 1) < d ? v[(k + 1) & (kSlots - 1)] : v[0];      // (k + 1 == d == 16 wraps to v[0] either way)
            has = has || ((unsigned)k < d && v[k] == a && nxt == b);
        }
        return has;
    }
    if (d != kDegFar) return false;
    const unsigned rank = ws.nbr[(size_t)s * kSlots], fd = ws.far_deg[rank];
    if (fd == kDegLeft || fd > 64) return false;
    const int *pl = ws.pool + ws.far_off[rank];
    for (unsigned k = 0; k < fd; ++k)
        if (pl[k] == (int)a) return pl[k + 1 == fd ? 0 : k + 1] == (int)b;
    return false;
}

// scan conversion of ONE triangle by one thread (vertices in canonical rotation: every copy sets up the same edge functions);
// large bounding boxes go to the list swept by whole waves
__device__ __forceinline__ void raster_tri(unsigned id, const D2 &q0, const D2 &q1, const D2 &q2, int H, int W, const DlWs &ws)
{
    const TriBox b = box_rows(q0, q1, q2, W, H, ws);
    if (b.x1 < b.x0 || b.y1 < b.y0) return;
    const long long area = (long long)(b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1);
    if (area > kSmallArea) {
        const unsigned long long slot = atomicAdd(&ws.head->big_n, 1ull);
        if (slot < ws.big_cap) ws.big[slot] = id; else atomicOr(&ws.head->err, 4u);
        return;
    }
    TriEdge te;
    if (!tri_setup(q0, q1, q2, te)) return;
    for (int gy = b.y0; gy <= b.y1; ++gy)
        for (int gx = b.x0; gx <= b.x1; ++gx)
            if (tri_inside(te, (double)gx, (double)gy)) atomicMin(&ws.owner[(size_t)gy * W + gx], id);
    // (per-row intervals as in dl_raster_big_kernel were measured here for the wide boxes of sheared lattices: the extra
    // registers and code slow every field down by 4 - 10 %, config 5 included)
}

// one triangle of a star by one thread.  Every triangle is listed by
// each of its three sites; the copy of the site with the smallest index is the one that is drawn -- unless that
// site's star does not list the triangle (stars that disagree on a co-circular cell), in which case this copy is drawn too.
// `trust`: bit 0 = `self` was settled by the mesh-cell pass (its neighbours are grid neighbours), bits 1 .. 4 = so were its
// grid neighbours NW, N, NE, W -- the only ones with a smaller index.  A star that came from the cell flags lists every
// triangle of its four (verified) cells, so the copy held by such a neighbour IS drawn and this one need not look it up.
__device__ __forceinline__ void thread_raster(unsigned id, unsigned self, const PosFn &pos, int H, int W, const DlWs &ws, unsigned far_base,
                                              unsigned trust = 0)
{
    const TriRef tr = dl_decode(id, far_base, ws);
    if (!tr.ok) return;
    if (tr.i0 != self) {
        bool skip = false;
        if (trust & 1u) {
            const int dv = (int)self - (int)tr.i0;
            skip = (dv == W + 1 && (trust & 2u)) || (dv == W && (trust & 4u)) || (dv == W - 1 && (trust & 8u)) || (dv == 1 && (trust & 16u));
        }
        if (skip || star_has(ws, tr.i0, tr.i1, tr.i2)) return;
    }
    raster_tri(id, pt(pos, tr.i0), pt(pos, tr.i1), pt(pos, tr.i2), H, W, ws);
}

// Cell-centric raster of the intact parts of the mesh: one thread per grid cell of a clean tile draws the cell's two
// triangles -- corner positions from two coalesced rows of the flow, no star to decode, no look-up into another site's star --
// with exactly the ids the site-wise raster gives them: a triangle belongs to its smallest-index vertex s, and when s is a
// clean site (site_clean) its position k in s's star follows from s's own cell flags:
//   diagonal a - c:  (a, b, c) = a's triangle 0 (E, SE),  (a, c, d) = a's triangle 1 (SE, S)
//   diagonal b - d:  (a, b, d) = a's triangle 0 (E, S),   (b, c, d) = b's triangle (S, SW): k = 1 + [b's own cell has diagonal a - c]
// (a = (x, y), b = (x + 1, y), c = (x + 1, y + 1), d = (x, y + 1): a < b < d < c in index order).  Triangles whose owner is
// not clean -- rims of holes and tears, neighbourhoods of dropped points -- stay with the site-wise raster, which in turn
// returns at once for clean sites: whole waves of it, since cleanliness goes by tiles.  An intact mesh with a hole or a
// moving object in it is drawn almost entirely here (4K: 0.49 -> 0.27 ms); on a field that is broken all over (speckled
// point masks, BASELINE config 5) no tile is clean and this kernel ends after one byte per workgroup.
__global__ __launch_bounds__(256)
void dl_raster_cells_kernel(const float *__restrict__ flow, int sign, int H, int W, DlWs ws, unsigned far_base, int tiles_x, int ntiles)
{
    const unsigned per = gridDim.x >> 3;                        // (grid padded to a multiple of 8) one contiguous eighth per XCD
    const int tile = (int)((blockIdx.x & 7u) * per + (blockIdx.x >> 3));
    if (tile >= ntiles || !ws.tileflag[tile]) return;           // (the owner of a triangle is clean only if the cell's own tile is)
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x = tx * 32 + (threadIdx.x & 31), y = ty * 8 + (threadIdx.x >> 5);
    const size_t ia = (size_t)y * W + x;
    const unsigned f = ws.cellflag[ia];                         // (a clean tile lies inside the grid, all flags non-zero)
    const bool sa = site_clean(ws.tileflag, x, y, W, H, tiles_x);
    const bool sb = f == 2 && site_clean(ws.tileflag, x + 1, y, W, H, tiles_x);
    if (!sa && !sb) return;
    const D2 a = point_of(flow, sign, W, x, y), b = point_of(flow, sign, W, x + 1, y);
    const D2 c = point_of(flow, sign, W, x + 1, y + 1), d = point_of(flow, sign, W, x, y + 1);
    if (sa) {
        raster_tri((unsigned)ia * kSlots, a, b, f == 1 ? c : d, H, W, ws);
        if (f == 1) raster_tri((unsigned)ia * kSlots + 1u, a, c, d, H, W, ws);
    }
    if (sb) raster_tri((unsigned)(ia + 1) * kSlots + 1u + (ws.cellflag[ia + 1] == 1 ? 1u : 0u), b, c, d, H, W, ws);
}

__global__ __launch_bounds__(256)
void dl_raster_small_kernel(const float *__restrict__ flow, int sign, int H, int W, DlWs ws, unsigned far_base, const unsigned *__restrict__ list)
{
    // The sites that are not clean, compacted in index order (dl_compact_kernel<4>): on a field with a few tears or holes they
    // are the rims -- one or two sites of 64 -- and a wave that also held the 62 clean ones ran as long as its rim sites took
    // (64-px stripes at 4K: 0.93 ms, every wave crossing a tear).
    // Workgroups go round-robin over the 8 XCDs: give each XCD one contiguous eighth of the list, so that the neighbour
    // rows a point looks up (the stars of the sites above and below it) are in ITS L2
    // (a row band draws from the sites around its rows only -- often one contiguous stretch of indices, which that mapping
    // would hand to a single XCD: 3.1 ms instead of 0.5 for an eighth of config 5 -- so bands keep the round-robin order)
    const unsigned n_list = ws.head->n_raster;
    const unsigned nb = (n_list + 255u) / 256u, per = (nb + 7u) / 8u;
    const PosFn pos(flow, sign, W);
    for (unsigned blk0 = blockIdx.x; blk0 < per * 8u; blk0 += gridDim.x) {
        const unsigned blk = ws.oy1 - ws.oy0 < H ? blk0 : (blk0 & 7u) * per + (blk0 >> 3);
        const unsigned i = blk * 256u + threadIdx.x;
        if (blk >= nb || i >= n_list) continue;
        const size_t p = list[i];
        const unsigned d = ws.deg[p];
        // which of this site and its four smaller-index grid neighbours were settled from the cell flags (three dword loads)
        unsigned trust = 0;
        {
            const int y = (int)(p / (unsigned)W), x = (int)(p - (size_t)y * W);
            if (x >= 2 && y >= 2 && x <= W - 3 && y <= H - 2) {
                const unsigned char *f = ws.cellflag + (size_t)(y - 2) * W + (x - 2);
                struct __attribute__((packed, aligned(1))) U32u { unsigned v; };
                const unsigned r0 = reinterpret_cast<const U32u *>(f)->v, r1 = reinterpret_cast<const U32u *>(f + W)->v,
                               r2 = reinterpret_cast<const U32u *>(f + 2 * (size_t)W)->v;
                auto nz = [](unsigned r, int k) { return ((r >> (8 * k)) & 0xFFu) != 0u; };
                if (nz(r2, 2) && nz(r2, 1) && nz(r1, 1) && nz(r1, 2)) {
                    trust = 1u;
                    if (nz(r1, 1) && nz(r1, 0) && nz(r0, 0) && nz(r0, 1)) trust |= 2u;       // NW
                    if (nz(r1, 2) && nz(r1, 1) && nz(r0, 1) && nz(r0, 2)) trust |= 4u;       // N
                    if (nz(r1, 3) && nz(r1, 2) && nz(r0, 2) && nz(r0, 3)) trust |= 8u;       // NE
                    if (nz(r2, 1) && nz(r2, 0) && nz(r1, 0) && nz(r1, 1)) trust |= 16u;      // W
                }
            }
        }
        for (unsigned k = 0; k < d; ++k) thread_raster((unsigned)p * kSlots + k, (unsigned)p, pos, H, W, ws, far_base, trust);
    }
}

// one WAVE per unfinished point (their stars run to hundreds of triangles: a thread per point would crawl)
__global__ __launch_bounds__(256)
void dl_raster_far_kernel(const float *__restrict__ flow, int sign, int H, int W, DlWs ws, unsigned far_base)
{
    const PosFn pos(flow, sign, W);
    const unsigned n_far = ws.head->n_far;
    for (unsigned rank = blockIdx.x * 4 + (threadIdx.x >> 6); rank < n_far; rank += gridDim.x * 4) {
        const unsigned d = ws.far_deg[rank];
        if (d == kDegLeft) continue;
        const unsigned self = ws.far_idx[rank];
        for (unsigned k = threadIdx.x & 63; k < d; k += 64) thread_raster(far_base + rank * kFarK + k, self, pos, H, W, ws, far_base);
    }
}

// One wave per large triangle, row by row: the three edge functions are linear in x, so each row's covered interval is
// solved for (padded by a node on each side; tri_inside still decides every node) and only that interval is walked.  A
// sliver from a hull point to a far outlier has the whole frame as its bounding box and covers a handful of nodes:
// box-wise it cost 130 000 steps at 4K, row-wise 2 160.
__global__ __launch_bounds__(256)
void dl_raster_big_kernel(const float *__restrict__ flow, int sign, int H, int W, DlWs ws, unsigned far_base)
{
    unsigned long long n = ws.head->big_n;
    if (n > ws.big_cap) n = ws.big_cap;
    const PosFn pos(flow, sign, W);
    const int lane = threadIdx.x & 63;
    for (unsigned long long j = (unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6); j < n; j += (unsigned long long)gridDim.x * 4) {
        const unsigned id = ws.big[j];
        const TriRef tr = dl_decode(id, far_base, ws);
        if (!tr.ok) continue;
        const D2 q0 = pt(pos, tr.i0), q1 = pt(pos, tr.i1), q2 = pt(pos, tr.i2);
        const TriBox b = box_rows(q0, q1, q2, W, H, ws);
        if (b.x1 < b.x0 || b.y1 < b.y0) continue;
        TriEdge te;
        if (!tri_setup(q0, q1, q2, te)) continue;
        // w_k(dx, dy) = A_k dx + B_k(dy) >= -tol with dx = gx - p0.x, the sign of det folded in (tri_inside)
        const double sg = te.det < 0 ? -1.0 : 1.0;
        const double A1 = sg * te.e2y, A2 = -sg * te.e1y, A0 = sg * (te.e1y - te.e2y);
        for (int gy = b.y0; gy <= b.y1; ++gy) {
            const double dy = (double)gy - te.p0.y;
            const double B1 = -sg * te.e2x * dy, B2 = sg * te.e1x * dy, B0 = sg * (te.det + (te.e2x - te.e1x) * dy);
            double lo = -1e300, hi = 1e300;
            const double Ak[3] = { A0, A1, A2 }, Bk[3] = { B0, B1, B2 };
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                // the bound of an almost horizontal edge (tiny A) is ill-conditioned: its slack grows with the rounding of
                // B / A, so that no node tri_inside accepts can fall outside (a wide slack merely restricts nothing)
                const double bnd = (-te.tol - Bk[k]) / Ak[k];
                const double slack = 1.0 + 8e-15 * ((fabs(Bk[k]) + te.tol) / fabs(Ak[k]) + fabs(bnd) + fabs(te.p0.x));
                if (Ak[k] > 0.0) lo = fmax(lo, bnd - slack);
                else if (Ak[k] < 0.0) hi = fmin(hi, bnd + slack);
            }
            // the interval in node coordinates (NaN-safe: an unordered compare keeps the box)
            const double xl = lo + te.p0.x, xh = hi + te.p0.x;
            int xa = b.x0, xb = b.x1;
            if (xl > (double)xa) xa = xl >= (double)xb + 1.0 ? xb + 1 : (int)floor(xl);
            if (xh < (double)xb) xb = xh <= (double)xa - 1.0 ? xa - 1 : (int)ceil(xh);
            for (int gx = xa + lane; gx <= xb; gx += 64)
                if (tri_inside(te, (double)gx, (double)gy)) atomicMin(&ws.owner[(size_t)gy * W + gx], id);
        }
    }
}

// ------------------------------------------------------------------------------------------------ resolve
template <typename VT>
__global__ __launch_bounds__(256)
void dl_resolve_kernel(const float *__restrict__ flow, int sign, const VT *__restrict__ vals, int C,
                       const uint8_t *__restrict__ vmask, int H, int W, int row0, int rows,
                       VT *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule, DlWs ws, unsigned far_base)
{
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int yl = blockIdx.y * 8 + (threadIdx.x >> 5), y = row0 + yl;
    if (x >= W || yl >= rows) return;
    const size_t o = (size_t)yl * W + x;
    const unsigned id = ws.owner[(size_t)y * W + x];
    TriRef tr;
    tr.ok = false;
    if (id != kNoOwner) tr = dl_decode(id, far_base, ws);
    if (!tr.ok) {
        for (int c = 0; c < C; ++c) out[o * C + c] = (VT)0;                  // outside the convex hull: NaN -> 0, utils.py:254
        if (valid) valid[o] = 0;
        return;
    }
    const PosFn pos(flow, sign, W);
    const size_t vi[3] = { tr.i0, tr.i1, tr.i2 };
    double c0, c1, c2;
    (void)bary(pt(pos, tr.i0), pt(pos, tr.i1), pt(pos, tr.i2), (double)x, (double)y, c0, c1, c2);
    resolve_emit(vals, C, vmask, vi, c0, c1, c2, valid_rule, out, valid, o);
}

// ------------------------------------------------------------------------------------------------ arbitrary positions
// neighbour k of site u (cyclic); -1 = an unbounded gap of the star (or no star)
struct StarRef { const unsigned *small; const int *far; unsigned d; };

__device__ __forceinline__ StarRef star_of(const DlWs &ws, unsigned u)
{
    StarRef r;
    r.small = nullptr; r.far = nullptr; r.d = 0;
    const unsigned d = ws.deg[u];
    if (d >= 1 && d <= kSlots) { r.small = ws.nbr + (size_t)u * kSlots; r.d = d; }
    else if (d == kDegFar) {
        const unsigned rank = ws.nbr[(size_t)u * kSlots], fd = ws.far_deg[rank];
        if (fd != kDegLeft && fd > 0) { r.far = ws.pool + ws.far_off[rank]; r.d = fd; }
    }
    return r;
}

__device__ __forceinline__ int star_at(const StarRef &s, unsigned k) { return s.small ? (int)s.small[k] : s.far[k]; }

// Last resort of dl_locate: the stars of the sites in the bucket rings around the position, triangle by triangle.  The walk
// below needs stars that AGREE along its way (the edge it leaves through must be listed by the star it enters); where four
// sites are co-circular they need not -- on an integer-valued field that is every cell -- and a walk that meets such a pair
// used to give up: 0.17 % of the positions of mode 2 't' on such fields came back "outside" although SciPy finds them inside
// (tools/soak_scatter.py --mode query).  Whatever the stars disagree on, their triangles together cover the hull, and none
// spans an unbounded gap: a position that lies in no triangle of any site within `rings` buckets IS outside (or in a triangle
// larger than that: rims of holes -- the walk finds those, they are not what it stumbles over).
__device__ bool dl_locate_brute(const DlWs &ws, const PosFn &pos, double qx, double qy, int rings, TriRef &tr, double &c0, double &c1, double &c2)
{
    const Grid g = ws.head->grid;
    const int bx = g.bx(qx), by = g.by(qy);
    const D2 q{ qx, qy };
    for (int r = 0; r <= rings; ++r)
        for (int row = by - r; row <= by + r; ++row) {
            if (row < 0 || row >= g.gy) continue;
            const bool full = row == by - r || row == by + r;
            for (int part = 0; part < (full ? 1 : 2); ++part) {
                int x0 = full ? bx - r : (part ? bx + r : bx - r), x1 = full ? bx + r : x0;
                if (!full && (x0 < 0 || x0 >= g.gx)) continue;
                x0 = max(x0, 0); x1 = min(x1, g.gx - 1);
                if (x1 < x0) continue;
                for (unsigned j = ws.bstart[(size_t)row * g.gx + x0]; j < ws.bstart[(size_t)row * g.gx + x1 + 1]; ++j) {
                    const unsigned sidx = ws.sorted[j];
                    if (sidx == 0xFFFFFFFFu) continue;
                    const StarRef ss = star_of(ws, sidx);
                    if (ss.d < 2) continue;
                    const D2 pa = pt(pos, sidx);
                    int n0 = star_at(ss, ss.d - 1);
                    D2 p0 = n0 >= 0 ? pt(pos, (unsigned)n0) : D2{ 0.0, 0.0 };
                    for (unsigned k = 0; k < ss.d; ++k) {
                        const int n1 = star_at(ss, k);
                        const D2 p1 = n1 >= 0 ? pt(pos, (unsigned)n1) : D2{ 0.0, 0.0 };
                        if (n0 >= 0 && n1 >= 0 && n0 != n1) {
                            const double det = cross2(pa, p0, p1);
                            if (det != 0.0) {
                                const double sg = det > 0.0 ? 1.0 : -1.0, tol = kEps * fabs(det);
                                const double wa = sg * cross2(p0, p1, q), wb = sg * cross2(p1, pa, q), wc = sg * cross2(pa, p0, q);
                                if (wa >= -tol && wb >= -tol && wc >= -tol) {
                                    tr.i0 = sidx; tr.i1 = (unsigned)n0; tr.i2 = (unsigned)n1; tr.ok = true;
                                    canonical3(tr.i0, tr.i1, tr.i2);
                                    (void)bary(pt(pos, tr.i0), pt(pos, tr.i1), pt(pos, tr.i2), qx, qy, c0, c1, c2);
                                    return true;
                                }
                            }
                        }
                        n0 = n1; p0 = p1;
                    }
                }
            }
        }
    return false;
}

// The triangle of the triangulation that contains (qx, qy): a visibility walk from the owner triangle of the grid node
// next to the position.  Stepping over the edge u -> v of the counter-clockwise triangle (u, v, w) leads to the triangle
// (u, x, v) with x the neighbour BEFORE v in u's star; a gap there means the position is outside the convex hull.
__device__ bool dl_locate(const DlWs &ws, unsigned far_base, const PosFn &pos, int H, int W, double qx, double qy,
                          TriRef &tr, double &c0, double &c1, double &c2)
{
    if (!(qx == qx && qy == qy)) return false;
    const int nx = (int)fmin(fmax(rint(qx), 0.0), (double)(W - 1)), ny = (int)fmin(fmax(rint(qy), 0.0), (double)(H - 1));
    unsigned id = kNoOwner;
    for (int r = 0; r <= 2 && id == kNoOwner; ++r)
        for (int dy = -r; dy <= r && id == kNoOwner; ++dy)
            for (int dx = -r; dx <= r && id == kNoOwner; ++dx) {
                if (max(abs(dx), abs(dy)) != r) continue;
                const int x = nx + dx, y = ny + dy;
                if (x < 0 || y < 0 || x >= W || y >= H) continue;
                id = ws.owner[(size_t)y * W + x];
            }
    if (id == kNoOwner) {
        // No owned node next to the position (it lies outside the image, or the hull does not cover the image there): seed
        // the walk from the point set instead -- the first site in growing bucket rings around the position that has a star
        // with two consecutive real neighbours.  A position inside the hull is reached from ANY triangle.
        const Grid g = ws.head->grid;
        const int bx = g.bx(qx), by = g.by(qy);
        const int rmax = max(max(bx, g.gx - 1 - bx), max(by, g.gy - 1 - by));
        bool seeded = false;
        for (int r = 0; r <= rmax && r <= 512 && !seeded; ++r)
            for (int row = by - r; row <= by + r && !seeded; ++row) {
                if (row < 0 || row >= g.gy) continue;
                const bool full = row == by - r || row == by + r;
                for (int part = 0; part < (full ? 1 : 2) && !seeded; ++part) {
                    int x0 = full ? bx - r : (part ? bx + r : bx - r), x1 = full ? bx + r : x0;
                    if (!full && (x0 < 0 || x0 >= g.gx)) continue;
                    x0 = max(x0, 0); x1 = min(x1, g.gx - 1);
                    if (x1 < x0) continue;
                    for (unsigned j = ws.bstart[(size_t)row * g.gx + x0]; j < ws.bstart[(size_t)row * g.gx + x1 + 1] && !seeded; ++j) {
                        const unsigned sidx = ws.sorted[j];
                        if (sidx == 0xFFFFFFFFu) continue;
                        const StarRef ss = star_of(ws, sidx);
                        for (unsigned k = 0; k < ss.d && !seeded; ++k) {
                            const int n0 = star_at(ss, k), n1 = star_at(ss, k + 1 == ss.d ? 0 : k + 1);
                            if (n0 < 0 || n1 < 0 || n0 == n1) continue;
                            tr.i0 = sidx; tr.i1 = (unsigned)n0; tr.i2 = (unsigned)n1; tr.ok = true;
                            seeded = true;
                        }
                    }
                }
            }
        if (!seeded) return false;
    } else {
        tr = dl_decode(id, far_base, ws);
        if (!tr.ok) return dl_locate_brute(ws, pos, qx, qy, kLocateRings, tr, c0, c1, c2);
    }
    unsigned a = tr.i0, b = tr.i1, c = tr.i2;
    D2 pa = pt(pos, a), pb = pt(pos, b), pc = pt(pos, c);
    if (cross2(pa, pb, pc) < 0.0) { const unsigned t = b; b = c; c = t; const D2 tp = pb; pb = pc; pc = tp; }     // counter-clockwise
    for (int step = 0; step < 16384; ++step) {
        // edge functions of q (positive inside)
        const double det = cross2(pa, pb, pc);
        if (!(det > 0.0)) return dl_locate_brute(ws, pos, qx, qy, kLocateRings, tr, c0, c1, c2);
        const double wa = cross2(pb, pc, D2{ qx, qy }), wb = cross2(pc, pa, D2{ qx, qy }), wc = cross2(pa, pb, D2{ qx, qy });
        const double tol = kEps * det;
        if (wa >= -tol && wb >= -tol && wc >= -tol) {
            tr.i0 = a; tr.i1 = b; tr.i2 = c;
            canonical3(tr.i0, tr.i1, tr.i2);
            tr.ok = true;
            (void)bary(pt(pos, tr.i0), pt(pos, tr.i1), pt(pos, tr.i2), qx, qy, c0, c1, c2);
            return true;
        }
        // leave through the most violated edge: wa belongs to edge b -> c, wb to c -> a, wc to a -> b
        unsigned u, v;
        D2 pu, pv;
        if (wa <= wb && wa <= wc) { u = b; v = c; pu = pb; pv = pc; }
        else if (wb <= wc)        { u = c; v = a; pu = pc; pv = pa; }
        else                      { u = a; v = b; pu = pa; pv = pb; }
        // the triangle on the other side of u -> v: the neighbour BEFORE v in u's star -- or, where u's star does not list v
        // (stars that disagree on a co-circular cell), the neighbour AFTER u in v's star, which is the same site when both do
        int x = -2;                                         // -2: neither star lists the edge, -1: an unbounded gap
        {
            const StarRef su = star_of(ws, u);
            unsigned k = 0;
            while (k < su.d && star_at(su, k) != (int)v) ++k;
            if (k < su.d) x = star_at(su, k == 0 ? su.d - 1 : k - 1);
        }
        if (x == -2) {
            const StarRef sv = star_of(ws, v);
            unsigned k = 0;
            while (k < sv.d && star_at(sv, k) != (int)u) ++k;
            if (k < sv.d) x = star_at(sv, k + 1 == sv.d ? 0 : k + 1);
        }
        if (x == -1) return false;                                     // an unbounded gap: outside the convex hull
        D2 px = x >= 0 ? pt(pos, (unsigned)x) : D2{ 0.0, 0.0 };
        if (x < 0 || !(cross2(pu, px, pv) > 0.0)) {
            // Neither star lists the edge, or the order of a star is no guide here (hull stars of lattice fields repeat a
            // neighbour around a collinear run: a, b, a): choose the apex GEOMETRICALLY -- among the neighbours of u and of v
            // that lie on the far side of u -> v, the one whose circle through u and v holds none of the others.
            int best = -1;
            D2 pbest{ 0.0, 0.0 };
            for (int side = 0; side < 2; ++side) {
                const StarRef sn = star_of(ws, side ? v : u);
                for (unsigned k = 0; k < sn.d; ++k) {
                    const int n = star_at(sn, k);
                    if (n < 0 || n == (int)u || n == (int)v || n == best) continue;
                    const D2 pn = pt(pos, (unsigned)n);
                    if (!(cross2(pu, pn, pv) > 0.0)) continue;
                    if (best < 0 || incircle(pu, pbest, pv, pn) > 0.0) { best = n; pbest = pn; }
                }
            }
            if (best < 0) return dl_locate_brute(ws, pos, qx, qy, kLocateRings, tr, c0, c1, c2);
            x = best; px = pbest;
        }
        // new triangle (u, x, v), counter-clockwise
        a = u; pa = pu; b = (unsigned)x; pb = px; c = v; pc = pv;
    }
    return dl_locate_brute(ws, pos, qx, qy, kLocateRings, tr, c0, c1, c2);
}

template <bool SPARSE>
__global__ __launch_bounds__(256)
void dl_query_kernel(const float *__restrict__ flow, int sign, const float *__restrict__ vals, int C,
                     const uint8_t *__restrict__ vmask, int H, int W, const void *__restrict__ query, size_t n,
                     void *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule, DlWs ws, unsigned far_base)
{
    const PosFn pos(flow, sign, W);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double qx, qy;
        if (SPARSE) { qx = ((const double *)query)[2 * i]; qy = ((const double *)query)[2 * i + 1]; }
        else { const float2 q = ((const float2 *)query)[i]; qx = (double)q.x; qy = (double)q.y; }
        TriRef tr;
        double c0 = 0, c1 = 0, c2 = 0;
        const bool found = dl_locate(ws, far_base, pos, H, W, qx, qy, tr, c0, c1, c2);
        if (SPARSE) {
            double *o = (double *)out + i * C;
            for (int c = 0; c < C; ++c)
                o[c] = found ? c0 * (double)vals[(size_t)tr.i0 * C + c] + c1 * (double)vals[(size_t)tr.i1 * C + c] +
                               c2 * (double)vals[(size_t)tr.i2 * C + c] : 0.0;
            valid[i] = found ? 1 : 0;
        } else if (found) {
            const size_t vi[3] = { tr.i0, tr.i1, tr.i2 };
            resolve_emit(vals, C, vmask, vi, c0, c1, c2, valid_rule, (float *)out, valid, i);
        } else {
            for (int c = 0; c < C; ++c) ((float *)out)[i * C + c] = 0.0f;
            if (valid) valid[i] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ slab mode: the exchange
// BASELINE config 5 ('s' at 4320 x 7680) over several GPUs: rank r builds the cell / fan / clip stars of ITS slab only (the
// bulk of the time), the ranks exchange the sites those passes left unfinished -- tens of thousands out of tens of millions --
// and every rank finishes ALL of them, since the later passes search nothing but the unfinished sites beyond the fine rings
// (a Delaunay neighbour of an unfinished site out there is itself unfinished) and a triangle of such a site can cross any
// band.  List layout: word 0 = entries, word 1 = the rank's error bits, words 2 - 3 = 0, then one neighbour row (kSlots
// words: seeds as the clip pass left them) per unfinished site with the site's index in slot 0 (which holds nothing yet).
constexpr unsigned kSlabHead = 4;

__global__ __launch_bounds__(256)
void dl_slab_emit_kernel(const float *__restrict__ flow, int sign, int W, size_t n, const DlHead *__restrict__ head,
                         const unsigned char *__restrict__ deg, const unsigned *__restrict__ nbr, double own_lo, double own_hi,
                         unsigned *__restrict__ list, unsigned cap)
{
    const PosFn pos(flow, sign, W);
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < n; p += (size_t)gridDim.x * 256) {
        if (deg[p] != kDegFar) continue;
        const P2 q = pos((int)p);
        if (!(q.y >= own_lo && q.y < own_hi)) continue;             // another rank's to report (the bands tile the real line)
        const unsigned slot = atomicAdd(&list[0], 1u);
        if (slot >= cap) continue;                                  // (the reader sees entries > capacity)
        const uint4 *src = reinterpret_cast<const uint4 *>(nbr + p * kSlots);
        uint4 *dst = reinterpret_cast<uint4 *>(list + kSlabHead + (size_t)slot * kSlots);
        uint4 r0 = src[0];
        r0.x = (unsigned)p;
        dst[0] = r0; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && head->err) atomicOr(&list[1], head->err);
}

__global__ __launch_bounds__(256)
void dl_slab_absorb_kernel(DlHead *__restrict__ head, const unsigned *__restrict__ lists, size_t stride_words, int n_lists, unsigned cap,
                           size_t n, unsigned char *__restrict__ deg, unsigned *__restrict__ nbr)
{
    for (int r = 0; r < n_lists; ++r) {
        const unsigned *list = lists + (size_t)r * stride_words;
        const unsigned total = list[0], cnt = total < cap ? total : cap;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const unsigned e = list[1] | (total > cap ? kErrSlabList : 0u);
            if (e) atomicOr(&head->err, e);
        }
        for (unsigned k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
            const uint4 *src = reinterpret_cast<const uint4 *>(list + kSlabHead + (size_t)k * kSlots);
            const uint4 r0 = src[0];
            const size_t p = r0.x;
            // not a list of this field?  (the later passes take positions from the seeds' site numbers: none may leave the field)
            const uint4 r1 = src[1], r2 = src[2], r3 = src[3];
            const unsigned sd[12] = { r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, r3.x, r3.y };
            const unsigned ns = r0.y < (unsigned)(kSlots - 4) ? r0.y : (unsigned)(kSlots - 4);
            bool bad = p >= n;
#pragma unroll
            for (unsigned k2 = 0; k2 < (unsigned)(kSlots - 4); ++k2) bad = bad || (k2 < ns && sd[k2] >= n && sd[k2] < 0xFFFFFFFCu);      // (-1 .. -4: box sides)
            if (bad) { atomicOr(&head->err, kErrSlabList); continue; }
            deg[p] = kDegFar;
            uint4 *dst = reinterpret_cast<uint4 *>(nbr + p * kSlots);
            dst[0] = r0; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
        }
    }
}

struct Sizes { size_t n, bcap, b1cap, pool_cap, big_cap; };

Sizes sizes_for(int H, int W)
{
    Sizes z;
    z.n = (size_t)H * W;
    z.bcap = z.n + z.n / 4 + 4096;  // ~ one site per bucket needs n + O(sqrt n) buckets; dl_params_kernel widens the buckets of a point set
                                    // whose shape would need more (2 n + 1 in the worst case: all sites on one slanted line)
    z.b1cap = z.bcap / 6 + 64;      // ceil(gx / 8) * ceil(gy / 8) <= gx gy / 64 + (gx + gy) / 8 + 1 with gx gy <= bcap, gx + gy <= bcap + 1
    z.pool_cap = 8 * z.n + 65536;
    z.big_cap = 6 * z.n + 1024;
    return z;
}

DlWs carve_exact(void *base, int H, int W, size_t *total = nullptr)
{
    const Sizes z = sizes_for(H, W);
    const size_t n = z.n;
    DlWs ws;
    char *p = (char *)base;
    ws.bcap = z.bcap; ws.b1cap = z.b1cap; ws.pool_cap = z.pool_cap; ws.big_cap = z.big_cap;
    ws.head = (DlHead *)p;                  p += 256;
    ws.bstart = (unsigned *)p;              p += align_up((ws.bcap + 1) * 4, 256);
    ws.scan_tmp = (unsigned *)p;            p += align_up(scan_tmp_elems(ws.bcap + 1) * 4, 256);
    ws.sorted = (unsigned *)p;              p += align_up(n * 4, 256);
    ws.deg = (unsigned char *)p;            p += align_up(n, 256);
    ws.todo_idx = (unsigned *)p;            p += align_up(n * 4, 256);
    ws.cellflag = (unsigned char *)p;       p += align_up(n, 256);
    ws.tileflag = (unsigned char *)p;       p += align_up((size_t)((W + 31) / 32) * ((H + 7) / 8), 256);
    ws.dup = (unsigned char *)p;            p += align_up(n, 256);
    ws.nbr = (unsigned *)p;                 p += align_up(n * kSlots * 4, 256);
    ws.far_idx = (unsigned *)p;             p += align_up(n * 4, 256);
    ws.far_deg = (unsigned *)p;             p += align_up(n * 4, 256);
    ws.far_off = (unsigned *)p;             p += align_up(n * 4, 256);
    ws.far_wide = (unsigned char *)p;       p += align_up(n, 256);
    ws.left_idx = (unsigned *)p;            p += align_up(n * 4, 256);
    ws.b1start = (unsigned *)p;             p += align_up((ws.b1cap + 1) * 4, 256);
    ws.b1cursor = (unsigned *)p;            p += align_up(ws.b1cap * 4, 256);
    ws.sorted1 = (unsigned *)p;             p += align_up(n * 4, 256);
    ws.sorted_xy = (P2 *)p;                 p += align_up(n * 16, 256);
    ws.sorted1_xy = (P2 *)p;                p += align_up(n * 16, 256);
    ws.sorted1_pt = (unsigned *)p;          p += align_up(n * 4, 256);
    ws.left_xy = (P2 *)p;                   p += align_up(n * 16, 256);
    ws.left_pt = (unsigned *)p;             p += align_up(n * 4, 256);
    ws.left_box = (double *)p;              p += align_up((n / 256 + 1) * 128, 256);
    ws.pool = (int *)p;                     p += align_up(ws.pool_cap * 4, 256);
    ws.cstride = align_up(std::max(n, ws.bcap) / kScanChunk + 5, 64);
    ws.cstate = (unsigned *)p;              p += align_up(6 * ws.cstride * 4, 256);
    ws.heavy_cap = n / kHeavy + 64;         // (a heavy bucket holds more than kHeavy of the n entries)
    ws.sub_cap = n + ws.heavy_cap + 64;     // (a grid of K x K cells for m entries: K = ceil(sqrt(m / 2)), K * K + 1 <= m for m > kHeavy)
    ws.heavy_bucket = (unsigned *)p;        p += align_up(ws.heavy_cap * 4, 256);
    ws.heavy_info = (SubGrid *)p;           p += align_up(ws.heavy_cap * sizeof(SubGrid), 256);
    ws.sub_start = (unsigned *)p;           p += align_up(ws.sub_cap * 4, 256);
    ws.big = (unsigned *)p;                 p += align_up(ws.big_cap * 4, 256);
    ws.owner = (uint32_t *)p;               p += align_up(n * 4, 256);
    ws.oy0 = 0; ws.oy1 = H;
    if (total) *total = (size_t)(p - (char *)base);
    return ws;
}

int scan_exclusive(unsigned *data, size_t n, unsigned *tmp, hipStream_t s)
{
    if (n <= (size_t)kScanBig) {
        hipLaunchKernelGGL(dl_scan_small_kernel, dim3(1), dim3(1024), 0, s, data, (unsigned)n);
    } else {
        const size_t m = (n + kScanBig - 1) / kScanBig;                // chunk sums; tmp holds them (and the next level's, recursively)
        hipLaunchKernelGGL(dl_scan_reduce_kernel, dim3((unsigned)m), dim3(256), 0, s, (const unsigned *)data, n, tmp);
        OFL_TRY(scan_exclusive(tmp, m, tmp + (m + 63) / 64 * 64, s));
        hipLaunchKernelGGL(dl_scan_apply_kernel, dim3((unsigned)m), dim3(256), 0, s, data, n, (const unsigned *)tmp);
    }
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

}  // namespace

namespace ofl_sc {

size_t exact_workspace_bytes(int H, int W)
{
    size_t total = 0;
    (void)carve_exact(nullptr, H, W, &total);
    return total;
}

}  // namespace ofl_sc

namespace {

DlWs carve_band(void *workspace, int H, int W, int row0, int rows)
{
    DlWs ws = carve_exact(workspace, H, W);
    ws.oy0 = row0; ws.oy1 = row0 + rows;
    ws.owner -= (size_t)row0 * W;
    return ws;
}

// Bins and the stars that cells, fans and the clip pass settle; what they leave is marked kDegFar with its seeds.
// `slab`: only for the sites within kSlabMargin buckets of rows [row0, row0 + rows) -- every triangle that reaches those rows
// has ALL its vertices there or is a triangle of an unfinished site: a finished star's neighbours lie within kRings + 1
// buckets of its site per axis (clip pass: a cell is final only when twice its reach is covered by the rings searched; cells
// and fans: circumcircles within kFanSpan buckets), so a triangle with one finished vertex spans at most 2 (kRings + 1) bucket
// rows.  The copy of a triangle that is drawn is the one of its smallest-index vertex -- whose star is therefore built here
// whenever the triangle matters, with the same neighbour order as in a whole-field run (the passes read the bins of ALL
// sites either way): owner ids, and with them every tie on a shared edge, come out the same.
int exact_stars(const float *flow, int sign_pp, const uint8_t *pmask, int H, int W, int row0, int rows, bool slab,
                void *workspace, size_t workspace_bytes, hipStream_t s, DlWs &ws)
{
    const size_t n = (size_t)H * W;
    if (n >= (1ull << 27)) return fail(OFL_E_INVALID, "ofl_scatter_linear: the exact path takes fields below 2^27 pixels");
    if (workspace_bytes < ofl_sc::exact_workspace_bytes(H, W)) return fail(OFL_E_INVALID, "ofl_scatter_linear: workspace too small for the exact path");
    ws = carve_band(workspace, H, W, row0, rows);
    OFL_HIP(hipMemsetAsync(ws.owner + (size_t)row0 * W, 0xFF, (size_t)rows * W * 4, s));
    DlHead init;
    memset(&init, 0, sizeof(init));
    init.kx0 = init.ky0 = ~0ull;
    OFL_HIP(hipMemcpyAsync(ws.head, &init, sizeof(init), hipMemcpyHostToDevice, s));
    OFL_HIP(hipMemsetAsync(ws.bstart, 0, (ws.bcap + 1) * 4, s));
    const unsigned nblk = (unsigned)((n + 255) / 256);
    const dim3 bgrid((W + 31) / 32, std::max(1, std::min((H + 7) / 8, 4096 / ((W + 31) / 32) + 1)));
    unsigned long long *partial = (unsigned long long *)ws.pool;        // (the neighbour pool is free until the cooperative passes; 8 n + 65 536 words)
    hipLaunchKernelGGL(dl_bbox_kernel, bgrid, dim3(256), 0, s, flow, sign_pp, pmask, H, W, partial);
    static const double bucket_scale = OFL_KNOB_DOUBLE("OFL_DL_BUCKET", 1.0);      // development knob (experiments build only)
    hipLaunchKernelGGL(dl_params_kernel, dim3(1), dim3(256), 0, s, ws.head, (unsigned long long)ws.bcap, bucket_scale, H, W, slab ? 1 : 0, row0, rows,
                       (const unsigned long long *)partial, bgrid.x * bgrid.y);
    OFL_HIP(hipMemsetAsync(ws.dup, 0, n, s));
    hipLaunchKernelGGL(dl_count_kernel, dim3(nblk), dim3(256), 0, s, flow, sign_pp, pmask, H, W, (const DlHead *)ws.head, ws.bstart, ws.dup,
                       ws.todo_idx);                                     // (the fan pass's list is free until the cells are done)
    OFL_HIP(hipGetLastError());
    OFL_TRY(scan_exclusive(ws.bstart, ws.bcap + 1, ws.scan_tmp, s));
    hipLaunchKernelGGL(dl_fill_kernel, dim3(nblk), dim3(256), 0, s, flow, sign_pp, pmask, H, W, (const DlHead *)ws.head,
                       (const unsigned *)ws.bstart, (const unsigned *)ws.todo_idx, ws.sorted, (const unsigned char *)ws.dup);
    hipLaunchKernelGGL(dl_sort_kernel, dim3(std::min<unsigned>((unsigned)((ws.bcap + 255) / 256), 65535u)), dim3(256), 0, s,
                       (const DlHead *)ws.head, 0, (const unsigned *)ws.bstart, ws.sorted);
    hipLaunchKernelGGL(dl_list_xy_kernel<0>, dim3(std::min<unsigned>(nblk, 65535u)), dim3(256), 0, s, flow, sign_pp, W, (const DlHead *)ws.head,
                       (const unsigned *)ws.sorted, (const unsigned *)nullptr, ws.sorted_xy, (unsigned *)nullptr);
    hipLaunchKernelGGL(dl_dedupe_kernel, dim3(std::min<unsigned>((unsigned)((ws.bcap + 255) / 256), 65535u)), dim3(256), 0, s,
                       ws.head, (const unsigned *)ws.bstart, ws.sorted, (const P2 *)ws.sorted_xy, ws.dup);
    {   // buckets too large for the pairwise dedupe (whole image blocks collapsed onto one spot): no-ops when there are none
        const unsigned gb = std::min<unsigned>(nblk, 16384u);
        hipLaunchKernelGGL(dl_big_clear_kernel, dim3(gb), dim3(256), 0, s, (const DlHead *)ws.head, (unsigned *)ws.pool, (unsigned long long)ws.pool_cap);
        hipLaunchKernelGGL(dl_big_kernel<0>, dim3(gb), dim3(256), 0, s, ws.head, (const unsigned *)ws.bstart, ws.sorted, (const P2 *)ws.sorted_xy,
                           ws.dup, (unsigned *)ws.pool, (unsigned long long)ws.pool_cap, n, flow, sign_pp, W);
        hipLaunchKernelGGL(dl_big_kernel<1>, dim3(gb), dim3(256), 0, s, ws.head, (const unsigned *)ws.bstart, ws.sorted, (const P2 *)ws.sorted_xy,
                           ws.dup, (unsigned *)ws.pool, (unsigned long long)ws.pool_cap, n, flow, sign_pp, W);
    }
    const unsigned fblk = (unsigned)((n + kScanChunk - 1) / kScanChunk);
    OFL_HIP(hipMemsetAsync(ws.cstate + 2 * ws.cstride, 0, 2 * ws.cstride * 4, s));      // tickets and tile words of the two compactions below
    {   // dense clusters: the buckets of more than kHeavy entries get grids of their own (two launches that return at once without such a bucket)
        OFL_HIP(hipMemsetAsync(ws.cstate + 5 * ws.cstride, 0, ws.cstride * 4, s));
        const unsigned bblk = (unsigned)((ws.bcap + kScanChunk - 1) / kScanChunk);
        hipLaunchKernelGGL(dl_compact_kernel<5>, dim3((bblk + kCompactTiles - 1) / kCompactTiles), dim3(256), 0, s, (const void *)ws.bstart, (const unsigned char *)nullptr,
                           ws.head, ws.heavy_cap, ws.cstate + 5 * ws.cstride, ws.heavy_bucket, (unsigned *)nullptr);
        hipLaunchKernelGGL(dl_sub_bin_kernel, dim3(1024), dim3(256), 0, s, ws.head, (const unsigned *)ws.bstart, ws.sorted, ws.sorted_xy,
                           (const unsigned *)ws.heavy_bucket, ws.heavy_info, ws.sub_start, (unsigned long long)ws.sub_cap, ws.far_off, ws.left_xy, ws.far_deg);
    }
    // mesh cells (one verification per triangle), the sites they settle, then the fans of the rest (compacted in index order)
    {
        const unsigned sblk = (unsigned)((n / fan_sample_stride(n) + 1 + 255) / 256);
        hipLaunchKernelGGL(dl_cell_sample_kernel, dim3(sblk), dim3(256), 0, s, flow, sign_pp, pmask, H, W, ws.head, (const unsigned *)ws.bstart,
                           (const unsigned *)ws.sorted, (const P2 *)ws.sorted_xy, (const unsigned char *)ws.dup);
        const int ctx = (W + 31) / 32, cty = (H + 7) / 8;
        hipLaunchKernelGGL(dl_cell_kernel, dim3((unsigned)((ctx * cty + 7) / 8 * 8)), dim3(256), 0, s, flow, sign_pp, pmask, H, W,
                           (const DlHead *)ws.head, (const unsigned *)ws.bstart, (const unsigned *)ws.sorted, (const P2 *)ws.sorted_xy,
                           (const unsigned char *)ws.dup, ws.cellflag, ws.tileflag, ctx, ctx * cty);
        hipLaunchKernelGGL(dl_site_cells_kernel, dim3(nblk), dim3(256), 0, s, flow, sign_pp, pmask, H, W, (const DlHead *)ws.head,
                           (const unsigned char *)ws.dup, (const unsigned char *)ws.cellflag, ws.deg, ws.nbr);
    }
    hipLaunchKernelGGL(dl_compact_kernel<2>, dim3((fblk + kCompactTiles - 1) / kCompactTiles), dim3(256), 0, s, (const void *)ws.deg, (const unsigned char *)nullptr, ws.head, n,
                       ws.cstate + 2 * ws.cstride, ws.todo_idx, (unsigned *)nullptr);
    hipLaunchKernelGGL(dl_star_fan_kernel, dim3(std::min<unsigned>((unsigned)((n + kFanBlock - 1) / kFanBlock), 16384u)), dim3(kFanBlock), 0, s, flow, sign_pp, pmask, H, W,
                       (const DlHead *)ws.head, (const unsigned *)ws.todo_idx, (const unsigned *)ws.bstart, (const unsigned *)ws.sorted, (const P2 *)ws.sorted_xy,
                       (const unsigned char *)ws.dup, ws.deg, ws.nbr);
    // what cells and fans did not settle, in index order, for the clip pass
    hipLaunchKernelGGL(dl_compact_kernel<3>, dim3((fblk + kCompactTiles - 1) / kCompactTiles), dim3(256), 0, s, (const void *)ws.deg, (const unsigned char *)nullptr, ws.head, n,
                       ws.cstate + 3 * ws.cstride, ws.far_idx, (unsigned *)nullptr);           // (far_idx is free until the unfinished points are listed)
    hipLaunchKernelGGL(dl_star_near_kernel<false>, dim3(std::min<unsigned>((unsigned)((n + 63) / 64), 16384u)), dim3(64), 0, s, flow, sign_pp, pmask, (const unsigned char *)ws.dup, H, W,
                       (const DlHead *)ws.head, (const unsigned *)ws.far_idx, (const unsigned *)ws.bstart, (const unsigned *)ws.sorted,
                       (const P2 *)ws.sorted_xy, ws.deg, ws.nbr, (const unsigned *)ws.heavy_bucket, (const SubGrid *)ws.heavy_info, (const unsigned *)ws.sub_start);
    hipLaunchKernelGGL(dl_star_near_kernel<true>, dim3(std::min<unsigned>((unsigned)((n + 63) / 64), 2048u)), dim3(64), 0, s, flow, sign_pp, pmask, (const unsigned char *)ws.dup, H, W,
                       (const DlHead *)ws.head, (const unsigned *)ws.far_idx, (const unsigned *)ws.bstart, (const unsigned *)ws.sorted,
                       (const P2 *)ws.sorted_xy, ws.deg, ws.nbr, (const unsigned *)ws.heavy_bucket, (const SubGrid *)ws.heavy_info, (const unsigned *)ws.sub_start);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

// The unfinished sites in index order, their stars (second per-thread pass, wave pass, workgroup pass) and the owner map
// of rows [row0, row0 + rows).  Asynchronous unless the caller asks for the counts.
int exact_finish(const float *flow, int sign_pp, int H, int W, int row0, int rows, uint64_t *info_host, hipStream_t s, DlWs &ws,
                 unsigned &far_base_out,
                 int &late_error)      // OFL_E_NOPOINTS / capacity errors known only after the fact: the owner map is complete (empty) all the same
{
    late_error = OFL_OK;
    const size_t n = (size_t)H * W;
    const unsigned nblk = (unsigned)((n + 255) / 256);
    const unsigned fblk = (unsigned)((n + kScanChunk - 1) / kScanChunk);
    static const bool debug = OFL_KNOB_SET("OFL_DL_DEBUG");                      // development aid (experiments build only)
    // unfinished points in index order
    OFL_HIP(hipMemsetAsync(ws.cstate, 0, 2 * ws.cstride * 4, s));
    OFL_HIP(hipMemsetAsync(ws.cstate + 4 * ws.cstride, 0, ws.cstride * 4, s));
    hipLaunchKernelGGL(dl_compact_kernel<0>, dim3((fblk + kCompactTiles - 1) / kCompactTiles), dim3(256), 0, s, (const void *)ws.deg, (const unsigned char *)nullptr, ws.head, n,
                       ws.cstate, ws.far_idx, ws.nbr);
    OFL_HIP(hipGetLastError());
    // From here on every launch is sized on the device: fixed grids walk the unfinished points, whose numbers stay in the
    // header.  Nothing is read back unless the caller asks for the counts (info_host) -- then ONE read-back at the end,
    // which also carries the errors that cannot be known earlier (no point kept, a capacity exceeded).
    const unsigned long long far_base = (unsigned long long)n * kSlots;
    const unsigned walk = (unsigned)std::min<size_t>(n, 16384);          // workgroups that walk ranks
    {
        // coarse grid of the unfinished points, per-thread / wave pass, then the workgroup pass for what is still left
        const unsigned rblk = std::min<unsigned>(nblk, 2048u);
        OFL_HIP(hipMemsetAsync(ws.b1start, 0, (ws.b1cap + 1) * 4, s));
        OFL_HIP(hipMemsetAsync(ws.b1cursor, 0, ws.b1cap * 4, s));
        hipLaunchKernelGGL(dl_count1_kernel, dim3(rblk), dim3(256), 0, s, flow, sign_pp, W, (const DlHead *)ws.head,
                           (const unsigned *)ws.far_idx, ws.b1start);
        OFL_TRY(scan_exclusive(ws.b1start, ws.b1cap + 1, ws.scan_tmp, s));
        hipLaunchKernelGGL(dl_fill1_kernel, dim3(rblk), dim3(256), 0, s, flow, sign_pp, W, (const DlHead *)ws.head,
                           (const unsigned *)ws.far_idx, (const unsigned *)ws.b1start, ws.b1cursor, ws.sorted1);
        hipLaunchKernelGGL(dl_sort_kernel, dim3(std::min<unsigned>((unsigned)((ws.b1cap + 255) / 256), 65535u)), dim3(256), 0, s,
                           (const DlHead *)ws.head, 1, (const unsigned *)ws.b1start, ws.sorted1);
        hipLaunchKernelGGL(dl_list_xy_kernel<1>, dim3(rblk), dim3(256), 0, s, flow, sign_pp, W, (const DlHead *)ws.head,
                           (const unsigned *)ws.sorted1, (const unsigned *)ws.far_idx, ws.sorted1_xy, ws.sorted1_pt);
        // (experiments build only) OFL_DL_NEAR2_MIN = 0 runs the per-thread pass on the smallest field: tests/test_gpu_scatter_exact.py
        const unsigned near2_min = (unsigned)OFL_KNOB_INT("OFL_DL_NEAR2_MIN", (int)kNear2MinPoints);
        hipLaunchKernelGGL(dl_star_near2_kernel, dim3(std::min<unsigned>((unsigned)((n + 63) / 64), 8192u)), dim3(64), 0, s, flow, sign_pp, H, W,
                           (const DlHead *)ws.head, (const unsigned *)ws.far_idx, (const unsigned *)ws.bstart, (const unsigned *)ws.sorted,
                           (const P2 *)ws.sorted_xy, (const unsigned *)ws.b1start, (const unsigned *)ws.sorted1_pt, (const P2 *)ws.sorted1_xy,
                           ws.deg, ws.nbr, ws.far_deg, ws.far_wide, near2_min, (const unsigned *)ws.heavy_bucket, (const SubGrid *)ws.heavy_info, (const unsigned *)ws.sub_start);
        hipLaunchKernelGGL(dl_star_mid_kernel, dim3(walk), dim3(64), 0, s, flow, sign_pp, H, W, ws.head,
                           (const unsigned *)ws.bstart, (const unsigned *)ws.sorted, (const P2 *)ws.sorted_xy, (const unsigned *)ws.b1start,
                           (const unsigned *)ws.sorted1_pt, (const P2 *)ws.sorted1_xy, (const unsigned *)ws.far_idx, (const unsigned char *)ws.deg, (const unsigned *)ws.nbr, ws.far_deg, ws.far_off, ws.pool,
                           (unsigned long long)ws.pool_cap, ws.far_wide);
        OFL_HIP(hipGetLastError());
        hipLaunchKernelGGL(dl_compact_kernel<1>, dim3((fblk + kCompactTiles - 1) / kCompactTiles), dim3(256), 0, s, (const void *)ws.far_deg, (const unsigned char *)ws.far_wide, ws.head, (size_t)0,
                           ws.cstate + ws.cstride, ws.left_idx, (unsigned *)nullptr);
        hipLaunchKernelGGL(dl_list_xy_kernel<2>, dim3(rblk), dim3(256), 0, s, flow, sign_pp, W,
                           (const DlHead *)ws.head, (const unsigned *)ws.left_idx, (const unsigned *)ws.far_idx, ws.left_xy, ws.left_pt);
        hipLaunchKernelGGL(dl_left_box_kernel, dim3(rblk), dim3(256), 0, s, (const DlHead *)ws.head, W, (const unsigned *)ws.left_pt, (const P2 *)ws.left_xy, ws.left_box);
        hipLaunchKernelGGL((dl_star_far_kernel<OFL_FAR_CAP, 64>), dim3(walk), dim3(64), 0, s, flow, sign_pp, H, W, ws.head,
                           (const unsigned *)ws.bstart, (const unsigned *)ws.sorted, (const P2 *)ws.sorted_xy, (const unsigned *)ws.b1start,
                           (const unsigned *)ws.sorted1_pt, (const P2 *)ws.sorted1_xy, (const unsigned *)ws.far_idx, (const unsigned *)ws.left_idx,
                           (const unsigned *)ws.left_pt, (const P2 *)ws.left_xy, (const double *)ws.left_box, (const unsigned *)ws.nbr,
                           ws.far_deg, ws.far_off, ws.pool, (unsigned long long)ws.pool_cap);
        hipLaunchKernelGGL((dl_star_far_kernel<kFarCap, 256>), dim3(std::min<unsigned>(walk, 2048u)), dim3(256), 0, s, flow, sign_pp, H, W, ws.head,
                           (const unsigned *)ws.bstart, (const unsigned *)ws.sorted, (const P2 *)ws.sorted_xy, (const unsigned *)ws.b1start,
                           (const unsigned *)ws.sorted1_pt, (const P2 *)ws.sorted1_xy, (const unsigned *)ws.far_idx, (const unsigned *)ws.left_idx,
                           (const unsigned *)ws.left_pt, (const P2 *)ws.left_xy, (const double *)ws.left_box, (const unsigned *)ws.nbr,
                           ws.far_deg, ws.far_off, ws.pool, (unsigned long long)ws.pool_cap);
        OFL_HIP(hipGetLastError());
    }
    {
        const int ctx = (W + 31) / 32, cty = (H + 7) / 8;
        hipLaunchKernelGGL(dl_raster_cells_kernel, dim3((unsigned)((ctx * cty + 7) / 8 * 8)), dim3(256), 0, s, flow, sign_pp, H, W, ws, (unsigned)far_base, ctx, ctx * cty);
    }
    hipLaunchKernelGGL(dl_compact_kernel<4>, dim3((fblk + kCompactTiles - 1) / kCompactTiles), dim3(256), 0, s, (const void *)ws.deg, (const unsigned char *)ws.tileflag, ws.head, n,
                       ws.cstate + 4 * ws.cstride, ws.todo_idx, (unsigned *)nullptr, W, H);     // (the fan pass's list is free again)
    hipLaunchKernelGGL(dl_raster_small_kernel, dim3(std::min<unsigned>((nblk + 7) / 8 * 8, 32768u)), dim3(256), 0, s, flow, sign_pp, H, W, ws, (unsigned)far_base,
                       (const unsigned *)ws.todo_idx);
    hipLaunchKernelGGL(dl_raster_far_kernel, dim3(std::min<unsigned>(walk, 4096u)), dim3(256), 0, s, flow, sign_pp, H, W, ws, (unsigned)far_base);
    hipLaunchKernelGGL(dl_raster_big_kernel, dim3((unsigned)rt().n_cu * 4), dim3(256), 0, s, flow, sign_pp, H, W, ws, (unsigned)far_base);
    far_base_out = (unsigned)far_base;
    OFL_HIP(hipGetLastError());
    if (info_host || debug) {
        DlHead h;
        OFL_HIP(hipMemcpyAsync(&h, ws.head, sizeof(h), hipMemcpyDeviceToHost, s));
        OFL_HIP(hipStreamSynchronize(s));
        if (info_host) { info_host[0] = h.kept; info_host[1] = h.n_far; info_host[2] = h.n_left; }
        if (debug) fprintf(stderr, "[ofl exact] kept %u, fan pass %u, clip pass %u, unfinished %u, left over %u\n", h.kept, h.n_fan, h.n_todo, h.n_far, h.n_left);
#ifdef OFL_EXPERIMENTS
        if (debug) {
            unsigned long long c[8], z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
            (void)hipMemcpyFromSymbol(c, HIP_SYMBOL(dl_dbg_counters), sizeof(c));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(dl_dbg_counters), z, sizeof(z));
            fprintf(stderr, "[ofl exact] heavy buckets %u (grid words %u): heavy runs %llu, candidates scanned in them %llu, cells given up %llu, heavy buckets without a grid %llu, grid rows walked %llu; ordinary runs: %llu candidates, %llu clip tests\n",
                    h.n_heavy, h.sub_used, c[0], c[1], c[2], c[3], c[5], c[6], c[7]);
        }
#endif
        if (debug && h.dbg[0]) fprintf(stderr, "[ofl exact] left-over pass: %u sites, per site %.1f chunks swept, %.1f clips before the sweep, %.1f clips in it, %.1f edges\n",
                                       h.dbg[0], (double)h.dbg[1] / h.dbg[0], (double)h.dbg[2] / h.dbg[0], (double)h.dbg[3] / h.dbg[0], (double)h.dbg[4] / h.dbg[0]);
        if (debug && h.dbg[0]) fprintf(stderr, "[ofl exact] left-over pass: most chunks swept by one site %u (of %u), sites with more than 12: %u, most clips of one site %u\n",
                                       h.dbg[5], (h.n_left + 255) / 256, h.dbg[6], h.dbg[7]);
        if (!info_host) return OFL_OK;
        if (h.err) {
            // a capacity or degeneracy error leaves SOME stars rasterised: blank the owner map, so that the result is what
            // include/ofl.h promises for an error -- all zero, all invalid -- rather than a partial warp
            OFL_HIP(hipMemsetAsync(ws.owner + (size_t)row0 * W, 0xFF, (size_t)rows * W * 4, s));
        }
        if (h.kept == 0) late_error = fail(OFL_E_NOPOINTS, "ofl_scatter_linear: no valid source points");
        else if (h.err) late_error = fail(OFL_E_INVALID, "ofl_scatter_linear: exact path capacity exceeded (flags %u: 1 = star of more than %d "
                                              "neighbours, 2 = neighbour pool, 4 = large-triangle list, 8 = unfinished stars beyond the "
                                              "triangle-id space, 16 = degenerate point set: thousands of coincident points or hundreds of "
                                              "thousands of unbounded cells; 32 = slab mode: a rank's list of unfinished sites exceeded its buffer)", h.err, kFarCap);
    }
    return OFL_OK;
}

// Bins, stars and the owner map of rows [row0, row0 + rows) in one go (every star of the field: one GPU, or replicated ranks).
int exact_prepare(const float *flow, int sign_pp, const uint8_t *pmask, int H, int W, int row0, int rows,
                  void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s, DlWs &ws, unsigned &far_base_out,
                  int &late_error)
{
    late_error = OFL_OK;
    OFL_TRY(exact_stars(flow, sign_pp, pmask, H, W, row0, rows, false, workspace, workspace_bytes, s, ws));
    return exact_finish(flow, sign_pp, H, W, row0, rows, info_host, s, ws, far_base_out, late_error);
}


}  // namespace

namespace ofl_sc {

// rows [row0, row0 + rows) of the grid result through the exact path
template <typename VT>
int exact_scatter(const float *flow, int sign_pp, const uint8_t *pmask, const VT *vals, int C, const uint8_t *vmask,
                  int H, int W, int row0, int rows, VT *out, uint8_t *valid, int valid_rule,
                  void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s)
{
    DlWs ws;
    unsigned far_base = 0;
    int late = OFL_OK;
    OFL_TRY(exact_prepare(flow, sign_pp, pmask, H, W, row0, rows, workspace, workspace_bytes, info_host, s, ws, far_base, late));
    // (the outputs are written even when the call reports no points / a refused point set: all zero, all invalid)
    const dim3 grid((W + 31) / 32, (rows + 7) / 8);
    hipLaunchKernelGGL(dl_resolve_kernel<VT>, grid, dim3(256), 0, s, flow, sign_pp, vals, C, vmask, H, W, row0, rows,
                       out, valid, valid_rule, ws, far_base);
    OFL_HIP(hipGetLastError());
    return late;
}

// arbitrary positions through the exact path (see walk_query_launch for the two layouts)
int exact_query(const float *flow, int sign_pp, const uint8_t *pmask, const float *vals, int C, const uint8_t *vmask,
                int H, int W, const void *query, size_t n, bool sparse, void *out, uint8_t *valid, int valid_rule,
                void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s)
{
    DlWs ws;
    unsigned far_base = 0;
    int late = OFL_OK;
    OFL_TRY(exact_prepare(flow, sign_pp, pmask, H, W, 0, H, workspace, workspace_bytes, info_host, s, ws, far_base, late));
    if (n == 0) return late;
    const size_t nb = (n + 255) / 256;
    const dim3 grid((unsigned)(nb < (1u << 20) ? nb : (1u << 20)));
    if (sparse)
        hipLaunchKernelGGL(dl_query_kernel<true>, grid, dim3(256), 0, s, flow, sign_pp, vals, C, vmask, H, W, query, n, out, valid, valid_rule, ws, far_base);
    else
        hipLaunchKernelGGL(dl_query_kernel<false>, grid, dim3(256), 0, s, flow, sign_pp, vals, C, vmask, H, W, query, n, out, valid, valid_rule, ws, far_base);
    OFL_HIP(hipGetLastError());
    return late;
}

// slab mode, step 1: the stars of the slab around rows [row0, row0 + rows) and the list of the unfinished sites this rank
// reports (those whose position lies in [row0, row0 + rows); the first band is open towards -inf, the last towards +inf)
int exact_slab_stars(const float *flow, int sign_pp, const uint8_t *pmask, int H, int W, int row0, int rows,
                     uint32_t *list, size_t list_bytes, void *workspace, size_t workspace_bytes, hipStream_t s)
{
    if (list_bytes < (kSlabHead + kSlots) * 4) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: list buffer too small");
    DlWs ws;
    OFL_TRY(exact_stars(flow, sign_pp, pmask, H, W, row0, rows, true, workspace, workspace_bytes, s, ws));
    const size_t n = (size_t)H * W;
    const size_t cap = std::min<size_t>((list_bytes / 4 - kSlabHead) / kSlots, 0xFFFFFFFFu);
    const double inf = std::numeric_limits<double>::infinity();
    OFL_HIP(hipMemsetAsync(list, 0, kSlabHead * 4, s));
    hipLaunchKernelGGL(dl_slab_emit_kernel, dim3(std::min<unsigned>((unsigned)((n + 255) / 256), 4096u)), dim3(256), 0, s, flow, sign_pp, W, n,
                       (const DlHead *)ws.head, (const unsigned char *)ws.deg, (const unsigned *)ws.nbr,
                       row0 <= 0 ? -inf : (double)row0, row0 + rows >= H ? inf : (double)(row0 + rows), list, (unsigned)cap);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

// slab mode, step 2: the lists of ALL ranks (this rank's included; `n_lists` buffers of `stride_bytes` each, as an all-gather
// leaves them) are merged into the star state step 1 left in `workspace`; then the unfinished stars, the owner map and the
// result of the band, exactly as a whole-field call produces them.
int exact_slab_finish(const float *flow, int sign_pp, const float *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows,
                      const uint32_t *lists, size_t stride_bytes, int n_lists, float *out, uint8_t *valid, int valid_rule,
                      void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s)
{
    if (workspace_bytes < exact_workspace_bytes(H, W)) return fail(OFL_E_INVALID, "ofl_scatter_slab_finish: workspace too small");
    if (n_lists < 1 || stride_bytes % 16 || stride_bytes < (kSlabHead + kSlots) * 4)
        return fail(OFL_E_INVALID, "ofl_scatter_slab_finish: bad list layout");
    DlWs ws = carve_band(workspace, H, W, row0, rows);
    const size_t n = (size_t)H * W;
    const size_t cap = std::min<size_t>((stride_bytes / 4 - kSlabHead) / kSlots, 0xFFFFFFFFu);
    {   // Is this the state step 1 left for THIS band?  Everything below trusts the lists and counts in the workspace: a
        // foreign one must not reach a kernel.  (One 4-byte read-back; the call synchronises at its end anyway.  The stamp is
        // cleared right away: a state is finished once.)
        unsigned stamp = 0;
        OFL_HIP(hipMemcpyAsync(&stamp, &ws.head->slab_stamp, 4, hipMemcpyDeviceToHost, s));
        OFL_HIP(hipStreamSynchronize(s));
        if (stamp != slab_stamp_of(H, W, row0, rows)) {
            if (C > 0) OFL_HIP(hipMemsetAsync(out, 0, (size_t)rows * W * C * sizeof(float), s));
            if (valid) OFL_HIP(hipMemsetAsync(valid, 0, (size_t)rows * W, s));
            return fail(OFL_E_INVALID, "ofl_scatter_slab_finish: the workspace does not hold the state ofl_scatter_slab_stars_dev left for rows "
                                       "[%d, %d) (another call used it in between, or step 1 ran for another band)", row0, row0 + rows);
        }
        OFL_HIP(hipMemsetAsync(&ws.head->slab_stamp, 0, 4, s));
    }
    hipLaunchKernelGGL(dl_slab_absorb_kernel, dim3(256), dim3(256), 0, s, ws.head, lists, stride_bytes / 4, n_lists, (unsigned)cap, n, ws.deg, ws.nbr);
    OFL_HIP(hipGetLastError());
    unsigned far_base = 0;
    int late = OFL_OK;
    uint64_t info_local[3];
    // (always read the counts back: an error bit another rank raised must blank this band too, and the exchange has
    // synchronised the ranks a moment ago anyway)
    OFL_TRY(exact_finish(flow, sign_pp, H, W, row0, rows, info_host ? info_host : info_local, s, ws, far_base, late));
    const dim3 grid((W + 31) / 32, (rows + 7) / 8);
    hipLaunchKernelGGL(dl_resolve_kernel<float>, grid, dim3(256), 0, s, flow, sign_pp, vals, C, vmask, H, W, row0, rows,
                       out, valid, valid_rule, ws, far_base);
    OFL_HIP(hipGetLastError());
    return late;
}

template int exact_scatter<float>(const float *, int, const uint8_t *, const float *, int, const uint8_t *, int, int, int, int,
                                  float *, uint8_t *, int, void *, size_t, uint64_t *, hipStream_t);
template int exact_scatter<double>(const float *, int, const uint8_t *, const double *, int, const uint8_t *, int, int, int, int,
                                   double *, uint8_t *, int, void *, size_t, uint64_t *, hipStream_t);

}  // namespace ofl_sc
