// ofl_scatter_walk.hip -- K3 fast path: the warped grid as a CERTIFIED Delaunay triangulation.
//
// scipy.interpolate.griddata(points, values, grid, 'linear') (src/oflibnumpy/utils.py:253) triangulates the
// warped grid points with Qhull.  When the cell-wise mesh of the warped grid -- every source cell split along its
// Delaunay diagonal -- has (1) only positively oriented triangles, (2) only locally Delaunay interior edges and
// (3) a border that is straight between the four warped image corners, then by the Delaunay lemma that mesh IS the
// Delaunay triangulation of the points (unique up to co-circular cells, where Qhull itself is arbitrary), and the
// convex hull of the points is the hull of the border points.  `scatter_certify_kernel` evaluates exactly these
// conditions in one pass over the flow (counts + border deviations, read back once per field and cached by the
// caller); for a certified field `scatter_walk_kernel` resolves every output node in ONE pass without an owner map,
// atomics, a hull on the host or any synchronisation: the node finds its source cell by Newton steps on the
// piecewise-affine map (the flow at the node gives the first estimate, each visited triangle's affine map the
// next) and interpolates with SciPy's inclusion rule and float64 barycentric coordinates.  Every affine field of the
// reference's tests and of BASELINE configs 1-4 is certified; anything else goes to the exact path (ofl_delaunay.hip).
#include "ofl_scatter_dev.h"
#include <math.h>


using namespace ofl;
using namespace ofl_sc;

namespace {

constexpr double kEdgeTol   = 1e-12;   // relative in-circle excess that counts as "not locally Delaunay" (beyond rounding)
constexpr double kBorderMax = 1e-4;    // px: largest deviation of a border point from its straight side for a certificate
constexpr int    kWalkIters = 16;
constexpr size_t kCertAffineAt = 192;  // byte offset of cert_affine_kernel's seven doubles in the certificate's 256-byte scratch

struct CertDev {                       // device record written by the certify kernel (128 bytes)
    uint32_t folded, bad_edges, dropped, degenerate;
    unsigned long long dev_min[4], dev_max[4];      // ordered keys of the signed border deviations per side (inward > 0)
    D2 corner[4];                                    // P(0,0), P(W-1,0), P(W-1,H-1), P(0,H-1)
};
static_assert(sizeof(CertDev) == 16 + 64 + 64, "CertDev layout");

struct WalkCert {
    D2 c[4]; double inv_len[4]; double delta; const uint32_t *diag; int diag_stride;     // inv_len[k]: 1 / |c[k + 1] - c[k]|
    float pa[4], pb[4], pc[4], lim32;      // side k's inward distance as a float32 plane pa x + pb y + pc, and the (conservative) limit for it
    // the affine map through the warped corners, inverted: source index (ai[0] x + ai[1] y + ai[2], ai[3] x + ai[4] y + ai[5]) of a position,
    // and the word the certificate leaves non-zero when some point of the field lies more than a quarter of a cell off that map
    double ai[6];
    const uint32_t *not_affine;
};

__device__ __forceinline__ unsigned long long okey(double d)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

double okey_inv(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k ^ 0x8000000000000000ull) : ~k;
    double d;
    memcpy(&d, &b, 8);
    return d;
}

__device__ __forceinline__ double shfl_xor_d(double v, int off)
{
    const int lo = __shfl_xor(__double2loint(v), off), hi = __shfl_xor(__double2hiint(v), off);
    return __hiloint2double(hi, lo);
}

// Is d inside the circumcircle of the positively oriented triangle (a, b, c) by more than `tol` times the fourth power of the local
// length scale?  (The in-circle determinant against tol * s^2 -- the quotient it stands for costs a float64 division, thirty-five
// instructions of a kernel that is bound by them.)
__device__ __forceinline__ bool incircle_beyond(const D2 &a, const D2 &b, const D2 &c, const D2 &d, double tol)
{
    const double ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const double ic = ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx);
    const double s = a2 + b2 + c2;
    return s > 0.0 && ic > tol * (s * s);
}

__device__ __forceinline__ void tri_pts(int diag, int t, const D2 &pa, const D2 &pb, const D2 &pc, const D2 &pd,
                                        D2 &q0, D2 &q1, D2 &q2)
{
    int i0, i1, i2;
    tri_corners(diag, t, i0, i1, i2);
    q0 = pick4(i0, pa, pb, pc, pd); q1 = pick4(i1, pa, pb, pc, pd); q2 = pick4(i2, pa, pb, pc, pd);
}

// pick_diagonal for a cell that is convex and positively oriented.  The certificate asks it for a cell's right and lower
// NEIGHBOUR (which vertex faces the shared edge): a neighbour that is not convex is counted as folded by its own thread and voids
// the certificate whatever is decided here.
__device__ __forceinline__ int diag_if_convex(const D2 &a, const D2 &b, const D2 &c, const D2 &d)
{
    return incircle_filtered(a, b, c, d) > 0 ? 1 : 0;
}

// The affine map x -> c0 + (c1 - c0) x / (W - 1) + (c3 - c0) y / (H - 1) through three warped corners, inverted (position -> source
// index); false when it has no inverse.  Evaluated by the certificate (is every point of the field within a quarter of a cell of
// it?) and by the walk's launcher (the first estimate of a node's source cell when it is) -- the same six numbers on both sides.
__host__ __device__ inline bool corner_affine_inverse(const D2 &c0, const D2 &c1, const D2 &c3, int H, int W, double (&ai)[6])
{
    const double ax = (c1.x - c0.x) / (double)(W - 1), ay = (c1.y - c0.y) / (double)(W - 1);      // d position / d x
    const double bx = (c3.x - c0.x) / (double)(H - 1), by = (c3.y - c0.y) / (double)(H - 1);      // d position / d y
    const double det = ax * by - ay * bx;
    if (!(fabs(det) > 1e-12) || !(fabs(det) < 1e12)) return false;
    ai[0] = by / det;  ai[1] = -bx / det; ai[2] = -(ai[0] * c0.x + ai[1] * c0.y);
    ai[3] = -ay / det; ai[4] = ax / det;  ai[5] = -(ai[3] * c0.x + ai[4] * c0.y);
    return true;
}

// the inverse of the corners' affine map for the certificate pass: ai[0 .. 5], then 1.0 when the map has an inverse, else 0.0
__global__ void cert_affine_kernel(const float *__restrict__ flow, int sign, int H, int W, double *__restrict__ affine, uint32_t *__restrict__ not_affine)
{
    if (not_affine) *not_affine = 0u;                   // (the certificate pass sets it)
    double ai[6] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
    bool ok = false;
    if (H >= 2 && W >= 2)
        ok = corner_affine_inverse(point_of(flow, sign, W, 0, 0), point_of(flow, sign, W, W - 1, 0), point_of(flow, sign, W, 0, H - 1), H, W, ai);
    for (int k = 0; k < 6; ++k) affine[k] = ok ? ai[k] : 0.0;
    affine[6] = ok ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------------ certificate
__global__ __launch_bounds__(256)
void scatter_certify_kernel(const float *__restrict__ flow, int sign, const uint8_t *__restrict__ pmask, int H, int W,
                            CertDev *__restrict__ out, uint32_t *__restrict__ diag_bits, int diag_stride,
                            const double *__restrict__ affine,      // cert_affine_kernel's seven numbers
                            int tile_row0, int tile_rows)      // this launch: tile rows [tile_row0, tile_row0 + gridDim.y) of tile_rows
{
    // The pass runs as two launches: a first sixteenth of the rows, then the rest.  Where the first part has already counted tens of
    // thousands of failing cells (a folded lattice like BASELINE config 5: every cell) the second part has nothing to add -- the
    // counts are exact up to ~65 000 and a lower bound beyond, include/ofl.h -- and ends here: 0.9 -> 0.06 ms at 8K.  (A look at the
    // counters of the RUNNING launch would have to be a coherent load per workgroup, which queues at the memory side; across a
    // kernel boundary a plain load sees them.)
    if (tile_row0 > 0 && (out->folded >= (1u << 16) || out->bad_edges >= (1u << 16))) return;
    const int by = tile_row0 + (int)blockIdx.y;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = by * 8 + (threadIdx.x >> 5);
    const bool inpt = x < W && y < H;
    int fold = 0, bad = 0, drop = 0, diag_bit = 0;
    if (inpt && pmask && !pmask[(size_t)y * W + x]) drop = 1;
    if (x < W - 1 && y < H - 1) {
        const D2 pa = point_of(flow, sign, W, x, y), pb = point_of(flow, sign, W, x + 1, y);
        const D2 pc = point_of(flow, sign, W, x + 1, y + 1), pd = point_of(flow, sign, W, x, y + 1);
        const int diag = pick_diagonal(pa, pb, pc, pd);
        diag_bit = diag;
        D2 u0, u1, u2, v0, v1, v2;
        tri_pts(diag, 0, pa, pb, pc, pd, u0, u1, u2);
        tri_pts(diag, 1, pa, pb, pc, pd, v0, v1, v2);
        // convex, positively oriented cells only (a reflex cell is a legal mesh cell, but the walk kernel's diagonal
        // rule assumes convexity; such fields take the exact path)
        if (!(cross2(pa, pb, pc) > 0.0 && cross2(pa, pc, pd) > 0.0 && cross2(pb, pc, pd) > 0.0 && cross2(pb, pd, pa) > 0.0)) {
            fold = 1;
        } else {
            // the cell's own diagonal is Delaunay by construction; its right and bottom edges are shared with the
            // neighbouring cells: the vertex opposite the edge in the neighbour's triangle must not lie inside the
            // circumcircle of this cell's triangle on the edge (local Delaunay test).  Left / top edges belong to the
            // neighbours on that side; edges on the image border have no second triangle.
            if (x + 2 < W) {
                const D2 pb2 = point_of(flow, sign, W, x + 2, y), pc2 = point_of(flow, sign, W, x + 2, y + 1);
                const int dr = diag_if_convex(pb, pb2, pc2, pc);
                const D2 opp = dr == 0 ? pc2 : pb2;                       // (a', c', d') or (b', d', a') holds the edge d'-a'
                // this cell's triangle on the edge b-c: diag 0 -> (a, b, c) = triangle 0; diag 1 -> (b, c, d) = triangle 0
                if (incircle_beyond(u0, u1, u2, opp, kEdgeTol)) bad += 1;
            }
            if (y + 2 < H) {
                const D2 pd2 = point_of(flow, sign, W, x, y + 2), pc3 = point_of(flow, sign, W, x + 1, y + 2);
                const int db = diag_if_convex(pd, pc, pc3, pd2);
                const D2 opp = db == 0 ? pc3 : pd2;                       // (a", b", c") or (b", d", a") holds the edge a"-b"
                // this cell's triangle on the edge c-d: diag 0 -> (a, c, d) = triangle 1; diag 1 -> (b, c, d) = triangle 0
                const bool t1 = diag == 0;
                if (incircle_beyond(t1 ? v0 : u0, t1 ? v1 : u1, t1 ? v2 : u2, opp, kEdgeTol)) bad += 1;
            }
        }
    }
    // The Delaunay diagonal of every cell, one bit per cell (rows padded to whole 32-bit words: a wave's two rows of 32
    // cells are two aligned words, no atomics): the walk kernel reads the bit instead of evaluating the float64 in-circle
    // determinant of the cell for every node that looks at it -- the same predicate on the same numbers, so the same bits out.
    if (diag_bits) {
        const unsigned long long m = __ballot(diag_bit != 0);
        // (row H - 1 holds no cell: its first word is the "not affine" flag below)
        if ((threadIdx.x & 31) == 0 && y < H - 1) diag_bits[(size_t)y * diag_stride + blockIdx.x] = (uint32_t)(m >> (threadIdx.x & 32));
        // Is the field the affine map through its corners, to a quarter of a cell?  Then the walk kernel starts every node from that
        // map's inverse instead of a Newton step on the node's own flow -- one round of loads less per node.  An idempotent plain
        // store by the workgroups that see a deviation: no atomics.
        // (the inverse map comes from cert_affine_kernel: ten float64 divisions that one thread makes once, not every thread of the field)
        int off = 0;
        if (inpt) {
            const double *ai = affine;                      // [6] and a validity word behind them
            const D2 p = point_of(flow, sign, W, x, y);
            // (float32 is plenty against a quarter of a cell at any practical size; where it is not -- a side of millions of nodes -- the field only loses the short cut)
            const float px = (float)p.x, py = (float)p.y;
            const float ex = fmaf((float)ai[0], px, fmaf((float)ai[1], py, (float)ai[2])) - (float)x, ey = fmaf((float)ai[3], px, fmaf((float)ai[4], py, (float)ai[5])) - (float)y;
            off = !(fabsf(ex) <= 0.25f && fabsf(ey) <= 0.25f) || ai[6] == 0.0;
        }
        if (__syncthreads_or(off) && threadIdx.x == 0) diag_bits[(size_t)(H - 1) * diag_stride] = 1u;
    }
    // border: signed distance of every border point from the straight line between the two warped corners of its side
    const bool border_block = blockIdx.x == 0 || by == 0 || blockIdx.x == gridDim.x - 1 || by == tile_rows - 1;
    int degen = 0;
    if (border_block) {
        const D2 c0 = point_of(flow, sign, W, 0, 0), c1 = point_of(flow, sign, W, W - 1, 0);
        const D2 c2 = point_of(flow, sign, W, W - 1, H - 1), c3 = point_of(flow, sign, W, 0, H - 1);
        D2 p = { 0.0, 0.0 };
        if (inpt) p = point_of(flow, sign, W, x, y);
#pragma unroll
        for (int side = 0; side < 4; ++side) {
            const bool on = inpt && (side == 0 ? y == 0 : side == 1 ? x == W - 1 : side == 2 ? y == H - 1 : x == 0);
            const D2 A = pick4(side, c0, c1, c2, c3), B = pick4((side + 1) & 3, c0, c1, c2, c3);
            const double len = sqrt((B.x - A.x) * (B.x - A.x) + (B.y - A.y) * (B.y - A.y));
            double dmin = 1e300, dmax = -1e300;
            if (on) {
                if (!(len > 0.0)) degen = 1;
                else dmin = dmax = cross2(A, B, p) / len;
            }
            if (__any(on)) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    dmin = fmin(dmin, shfl_xor_d(dmin, off));
                    dmax = fmax(dmax, shfl_xor_d(dmax, off));
                }
                if ((threadIdx.x & 63) == 0 && dmin <= dmax) {
                    atomicMin(&out->dev_min[side], okey(dmin));
                    atomicMax(&out->dev_max[side], okey(dmax));
                }
            }
        }
        if (inpt) {
            if (x == 0 && y == 0) out->corner[0] = p;
            if (x == W - 1 && y == 0) out->corner[1] = p;
            if (x == W - 1 && y == H - 1) out->corner[2] = p;
            if (x == 0 && y == H - 1) out->corner[3] = p;
        }
    }
    const int nf = __syncthreads_count(fold), nd = __syncthreads_or(drop), ng = __syncthreads_or(degen);
    // bad edges: up to two per thread
    const int nb = __syncthreads_count(bad >= 1) + __syncthreads_count(bad >= 2);
    if (threadIdx.x == 0) {
        // (a certificate only asks "zero or not"; the counts are reported, but once they are in the tens of thousands more
        // atomics on the same two words only serialise a hundred thousand workgroups: 3 ms at 8K on a folded field)
        if (nf && out->folded < (1u << 16)) atomicAdd(&out->folded, (uint32_t)nf);
        if (nb && out->bad_edges < (1u << 16)) atomicAdd(&out->bad_edges, (uint32_t)nb);
        if (nd) out->dropped = 1u;
        if (ng) out->degenerate = 1u;
    }
}

// ------------------------------------------------------------------------------------------------ walk
struct Hit { uint32_t vi[3]; double c0, c1, c2; };


struct __attribute__((aligned(8))) Pair2 { float lo_u, lo_v, hi_u, hi_v; };    // two adjacent vectors: ONE 16-byte load

template <int SP>          // SP = sign with the point precision folded in: +-1 float64 positions, +-2 rounded to float32
__device__ __forceinline__ D2 warp_pt(int x, int y, float u, float v)
{
    D2 p;
    p.x = SP >= 0 ? (double)x + (double)u : (double)x - (double)u;
    p.y = SP >= 0 ? (double)y + (double)v : (double)y - (double)v;
    if (SP == 2 || SP == -2) { p.x = (double)(float)p.x; p.y = (double)(float)p.y; }
    return p;
}

// 1 / d for a positive, ordinary d (a triangle's doubled area) to a few ulp: hardware estimate + two Newton steps -- five
// instructions where the IEEE division sequence takes about thirty-five (the kernel is bound by float64 instructions;
// the coordinates feed float32 results and a `== 1` that holds for any c0 = 1 - c1 - c2)
__device__ __forceinline__ double rcp_newton(double d)
{
    double x = __builtin_amdgcn_rcp(d);
    double r = fma(-d, x, 1.0);
    x = fma(x, r, x);
    r = fma(-d, x, 1.0);
    return fma(x, r, x);
}

// The test of one source cell whose four warped corners (pa .. pd) and Delaunay diagonal are known.
// The cell's diagonal runs q0 -> q2; triangle 0 = (q0, q1, q2) lies on its right, triangle 1 = (q0, q2, r2) on its left
// (DIAG 0: a, b, c, d; DIAG 1: b, c, d, a).  w1 -- triangle 0's coordinate of q1 times its doubled area, and minus
// triangle 1's coordinate of r2 -- vanishes on the diagonal, so its sign picks the ONE triangle that is evaluated: triangle 1
// when the position lies beyond the diagonal by more than triangle 0's own tolerance.  Both triangles share the products
// below up to sign (a - b == -(b - a) exactly), so each one's numbers are those of its own edge functions.
// DIAG is a template argument because it is the same for every cell of an affine field and for whole regions of a smooth one:
// the caller branches on it (uniform for almost every wave), and the sixteen selects of the corners disappear.
template <int DIAG>
__device__ __forceinline__ bool cell_core(const D2 &pa, const D2 &pb, const D2 &pc, const D2 &pd, int W, int cx, int cy,
                                          double qx, double qy, Hit &h, double &ex, double &ey)
{
    const D2 q0 = DIAG ? pb : pa, q1 = DIAG ? pc : pb, q2 = DIAG ? pd : pc, r2 = DIAG ? pa : pd;
    const double e2x = q2.x - q0.x, e2y = q2.y - q0.y, dx = qx - q0.x, dy = qy - q0.y;
    const double w1 = dx * e2y - dy * e2x;
    const double e1x = q1.x - q0.x, e1y = q1.y - q0.y;
    const double det0 = e1x * e2y - e1y * e2x;                          // > 0 (certificate)
    const bool t = w1 < -(kEps * det0);
    const double gx = t ? r2.x - q0.x : e1x, gy = t ? r2.y - q0.y : e1y;   // the edge q0 -> third vertex
    double det = gx * e2y - gy * e2x, wq = gx * dy - gy * dx, wt = w1;  // x det: coordinates of q2 (wq) and of the third vertex (wt)
    if (t) { det = -det; wq = -wq; wt = -wt; }
    const bool in = fmin(det - wt - wq, fmin(wt, wq)) >= -(kEps * det);
    const double inv = rcp_newton(det);
    const double c1 = wt * inv, c2 = wq * inv, c0 = 1.0 - c1 - c2;
    constexpr int i0 = DIAG, i2 = DIAG + 2;                              // corner numbers a = 0 ... d = 3; corner k sits at
    const int i1 = t ? (DIAG + 3) & 3 : DIAG + 1;                        // (cx + (((k + 1) >> 1) & 1), cy + (k >> 1))
    const int x0 = cx + (((i0 + 1) >> 1) & 1), y0 = cy + (i0 >> 1);
    const int x1 = cx + (((i1 + 1) >> 1) & 1), y1 = cy + (i1 >> 1);
    const int x2 = cx + (((i2 + 1) >> 1) & 1), y2 = cy + (i2 >> 1);
    if (in) {
        h.vi[0] = (uint32_t)y0 * (uint32_t)W + (uint32_t)x0; h.vi[1] = (uint32_t)y1 * (uint32_t)W + (uint32_t)x1;
        h.vi[2] = (uint32_t)y2 * (uint32_t)W + (uint32_t)x2;
        h.c0 = c0; h.c1 = c1; h.c2 = c2;
        return true;
    }
    ex = c0 * x0 + c1 * x1 + c2 * x2;                                  // Newton step: the triangle's affine map applied to the position
    ey = c0 * y0 + c1 * y1 + c2 * y2;
    return false;
}

// Tests the two triangles of source cell (cx, cy) for the position (qx, qy) with SciPy's inclusion rule (division-free
// edge functions; the certificate guarantees convex, positively oriented cells, so the Delaunay diagonal is one
// in-circle sign).  On a miss the affine map of the triangle on the position's side of the diagonal turns the position into
// a new estimate (ex, ey) of its source index (a Newton step on the piecewise-affine map).
template <int SP, bool BITS>
__device__ __forceinline__ bool try_cell(const float *__restrict__ flow, int W, int cx, int cy,
                                         double qx, double qy, Hit &h, double &ex, double &ey, const WalkCert &wc)
{
    const uint32_t i00 = (uint32_t)cy * (uint32_t)W + (uint32_t)cx;          // H * W < 2^29
    const Pair2 r0 = *reinterpret_cast<const Pair2 *>(flow + 2 * (size_t)i00);
    const Pair2 r1 = *reinterpret_cast<const Pair2 *>(flow + 2 * (size_t)(i00 + (uint32_t)W));
    const D2 pa = warp_pt<SP>(cx, cy, r0.lo_u, r0.lo_v), pb = warp_pt<SP>(cx + 1, cy, r0.hi_u, r0.hi_v);
    const D2 pc = warp_pt<SP>(cx + 1, cy + 1, r1.hi_u, r1.hi_v), pd = warp_pt<SP>(cx, cy + 1, r1.lo_u, r1.lo_v);
    // the cell's Delaunay diagonal: from the certificate's bit-plane, or (no plane given) as pick_diagonal decides it for a
    // convex, positive cell
    int diag;
    if (BITS) diag = (int)((wc.diag[(size_t)cy * wc.diag_stride + (cx >> 5)] >> (cx & 31)) & 1u);
    else      diag = incircle_filtered(pa, pb, pc, pd) > 0 ? 1 : 0;
    if (diag) return cell_core<1>(pa, pb, pc, pd, W, cx, cy, qx, qy, h, ex, ey);
    return cell_core<0>(pa, pb, pc, pd, W, cx, cy, qx, qy, h, ex, ey);
}

template <int SP, bool BITS, bool NODE = false>      // the position (qx, qy) starts from the grid node (x, y) next to it (NODE: it is that node)
__device__ __forceinline__ bool walk_locate(const float *__restrict__ flow, int H, int W, int x, int y, double qx, double qy, Hit &h,
                                            const WalkCert &wc)
{
    double ex, ey;
    if (wc.not_affine && *wc.not_affine == 0u) {
        // (uniform) the field is the affine map through its corners to a quarter of a cell: that map's inverse is the estimate, and
        // the node's own flow is not looked at -- one round of loads less
        ex = fma(wc.ai[0], qx, fma(wc.ai[1], qy, wc.ai[2])); ey = fma(wc.ai[3], qx, fma(wc.ai[4], qy, wc.ai[5]));
    } else {
    // first estimate of the source index: one Newton step from the node itself, i = q - J^-1 (P(q) - q), with the
    // Jacobian J = I + s * grad f from the differences to the node's right / lower neighbours (exact for affine fields;
    // float32 is plenty for an estimate that only selects the first cell)
    const int xn = x + 1 < W ? x + 1 : x - 1, yn = y + 1 < H ? y + 1 : y - 1;
    const float2 f0 = *reinterpret_cast<const float2 *>(flow + ((size_t)y * W + x) * 2);
    const float2 fxn = *reinterpret_cast<const float2 *>(flow + ((size_t)y * W + xn) * 2);
    const float2 fyn = *reinterpret_cast<const float2 *>(flow + ((size_t)yn * W + x) * 2);
    const float sg = SP >= 0 ? 1.0f : -1.0f, hx = sg * (float)(xn - x), hy = sg * (float)(yn - y);
    const float ja = 1.0f + (fxn.x - f0.x) * hx, jb = (fyn.x - f0.x) * hy;
    const float jc = (fxn.y - f0.y) * hx, jd = 1.0f + (fyn.y - f0.y) * hy;
    // residual of the node: P(node) - q
    const float jdet = ja * jd - jb * jc;
    const float rx = NODE ? sg * f0.x : sg * f0.x + (float)((double)x - qx), ry = NODE ? sg * f0.y : sg * f0.y + (float)((double)y - qy);
    ex = (double)x - (double)rx; ey = (double)y - (double)ry;
    if (fabsf(jdet) > 1e-3f) {
        const float ij = __builtin_amdgcn_rcpf(jdet);                  // (1 ulp: an estimate)
        ex = (double)x - (double)((jd * rx - jb * ry) * ij);
        ey = (double)y - (double)((ja * ry - jc * rx) * ij);
    }
    }
    int pcx = -1, pcy = -1, ppcx = -2, ppcy = -2;
    for (int it = 0; it < kWalkIters; ++it) {
        // (clamped first, so the truncating conversion is the floor; fmax sends a NaN to 0)
        const int cx = (int)fmin(fmax(ex, 0.0), (double)(W - 2)), cy = (int)fmin(fmax(ey, 0.0), (double)(H - 2));
        if ((cx == pcx && cy == pcy) || (cx == ppcx && cy == ppcy)) break;      // no progress / a 2-cycle across an edge
        ppcx = pcx; ppcy = pcy; pcx = cx; pcy = cy;
        if (try_cell<SP, BITS>(flow, W, cx, cy, qx, qy, h, ex, ey, wc)) return true;
    }
    // the estimate stopped moving without a hit (a node outside the mesh ends here, clamped to a border cell; a node on
    // a cell edge may alternate between its two sides): the cells around the last two stops decide
    double dx, dy;
    for (int k = 0; k < 2; ++k) {
        const int bx = k ? ppcx : pcx, by = k ? ppcy : pcy;
        if (bx < 0) continue;
        for (int oy = -1; oy <= 1; ++oy)
            for (int ox = -1; ox <= 1; ++ox) {
                const int cx = bx + ox, cy = by + oy;
                if (cx < 0 || cy < 0 || cx > W - 2 || cy > H - 2) continue;
                if (try_cell<SP, BITS>(flow, W, cx, cy, qx, qy, h, dx, dy, wc)) return true;
            }
    }
    return false;
}

// Exact hull membership of a node that no mesh triangle covers and that lies within the certificate's band around the
// straight line of border side `side`: with o the outward and s the tangential coordinate of the side's points
// relative to the node, the node is inside the convex hull iff a point on its left and a point on its right span a
// chord that passes on or outside it: max_L o/|s| + max_R o/s >= 0.  SciPy covers such a node with the sliver triangle
// on that chord, i.e. (up to the sliver's height) the interpolation between the chord's end points.
__device__ bool side_scan(const float *__restrict__ flow, int sign, int H, int W, int side, const WalkCert &wc,
                          double qx, double qy, Hit &h)
{
    const D2 A = wc.c[side], B = wc.c[(side + 1) & 3];
    const double len = sqrt((B.x - A.x) * (B.x - A.x) + (B.y - A.y) * (B.y - A.y));
    const double tx = (B.x - A.x) / len, ty = (B.y - A.y) / len;
    const int n = (side & 1) ? H : W;
    double aL = -1e300, aR = -1e300, sL = 0.0, sR = 0.0;
    size_t iL = 0, iR = 0;
    for (int i = 0; i < n; ++i) {
        const int px = side == 0 ? i : side == 1 ? W - 1 : side == 2 ? W - 1 - i : 0;
        const int py = side == 0 ? 0 : side == 1 ? i : side == 2 ? H - 1 : H - 1 - i;
        const D2 p = point_of(flow, sign, W, px, py);
        const double dx = p.x - qx, dy = p.y - qy;
        const double s = dx * tx + dy * ty, o = -(-ty * dx + tx * dy);
        if (s == 0.0) {
            if (o >= 0.0) { iL = iR = (size_t)py * W + px; sL = -1.0; sR = 1.0; aL = aR = 0.0; break; }
            continue;
        }
        const double a = o / fabs(s);
        if (s < 0.0) { if (a > aL) { aL = a; sL = s; iL = (size_t)py * W + px; } }
        else         { if (a > aR) { aR = a; sR = s; iR = (size_t)py * W + px; } }
    }
    if (aL == -1e300 || aR == -1e300 || !(aL + aR >= -1e-14)) return false;
    const double span = sR - sL;
    h.vi[0] = (uint32_t)iL; h.vi[1] = (uint32_t)iR; h.vi[2] = (uint32_t)iL;
    h.c0 = sR / span; h.c1 = -sL / span; h.c2 = 0.0;
    return true;
}

// signed distance (inward > 0) of a position from the straight line of hull side k
__device__ __forceinline__ double side_dist(const WalkCert &wc, int k, double qx, double qy)
{
    return cross2(wc.c[k], wc.c[(k + 1) & 3], D2{ qx, qy }) * wc.inv_len[k];
}

// Beyond the noise band of a side: outside the convex hull, whatever the mesh looks like (the certificate bounds the border's
// deviation from the four straight sides by delta).  Asked BEFORE the search: a node out there used to run the Newton
// steps into a border cell and fourteen cell tests around it before the same four distances said "outside" -- on BASELINE
// config 3 (scaling 0.9: a fifth of the frame is uncovered) that was the bulk of the kernel's work.
__device__ __forceinline__ bool clearly_outside(const WalkCert &wc, double qx, double qy)
{
    const double lim = -2.0 * wc.delta - 1e-12;
    return side_dist(wc, 0, qx, qy) < lim || side_dist(wc, 1, qx, qy) < lim || side_dist(wc, 2, qx, qy) < lim || side_dist(wc, 3, qx, qy) < lim;
}

// A position no mesh triangle covers is outside the convex hull -- unless it sits within the noise band of a border
// side, where the hull of the (almost collinear) border points decides.
__device__ bool hull_band_locate(const float *__restrict__ flow, int sign, int H, int W, const WalkCert &wc,
                                 double qx, double qy, Hit &h, uint32_t *__restrict__ fail)
{
    bool outside = false, inside_all = true;
    int near_mask = 0;
#pragma unroll
    for (int side = 0; side < 4; ++side) {
        const double d = side_dist(wc, side, qx, qy);                 // inward > 0
        if (d < -2.0 * wc.delta - 1e-12) outside = true;
        else if (d <= 2.0 * wc.delta + 1e-12) near_mask |= 1 << side;
    }
    if (outside) return false;
    if (!near_mask) {
        if (fail) atomicAdd(fail, 1u);            // well inside the hull and still no triangle: the certificate was wrong
        return false;
    }
    bool first = true;
    for (int side = 0; side < 4 && inside_all; ++side) {
        if (!((near_mask >> side) & 1)) continue;
        Hit hs;
        if (!side_scan(flow, sign, H, W, side, wc, qx, qy, hs)) inside_all = false;
        else if (first) { h = hs; first = false; }
    }
    return inside_all;
}

// The same question for a grid node in float32, asked first: the plane form of the four distances costs a tenth of the
// float64 cross products, and its limit leaves room for its rounding (set_planes), so "yes" here implies "yes" there; the
// nodes in between fall through to the search and to hull_band_locate, which decides with the float64 distances as before.
__device__ __forceinline__ bool clearly_outside32(const WalkCert &wc, float x, float y)
{
    bool out = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) out |= fmaf(wc.pa[k], x, fmaf(wc.pb[k], y, wc.pc[k])) < wc.lim32;
    return out;
}

#ifndef OFL_WALK_WAVES
#define OFL_WALK_WAVES 5      // waves per SIMD the allocator leaves room for: the kernel is VALU-bound (5: 156 us, 7: 159 us, 8 spills: 246 us at 4K)
#endif
template <typename VT, int SP, bool BITS>
__global__ __launch_bounds__(256, OFL_WALK_WAVES)
void scatter_walk_kernel(const float *__restrict__ flow, const VT *__restrict__ vals, int C,
                         const uint8_t *__restrict__ vmask, int H, int W, int row0, int rows,
                         VT *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule, WalkCert wc,
                         uint32_t *__restrict__ fail, int tiling)
{
    // tiling bit 1: tiles of 64 x 4 nodes instead of 32 x 8; bit 0: a 1-D grid whose workgroup b (XCD b % 8) takes tile
    // (b % 8) * chunk + b / 8 -- every XCD sweeps ONE band of rows, so the halo lines of neighbouring tiles meet in one L2
    const int tws = (tiling & 2) ? 6 : 5;
    int bx = blockIdx.x, by = blockIdx.y;
    if (tiling & 1) {
        const int gx = (W + (1 << tws) - 1) >> tws, gy = (rows + (256 >> tws) - 1) / (256 >> tws);
        const int rs = (tiling >> 2) & 3, xcd = (int)(blockIdx.x & 7), j = (int)(blockIdx.x >> 3);
        if (rs == 0) {                                           // one contiguous band of tiles per XCD
            const int nb = gx * gy, chunk = (nb + 7) >> 3, tile = xcd * chunk + j;
            if (j >= chunk || tile >= nb) return;
            by = tile / gx; bx = tile - by * gx;
        } else {                                                 // bands of 2^rs tile rows dealt round-robin to the XCDs, walked column by column
            const int sr = (int)blockIdx.y, jj = j;              // (grid: 8 * (gx << rs) by rounds -- no division)
            bx = jj >> rs; by = (((sr << 3) + xcd) << rs) + (jj & ((1 << rs) - 1));
            if (by >= gy) return;
        }
    }
    const int x = (bx << tws) + (threadIdx.x & ((1 << tws) - 1));
    const int yl = by * (256 >> tws) + (threadIdx.x >> tws), y = row0 + yl;
    if (x >= W || yl >= rows) return;
    const size_t o = (size_t)yl * W + x;
    Hit h;
    const int sign = SP;
    bool found = false;
    if (!clearly_outside32(wc, (float)x, (float)y)) {
        found = walk_locate<SP, BITS, true>(flow, H, W, x, y, (double)x, (double)y, h, wc);
        if (!found) found = hull_band_locate(flow, sign, H, W, wc, (double)x, (double)y, h, fail);
    }
    if (found && sizeof(VT) == 4 && C == 2 && !(valid_rule & OFL_SCATTER_ROUND) && (((uintptr_t)vals | (uintptr_t)out) & 7) == 0) {
        // two-channel float32 values (flow fields: every scatter of the Flow algebra): resolve_emit's arithmetic on 8-byte loads
        // and one 8-byte store; the negation (exact) is a sign flip of the rounded result
        const float2 *v2 = reinterpret_cast<const float2 *>(vals);
        const float2 a = v2[h.vi[0]], b = v2[h.vi[1]], c = v2[h.vi[2]];
        const double vu = h.c0 * (double)a.x + h.c1 * (double)b.x + h.c2 * (double)c.x;
        const double vv = h.c0 * (double)a.y + h.c1 * (double)b.y + h.c2 * (double)c.y;
        const uint32_t flip = (valid_rule & OFL_SCATTER_NEGATE) ? 0x80000000u : 0u;
        float2 r;
        r.x = __uint_as_float(__float_as_uint((float)vu) ^ flip);
        r.y = __uint_as_float(__float_as_uint((float)vv) ^ flip);
        reinterpret_cast<float2 *>(out)[o] = r;
        if (valid) {
            double m;
            if (vmask) m = h.c0 * (double)(vmask[h.vi[0]] != 0) + h.c1 * (double)(vmask[h.vi[1]] != 0) + h.c2 * (double)(vmask[h.vi[2]] != 0);
            else       m = h.c0 + h.c1 + h.c2;
            const float mf = (float)m;
            const int vr = valid_rule & 3;
            valid[o] = vr == 0 ? (mf == 1.0f) : (vr == 1 ? (m > 0.99) : (rint(m) == 1.0));
        }
    } else if (found) {
        const size_t vi[3] = { h.vi[0], h.vi[1], h.vi[2] };
        resolve_emit(vals, C, vmask, vi, h.c0, h.c1, h.c2, valid_rule, out, valid, o);
    } else {
        for (int c = 0; c < C; ++c) out[o * C + c] = (VT)0;              // NaN -> 0, utils.py:254
        if (valid) valid[o] = 0;
    }
}

// The same for arbitrary positions (mode 2 / ref 't', flow_class.py:1398-1410: query [n][2] float32, float32 results and
// validity; point tracking, utils.py:603-615: query [n][2] float64, float64 results and found flags).
template <int SP, bool SPARSE>
__global__ __launch_bounds__(256)
void scatter_walk_query_kernel(const float *__restrict__ flow, const float *__restrict__ vals, int C,
                               const uint8_t *__restrict__ vmask, int H, int W, const void *__restrict__ query, size_t n,
                               void *__restrict__ out, uint8_t *__restrict__ valid, int valid_rule, WalkCert wc,
                               uint32_t *__restrict__ fail)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        double qx, qy;
        if (SPARSE) { qx = ((const double *)query)[2 * i]; qy = ((const double *)query)[2 * i + 1]; }
        else { const float2 q = ((const float2 *)query)[i]; qx = (double)q.x; qy = (double)q.y; }
        Hit h;
        bool found = false;
        if (qx == qx && qy == qy && !clearly_outside(wc, qx, qy)) {
            const int nx = (int)fmin(fmax(rint(qx), 0.0), (double)(W - 1)), ny = (int)fmin(fmax(rint(qy), 0.0), (double)(H - 1));
            found = walk_locate<SP, false>(flow, H, W, nx, ny, qx, qy, h, wc);
            if (!found) found = hull_band_locate(flow, SP, H, W, wc, qx, qy, h, fail);
        }
        if (SPARSE) {
            double *o = (double *)out + i * C;
            for (int c = 0; c < C; ++c)
                o[c] = found ? h.c0 * (double)vals[(size_t)h.vi[0] * C + c] + h.c1 * (double)vals[(size_t)h.vi[1] * C + c] +
                               h.c2 * (double)vals[(size_t)h.vi[2] * C + c] : 0.0;
            valid[i] = found ? 1 : 0;
        } else if (found) {
            const size_t vi[3] = { h.vi[0], h.vi[1], h.vi[2] };
            resolve_emit(vals, C, vmask, vi, h.c0, h.c1, h.c2, valid_rule, (float *)out, valid, i);
        } else {
            for (int c = 0; c < C; ++c) ((float *)out)[i * C + c] = 0.0f;        // NaN -> 0 / fill_value = 0
            if (valid) valid[i] = 0;
        }
    }
}

}  // namespace

namespace ofl_sc {

// float32 planes of the four hull sides for clearly_outside32.  With unit normals and |x|, |y| < W + H the float32
// evaluation is off by less than (W + H + |pc|) * 2^-21 (coefficient rounding + two fused steps); the limit steps back by four
// times that and a hundredth of a pixel.
static void set_planes(WalkCert &wc, int H, int W)
{
    double worst = 0.0;
    for (int k = 0; k < 4; ++k) {
        const D2 a = wc.c[k], b = wc.c[(k + 1) & 3];
        const double A = -(b.y - a.y) * wc.inv_len[k], B = (b.x - a.x) * wc.inv_len[k], C = -(A * a.x + B * a.y);
        wc.pa[k] = (float)A; wc.pb[k] = (float)B; wc.pc[k] = (float)C;
        worst = fmax(worst, fabs(C));
    }
    const double slack = 4.0 * ((double)W + (double)H + worst) / 2097152.0 + 0.01;
    wc.lim32 = (float)(-2.0 * wc.delta - slack);
    if (!(wc.lim32 == wc.lim32)) wc.lim32 = -INFINITY;               // (never with a certificate: delta <= kBorderMax)
}

// host side of the certificate: launch, ONE small read-back, evaluation
int certify_mesh(const float *flow, int sign_pp, const uint8_t *pmask, int H, int W, void *scratch128,
                 ofl_mesh_cert *cert, uint32_t *diag_bits, hipStream_t s)
{
    memset(cert, 0, sizeof(*cert));
    if (H < 2 || W < 2) return OFL_OK;                                   // no cells: never certified
    // the record travels through a pinned staging buffer of the calling thread (two halves: the initial values up, the result down):
    // copies from / to pageable memory go through the runtime's own bounce buffers and cost this call -- a kernel, a read-back,
    // a synchronisation -- a third of its 60 us of host time
    static thread_local char *pinned = nullptr;
    if (!pinned) { void *h = nullptr; if (hipHostMalloc(&h, 512, hipHostMallocDefault) == hipSuccess) pinned = (char *)h; else (void)hipGetLastError(); }
    CertDev init_stack;
    CertDev &init = pinned ? *reinterpret_cast<CertDev *>(pinned) : init_stack;
    memset(&init, 0, sizeof(init));
    for (int k = 0; k < 4; ++k) { init.dev_min[k] = ~0ull; init.dev_max[k] = 0ull; }
    CertDev *dev = (CertDev *)scratch128;
    OFL_HIP(hipMemcpyAsync(dev, &init, sizeof(init), hipMemcpyHostToDevice, s));
    static_assert(sizeof(CertDev) <= kSlabStampAt, "the certificate record must end before the slab stamp");
    OFL_HIP(hipMemsetAsync((char *)scratch128 + kSlabStampAt, 0, 4, s));          // (callers give at least 256 bytes) a slab state in this workspace is void now
    const int tiles_x = (W + 31) / 32, tile_rows = (H + 7) / 8, first = tile_rows >= 32 ? tile_rows / 16 : tile_rows;
    double *affine = (double *)((char *)scratch128 + kCertAffineAt);              // (behind the record and the slab stamp, within the 256 bytes every caller gives)
    static_assert(kCertAffineAt >= sizeof(CertDev) && kCertAffineAt >= kSlabStampAt + 4 && kCertAffineAt + 7 * sizeof(double) <= 256, "certificate scratch layout");
    hipLaunchKernelGGL(cert_affine_kernel, dim3(1), dim3(1), 0, s, flow, sign_pp, H, W, affine,
                       diag_bits ? diag_bits + (size_t)(H - 1) * ((W + 31) / 32) : (uint32_t *)nullptr);      // (the "not affine" word: row H - 1 of the plane)
    hipLaunchKernelGGL(scatter_certify_kernel, dim3(tiles_x, first), dim3(256), 0, s, flow, sign_pp, pmask, H, W, dev, diag_bits, tiles_x, (const double *)affine, 0, tile_rows);
    if (first < tile_rows)
        hipLaunchKernelGGL(scatter_certify_kernel, dim3(tiles_x, tile_rows - first), dim3(256), 0, s, flow, sign_pp, pmask, H, W, dev, diag_bits, tiles_x,
                           (const double *)affine, first, tile_rows);
    cert->diag_bits = diag_bits;
    OFL_HIP(hipGetLastError());
    CertDev r_stack;
    CertDev &r = pinned ? *reinterpret_cast<CertDev *>(pinned + 256) : r_stack;
    OFL_HIP(hipMemcpyAsync(&r, dev, sizeof(r), hipMemcpyDeviceToHost, s));
    OFL_HIP(hipStreamSynchronize(s));
    cert->folded_cells = r.folded;
    cert->bad_edges = r.bad_edges;
    cert->dropped = r.dropped;
    double dev_abs = 0.0;
    for (int k = 0; k < 4; ++k) {
        if (r.dev_min[k] == ~0ull) { dev_abs = INFINITY; break; }
        dev_abs = fmax(dev_abs, fmax(fabs(okey_inv(r.dev_min[k])), fabs(okey_inv(r.dev_max[k]))));
    }
    cert->border_dev = dev_abs;
    bool convex = !r.degenerate;
    for (int k = 0; k < 4; ++k) {
        cert->corner[k][0] = r.corner[k].x; cert->corner[k][1] = r.corner[k].y;
        const D2 a = r.corner[k], b = r.corner[(k + 1) & 3], c = r.corner[(k + 2) & 3];
        if (!((b.x - a.x) * (c.y - a.y) - (b.y - a.y) * (c.x - a.x) > 0.0)) convex = false;
    }
    cert->certified = (r.folded == 0 && r.bad_edges == 0 && r.dropped == 0 && convex && dev_abs <= kBorderMax) ? 1u : 0u;
    return OFL_OK;
}

template <typename VT>
int walk_launch(const float *flow, int sign_pp, const VT *vals, int C, const uint8_t *vmask, int H, int W,
                int row0, int rows, VT *out, uint8_t *valid, int valid_rule, const ofl_mesh_cert *cert,
                uint32_t *fail_dev, hipStream_t s)
{
    WalkCert wc;
    for (int k = 0; k < 4; ++k) { wc.c[k].x = cert->corner[k][0]; wc.c[k].y = cert->corner[k][1]; }
    wc.delta = cert->border_dev;
    for (int k = 0; k < 4; ++k) {
        const double dx = wc.c[(k + 1) & 3].x - wc.c[k].x, dy = wc.c[(k + 1) & 3].y - wc.c[k].y;
        wc.inv_len[k] = 1.0 / sqrt(dx * dx + dy * dy);
    }
    wc.diag = cert->diag_bits; wc.diag_stride = (W + 31) / 32;
    set_planes(wc, H, W);
    wc.not_affine = nullptr;
    for (int k = 0; k < 6; ++k) wc.ai[k] = 0.0;
    if (wc.diag && H >= 2 && W >= 2 && corner_affine_inverse(wc.c[0], wc.c[1], wc.c[3], H, W, wc.ai)) wc.not_affine = wc.diag + (size_t)(H - 1) * wc.diag_stride;
    const int tiling = OFL_KNOB_INT("OFL_WALK_TILING", 0);
    const int tws = (tiling & 2) ? 6 : 5, gx = (W + (1 << tws) - 1) >> tws, gy = (rows + (256 >> tws) - 1) / (256 >> tws);
    const int rs = (tiling >> 2) & 3, bands = ((gy + (1 << rs) - 1) >> rs), rounds = (bands + 7) / 8;
    const dim3 grid(!(tiling & 1) ? gx : rs == 0 ? 8 * (((size_t)gx * gy + 7) / 8) : (size_t)8 * (gx << rs), !(tiling & 1) ? gy : rs == 0 ? 1 : rounds), block(256);
    // (a variant that took flow-valued targets -- invert, switch_ref -- and their mask bytes straight from the registers of the
    // cell test was measured SLOWER twice, in round 2 (168 vs 157 us at 4K) and again on this leaner kernel (85 vs 77 us): keeping
    // the corner loads alive to the end costs 30 VGPRs, more than the three cached reloads it saves; fetching only the corners'
    // mask bytes with the cell -- 6 VGPRs -- changed nothing: 78.7 vs 78.2 us)
#define OFL_WALK_LAUNCH(SP)                                                                                                       \
    do { if (wc.diag) hipLaunchKernelGGL((scatter_walk_kernel<VT, SP, true>), grid, block, 0, s, flow, vals, C, vmask, H, W, row0, rows, \
                                         out, valid, valid_rule, wc, fail_dev, tiling);                                                  \
         else hipLaunchKernelGGL((scatter_walk_kernel<VT, SP, false>), grid, block, 0, s, flow, vals, C, vmask, H, W, row0, rows, out,  \
                                 valid, valid_rule, wc, fail_dev, tiling); } while (0)
    if (sign_pp == 1) OFL_WALK_LAUNCH(1); else if (sign_pp == -1) OFL_WALK_LAUNCH(-1);
    else if (sign_pp == 2) OFL_WALK_LAUNCH(2); else OFL_WALK_LAUNCH(-2);
#undef OFL_WALK_LAUNCH
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int walk_query_launch(const float *flow, int sign_pp, const float *vals, int C, const uint8_t *vmask, int H, int W,
                      const void *query, size_t n, bool sparse, void *out, uint8_t *valid, int valid_rule,
                      const ofl_mesh_cert *cert, uint32_t *fail_dev, hipStream_t s)
{
    WalkCert wc;
    for (int k = 0; k < 4; ++k) { wc.c[k].x = cert->corner[k][0]; wc.c[k].y = cert->corner[k][1]; }
    wc.delta = cert->border_dev;
    for (int k = 0; k < 4; ++k) {
        const double dx = wc.c[(k + 1) & 3].x - wc.c[k].x, dy = wc.c[(k + 1) & 3].y - wc.c[k].y;
        wc.inv_len[k] = 1.0 / sqrt(dx * dx + dy * dy);
    }
    wc.diag = nullptr; wc.diag_stride = 0; wc.not_affine = nullptr;
    for (int k = 0; k < 6; ++k) wc.ai[k] = 0.0;
    // (the query kernel decides diagonals itself, but the certificate's "not affine" word -- in the last row of its plane -- serves it too)
    if (cert->diag_bits && H >= 2 && W >= 2 && corner_affine_inverse(wc.c[0], wc.c[1], wc.c[3], H, W, wc.ai)) wc.not_affine = cert->diag_bits + (size_t)(H - 1) * ((W + 31) / 32);
    set_planes(wc, H, W);
    if (n == 0) return OFL_OK;
    const size_t nb = (n + 255) / 256;
    const dim3 grid((unsigned)(nb < (1u << 20) ? nb : (1u << 20))), block(256);
#define OFL_WQ_LAUNCH(SP, SPARSE)                                                                                    \
    hipLaunchKernelGGL((scatter_walk_query_kernel<SP, SPARSE>), grid, block, 0, s, flow, vals, C, vmask, H, W, query, \
                       n, out, valid, valid_rule, wc, fail_dev)
    if (sparse) {
        if (sign_pp == 1) OFL_WQ_LAUNCH(1, true); else if (sign_pp == -1) OFL_WQ_LAUNCH(-1, true);
        else if (sign_pp == 2) OFL_WQ_LAUNCH(2, true); else OFL_WQ_LAUNCH(-2, true);
    } else {
        if (sign_pp == 1) OFL_WQ_LAUNCH(1, false); else if (sign_pp == -1) OFL_WQ_LAUNCH(-1, false);
        else if (sign_pp == 2) OFL_WQ_LAUNCH(2, false); else OFL_WQ_LAUNCH(-2, false);
    }
#undef OFL_WQ_LAUNCH
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

template int walk_launch<float>(const float *, int, const float *, int, const uint8_t *, int, int, int, int, float *,
                                uint8_t *, int, const ofl_mesh_cert *, uint32_t *, hipStream_t);
template int walk_launch<double>(const float *, int, const double *, int, const uint8_t *, int, int, int, int, double *,
                                 uint8_t *, int, const ofl_mesh_cert *, uint32_t *, hipStream_t);

}  // namespace ofl_sc
