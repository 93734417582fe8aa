// ofl_scatter_dev.h -- device helpers shared by the scatter kernels (K3): warped point positions in the precision
// SciPy sees them, orientation / in-circle predicates, the cell-wise triangle numbering, SciPy's inclusion rule and
// the per-node value / validity emission (utils.py:253-258, flow_class.py:668, :1410).
#pragma once
#include "ofl_common.h"

namespace ofl_sc {

constexpr double   kEps       = 100.0 * 2.220446049250313e-16;   // scipy _qhull: eps = 100 * DBL_EPSILON
// Byte offset, in the first 256 bytes of a scatter workspace, of the stamp step 1 of the slab-wise scatter leaves for step 2
// (ofl_delaunay.hip: DlHead::slab_stamp, static_assert there).  The certificate pass writes its own 144-byte record over the
// front of the same header: it clears the stamp, so that a slab state it has trampled on is refused by step 2.
constexpr size_t   kSlabStampAt = 184;

struct D2 { double x, y; };

// `sign` carries the point precision in bit 1 of its magnitude: +-1 = float64 positions (utils.py:242),
// +-2 = positions rounded to float32 first (flow_class.py:1398-1400 builds them in a float32 array).
__device__ __forceinline__ D2 point_of(const float *flow, int sign, int W, int x, int y)
{
    const float2 f = *reinterpret_cast<const float2 *>(flow + ((size_t)y * W + x) * 2);
    D2 p;
    p.x = sign >= 0 ? (double)x + (double)f.x : (double)x - (double)f.x;
    p.y = sign >= 0 ? (double)y + (double)f.y : (double)y - (double)f.y;
    if (sign == 2 || sign == -2) { p.x = (double)(float)p.x; p.y = (double)(float)p.y; }
    return p;
}

__device__ __forceinline__ double cross2(const D2 &o, const D2 &a, const D2 &b)
{
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}

// > 0 when d lies inside the circumcircle of the counter-clockwise triangle (a, b, c)
__device__ __forceinline__ double incircle(const D2 &a, const D2 &b, const D2 &c, const D2 &d)
{
    const double ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    return ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx);
}

// incircle with "on the circle as far as float64 can tell" made explicit: 0 within 4e-15 of the sum of the absolute terms
// (Shewchuk's static filter for this expression is 1.1e-15).  Every place that decides a diagonal -- here, the mesh-cell and
// mesh-fan passes and the clip pass of the Delaunay path (ofl_dl::incircle_origin_filtered) -- evaluates a DIFFERENT
// expression of the same four sites; for sites that are co-circular in exact arithmetic (similarity transforms, lattices)
// their results are rounding noise of either sign, and only "all call it a tie, the index rule decides" makes them agree.
__device__ __forceinline__ double incircle_filtered(const D2 &a, const D2 &b, const D2 &c, const D2 &d)
{
    const double ax = a.x - d.x, ay = a.y - d.y, bx = b.x - d.x, by = b.y - d.y, cx = c.x - d.x, cy = c.y - d.y;
    const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by, c2 = cx * cx + cy * cy;
    const double det = ax * (by * c2 - b2 * cy) - ay * (bx * c2 - b2 * cx) + a2 * (bx * cy - by * cx);
    const double perm = fabs(ax) * (fabs(by) * c2 + b2 * fabs(cy)) + fabs(ay) * (fabs(bx) * c2 + b2 * fabs(cx)) + a2 * (fabs(bx * cy) + fabs(by * cx));
    return fabs(det) > 4e-15 * perm ? det : 0.0;
}

// Which diagonal splits the warped cell a=P(x,y), b=P(x+1,y), c=P(x+1,y+1), d=P(x,y+1):
// returns 0 for a-c, 1 for b-d.
__device__ __forceinline__ int pick_diagonal(const D2 &a, const D2 &b, const D2 &c, const D2 &d)
{
    const double o_abc = cross2(a, b, c), o_acd = cross2(a, c, d);
    const double o_bcd = cross2(b, c, d), o_bda = cross2(b, d, a);
    const bool ac_ok = (o_abc > 0) == (o_acd > 0) && o_abc != 0 && o_acd != 0;   // both halves same orientation
    const bool bd_ok = (o_bcd > 0) == (o_bda > 0) && o_bcd != 0 && o_bda != 0;
    if (ac_ok && !bd_ok) return 0;
    if (bd_ok && !ac_ok) return 1;
    if (!ac_ok && !bd_ok) return 0;                 // folded cell: the reference is arbitrary here too
    double ic = incircle_filtered(a, b, c, d);      // convex cell: Delaunay criterion (a tie keeps a-c: a is the cell's smallest index)
    if (o_abc < 0) ic = -ic;
    return ic > 0 ? 1 : 0;                          // d inside circle(a, b, c) -> a-c is illegal
}

// vertices of triangle t (0/1) of a cell split along `diag`; corner numbers a=0, b=1, c=2, d=3:
//   diag 0: (a, b, c), (a, c, d)        diag 1: (b, c, d), (b, d, a)
__device__ __forceinline__ void tri_corners(int diag, int t, int &i0, int &i1, int &i2)
{
    i0 = diag;
    i1 = diag + 1 + t;
    i2 = (diag + 2 + t) & 3;
}

// Value selection by corner number WITHOUT indexing a register array at run time (hipcc would place
// such an array in scratch memory).
template <typename T>
__device__ __forceinline__ T pick4(int i, const T &a, const T &b, const T &c, const T &d)
{
    return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d));
}

__device__ __forceinline__ D2 pick4(int i, const D2 &a, const D2 &b, const D2 &c, const D2 &d)
{
    D2 r;
    r.x = i == 0 ? a.x : (i == 1 ? b.x : (i == 2 ? c.x : d.x));
    r.y = i == 0 ? a.y : (i == 1 ? b.y : (i == 2 ? c.y : d.y));
    return r;
}

__device__ __forceinline__ bool bary(const D2 &p0, const D2 &p1, const D2 &p2, double gx, double gy,
                                     double &c0, double &c1, double &c2)
{
    const double e1x = p1.x - p0.x, e1y = p1.y - p0.y, e2x = p2.x - p0.x, e2y = p2.y - p0.y;
    const double det = e1x * e2y - e1y * e2x;
    if (det == 0.0) return false;
    const double dx = gx - p0.x, dy = gy - p0.y;
    c1 = (dx * e2y - dy * e2x) / det;
    c2 = (e1x * dy - e1y * dx) / det;
    c0 = 1.0 - c1 - c2;
    return c0 >= -kEps && c0 <= 1.0 + kEps && c1 >= -kEps && c1 <= 1.0 + kEps && c2 >= -kEps && c2 <= 1.0 + kEps;
}

// Inside test of the raster passes, division-free: c_i >= -eps  <=>  w_i * sign(det) >= -eps * |det| with the
// three edge functions w_i of the barycentric numerators (c_i <= 1 + eps follows from the other two
// being >= -eps up to O(eps)).  The resolve pass evaluates the coordinates themselves.
struct TriEdge { D2 p0; double e1x, e1y, e2x, e2y, det, tol; };

__device__ __forceinline__ bool tri_setup(const D2 &p0, const D2 &p1, const D2 &p2, TriEdge &t)
{
    t.p0 = p0;
    t.e1x = p1.x - p0.x; t.e1y = p1.y - p0.y; t.e2x = p2.x - p0.x; t.e2y = p2.y - p0.y;
    t.det = t.e1x * t.e2y - t.e1y * t.e2x;
    t.tol = kEps * fabs(t.det);
    return t.det != 0.0;
}

__device__ __forceinline__ bool tri_inside(const TriEdge &t, double gx, double gy)
{
    const double dx = gx - t.p0.x, dy = gy - t.p0.y;
    double w1 = dx * t.e2y - dy * t.e2x, w2 = t.e1x * dy - t.e1y * dx;
    double w0 = t.det - w1 - w2;
    if (t.det < 0) { w0 = -w0; w1 = -w1; w2 = -w2; }
    return w0 >= -t.tol && w1 >= -t.tol && w2 >= -t.tol;
}

struct TriBox { int x0, x1, y0, y1; };

__device__ __forceinline__ TriBox tri_box(const D2 &p0, const D2 &p1, const D2 &p2, int W, int H)
{
    const double xmin = fmin(p0.x, fmin(p1.x, p2.x)), xmax = fmax(p0.x, fmax(p1.x, p2.x));
    const double ymin = fmin(p0.y, fmin(p1.y, p2.y)), ymax = fmax(p0.y, fmax(p1.y, p2.y));
    TriBox b;
    // clamp in double first: positions may be astronomically large
    b.x0 = (int)fmax(ceil(xmin - 1e-9), 0.0);
    b.y0 = (int)fmax(ceil(ymin - 1e-9), 0.0);
    b.x1 = (int)fmin(floor(xmax + 1e-9), (double)(W - 1));
    b.y1 = (int)fmin(floor(ymax + 1e-9), (double)(H - 1));
    if (!(xmax >= -1.0) || !(ymax >= -1.0)) { b.x1 = -1; b.y1 = -1; }
    return b;
}

// values and validity of one output element from its triangle (vi) and barycentric coordinates
template <typename VT>      // float, or double for float64 targets (griddata's own precision, utils.py:253)
__device__ __forceinline__ void resolve_emit(const VT *__restrict__ vals, int C, const uint8_t *__restrict__ vmask,
                                             const size_t (&vi)[3], double c0, double c1, double c2, int valid_rule,
                                             VT *__restrict__ out, uint8_t *__restrict__ valid, size_t o)
{
    for (int c = 0; c < C; ++c) {
        const double v = c0 * (double)vals[vi[0] * C + c] + c1 * (double)vals[vi[1] * C + c] + c2 * (double)vals[vi[2] * C + c];
        const double r = (valid_rule & OFL_SCATTER_ROUND) ? rint(v) : v;             // np.round of the float64 result, utils.py:256-257
        out[o * C + c] = (VT)((valid_rule & OFL_SCATTER_NEGATE) ? -r : r);          // values = -vals (Flow.invert: apply(-self)), exact
    }
    if (valid) {
        double m = 1.0;
        if (vmask) m = c0 * (double)(vmask[vi[0]] != 0) + c1 * (double)(vmask[vi[1]] != 0) + c2 * (double)(vmask[vi[2]] != 0);
        else       m = c0 + c1 + c2;
        const float mf = (float)m;                                   // result.astype(target.dtype), utils.py:258
        const int vr = valid_rule & 3;                               // flow_class.py:668 / :1410; 2: integer-typed targets, where
        valid[o] = vr == 0 ? (mf == 1.0f) : (vr == 1 ? (m > 0.99) : (rint(m) == 1.0));   // the mask channel is np.round-ed first (utils.py:256)
    }
}

// host side of the certified fast path (ofl_scatter_walk.hip); sign_pp = sign with the point precision folded in (+-1 / +-2)
// diag_bits (device, H * ((W + 31) / 32) words, or NULL): receives the Delaunay diagonal of every cell, one bit each
int certify_mesh(const float *flow, int sign_pp, const uint8_t *pmask, int H, int W, void *scratch256,
                 ofl_mesh_cert *cert, uint32_t *diag_bits, hipStream_t s);
template <typename VT>
int walk_launch(const float *flow, int sign_pp, const VT *vals, int C, const uint8_t *vmask, int H, int W,
                int row0, int rows, VT *out, uint8_t *valid, int valid_rule, const ofl_mesh_cert *cert,
                uint32_t *fail_dev, hipStream_t s);

// arbitrary positions on a certified mesh: query = float32 [n][2] (dense: float32 out [n][C] + validity by valid_rule) or
// float64 [n][2] (sparse: float64 out [n][C] + found flags)
int walk_query_launch(const float *flow, int sign_pp, const float *vals, int C, const uint8_t *vmask, int H, int W,
                      const void *query, size_t n, bool sparse, void *out, uint8_t *valid, int valid_rule,
                      const ofl_mesh_cert *cert, uint32_t *fail_dev, hipStream_t s);

// exact path (ofl_delaunay.hip): Delaunay triangulation of the kept points on the GPU
size_t exact_workspace_bytes(int H, int W);
template <typename VT>
int exact_scatter(const float *flow, int sign_pp, const uint8_t *pmask, const VT *vals, int C, const uint8_t *vmask,
                  int H, int W, int row0, int rows, VT *out, uint8_t *valid, int valid_rule,
                  void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s);
int exact_query(const float *flow, int sign_pp, const uint8_t *pmask, const float *vals, int C, const uint8_t *vmask,
                int H, int W, const void *query, size_t n, bool sparse, void *out, uint8_t *valid, int valid_rule,
                void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s);

// slab mode of the exact path (one row band per rank with ONE exchange of the unfinished sites in between: include/ofl.h)
int exact_slab_stars(const float *flow, int sign_pp, const uint8_t *pmask, int H, int W, int row0, int rows,
                     uint32_t *list, size_t list_bytes, void *workspace, size_t workspace_bytes, hipStream_t s);
int exact_slab_finish(const float *flow, int sign_pp, const float *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows,
                      const uint32_t *lists, size_t stride_bytes, int n_lists, float *out, uint8_t *valid, int valid_rule,
                      void *workspace, size_t workspace_bytes, uint64_t *info_host, hipStream_t s);

}  // namespace ofl_sc
