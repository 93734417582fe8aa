// ofl_scatter.hip -- K3: scattered -> regular-grid linear interpolation for gfx950, entry points and dispatch.
//
// Replaces scipy.interpolate.griddata(points, values, grid, 'linear') as the reference uses it
// (src/oflibnumpy/utils.py:237-258; flow_class.py:1398-1410; utils.py:599-615): the scattered points are the regular
// grid displaced by a flow.  Two device paths produce SciPy's result:
//
//   certified   (ofl_scatter_walk.hip)  one pass over the flow proves that the cell-wise mesh of the warped grid IS the
//               Delaunay triangulation Qhull builds (no folds, every edge locally Delaunay, no dropped point, straight
//               border); every output node -- or query position -- then finds its triangle by Newton steps on the
//               piecewise-affine map: one kernel, no atomics, no owner map.  All affine fields take this path.
//   Delaunay    (ofl_delaunay.hip)      everything else: the kept points are bucket-sorted, every point builds its
//               Delaunay star (Voronoi cell by half-plane clipping with a security radius), the stars' triangles are
//               scan-converted into an owner map and resolved; query positions walk through the stars.
//
// There is no approximate path: wherever SciPy's triangulation is unique the interpolant is the same function
// (tests/test_gpu_scatter*.py against outputs of the real reference).
#include "ofl_common.h"
#include "ofl_scatter_dev.h"
#include <algorithm>

using namespace ofl;
using namespace ofl_sc;

namespace {

constexpr size_t kMinWorkspace = 4096;            // the certificate's device record (first 256 bytes) + the walk kernel's failure counter
constexpr size_t kWalkFailAt   = 1024;            // byte offset of that counter
size_t diag_bytes(int H, int W) { return (size_t)H * (size_t)((W + 31) / 32) * 4; }     // the certificate's diagonal bit-plane
// The one-shot entries keep the plane at the END of the workspace (the Delaunay path carves from the front and is only
// entered when the certificate fails or the walk loses nodes -- by then the plane is no longer needed).
uint32_t *diag_in(void *workspace, size_t workspace_bytes, int H, int W)
{
    const size_t need = (diag_bytes(H, W) + 255) / 256 * 256;
    return workspace_bytes >= kMinWorkspace + need ? (uint32_t *)((char *)workspace + (workspace_bytes - need) / 256 * 256) : nullptr;
}

// A certificate proves that the cell-wise mesh IS the Delaunay triangulation, not that the walk kernel's Newton steps find
// every node's triangle in it (a certified field may compress 99 % of the image into a corner): the kernel counts the nodes
// it could not locate although they lie well inside the hull, the entry reads the count back -- it has synchronised for the
// certificate already -- and a non-zero count sends the call through the Delaunay path, which searches nothing.
int walk_failures(void *workspace, hipStream_t s, uint32_t &n)
{
    OFL_HIP(hipMemcpyAsync(&n, (char *)workspace + kWalkFailAt, 4, hipMemcpyDeviceToHost, s));
    OFL_HIP(hipStreamSynchronize(s));
    return OFL_OK;
}

int check_common(const char *who, const float *flow, int &sign, int point_precision, int C, const void *vals, const void *out,
                 const uint8_t *valid, int valid_rule, int H, int W, void *workspace, size_t workspace_bytes)
{
    if (!flow || !workspace) return fail(OFL_E_INVALID, "%s: NULL pointer", who);
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 29)) return fail(OFL_E_INVALID, "%s: H*W must be in [1, 2^29)", who);
    if (C < 0 || (C > 0 && (!vals || !out))) return fail(OFL_E_INVALID, "%s: C > 0 needs vals and out", who);
    if (C == 0 && !valid) return fail(OFL_E_INVALID, "%s: nothing to compute", who);
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "%s: sign must be +1 or -1", who);
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "%s: bad point_precision", who);
    if ((valid_rule & ~(3 | OFL_SCATTER_ROUND | OFL_SCATTER_NEGATE | OFL_SCATTER_UNCERTIFIED)) || (valid_rule & 3) == 3) return fail(OFL_E_INVALID, "%s: bad valid_rule", who);
    if (workspace_bytes < kMinWorkspace) return fail(OFL_E_INVALID, "%s: workspace too small (%zu < %zu)", who, workspace_bytes, kMinWorkspace);
    if (point_precision == 1) sign *= 2;          // the point precision rides in bit 1 of the sign's magnitude
    return OFL_OK;
}

// grid nodes: argument checks, certificate, then one of the two paths on rows [row0, row0 + rows)
template <typename VT>
int scatter_grid_impl(const char *who, const float *flow, int sign, int point_precision, const uint8_t *pmask,
                      const VT *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows,
                      VT *out, uint8_t *valid, int valid_rule, void *workspace, size_t workspace_bytes,
                      uint64_t *info_host, hipStream_t s)
{
    OFL_TRY(check_common(who, flow, sign, point_precision, C, vals, out, valid, valid_rule, H, W, workspace, workspace_bytes));
    if (row0 < 0 || rows <= 0 || row0 + rows > H)
        return fail(OFL_E_INVALID, "%s: rows [%d, %d) outside the %d-row grid", who, row0, row0 + rows, H);
    if (!(valid_rule & OFL_SCATTER_UNCERTIFIED)) {
        ofl_mesh_cert cert;
        OFL_TRY(certify_mesh(flow, sign, pmask, H, W, workspace, &cert, diag_in(workspace, workspace_bytes, H, W), s));       // a point mask without zeros drops nothing
        if (cert.certified) {
            uint32_t *fail = (uint32_t *)((char *)workspace + kWalkFailAt), n_fail = 0;
            OFL_HIP(hipMemsetAsync(fail, 0, 4, s));
            OFL_TRY(walk_launch<VT>(flow, sign, vals, C, vmask, H, W, row0, rows, out, valid, valid_rule, &cert, fail, s));
            OFL_TRY(walk_failures(workspace, s, n_fail));
            if (n_fail == 0) {
                if (info_host) { info_host[0] = (uint64_t)H * W; info_host[1] = 0; info_host[2] = 0; }
                return OFL_OK;
            }
        }
    }
    return exact_scatter<VT>(flow, sign, pmask, vals, C, vmask, H, W, row0, rows, out, valid, valid_rule,
                             workspace, workspace_bytes, info_host, s);
}

// query positions: dense float32 [H][W][2] (mode 2 / ref 't') or sparse float64 [n][2] (point tracking)
int scatter_query_impl(const char *who, const float *flow, int sign, int point_precision, const uint8_t *pmask,
                       const float *vals, int C, const uint8_t *vmask, int H, int W, const void *query, size_t n, bool sparse,
                       void *out, uint8_t *valid, int valid_rule, void *workspace, size_t workspace_bytes,
                       uint64_t *info_host, hipStream_t s)
{
    OFL_TRY(check_common(who, flow, sign, point_precision, C, vals, out, valid, valid_rule, H, W, workspace, workspace_bytes));
    if (!(valid_rule & OFL_SCATTER_UNCERTIFIED)) {
        ofl_mesh_cert cert;
        OFL_TRY(certify_mesh(flow, sign, pmask, H, W, workspace, &cert, nullptr, s));
        if (cert.certified) {
            uint32_t *fail = (uint32_t *)((char *)workspace + kWalkFailAt), n_fail = 0;
            OFL_HIP(hipMemsetAsync(fail, 0, 4, s));
            OFL_TRY(walk_query_launch(flow, sign, vals, C, vmask, H, W, query, n, sparse, out, valid, valid_rule, &cert, fail, s));
            OFL_TRY(walk_failures(workspace, s, n_fail));
            if (n_fail == 0) {
                if (info_host) { info_host[0] = (uint64_t)H * W; info_host[1] = 0; info_host[2] = 0; }
                return OFL_OK;
            }
        }
    }
    return exact_query(flow, sign, pmask, vals, C, vmask, H, W, query, n, sparse, out, valid, valid_rule,
                       workspace, workspace_bytes, info_host, s);
}

}  // namespace

extern "C" {

int ofl_scatter_workspace_bytes(int H, int W, int C, size_t *bytes)
{
    (void)C;
    if (!bytes) return fail(OFL_E_INVALID, "ofl_scatter_workspace_bytes: NULL");
    if (H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_scatter_workspace_bytes: bad shape");
    *bytes = kMinWorkspace;
    if ((long long)H * W < (1ll << 27)) *bytes = std::max(*bytes, exact_workspace_bytes(H, W));     // larger fields: certified meshes only
    *bytes += (diag_bytes(H, W) + 255) / 256 * 256 + 256;                                           // + the diagonal bit-plane of the one-shot entries
    return OFL_OK;
}

int ofl_scatter_linear_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                           const float *vals, int C, const uint8_t *vmask, int H, int W,
                           const float *query, float *out, uint8_t *valid, int valid_rule,
                           void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream)
{
    OFL_TRY(need_device());
    hipStream_t s = stream_of(stream);
    if (!query)
        return scatter_grid_impl<float>("ofl_scatter_linear", flow, sign, point_precision, pmask, vals, C, vmask, H, W, 0, H,
                                        out, valid, valid_rule, workspace, workspace_bytes, info_host, s);
    return scatter_query_impl("ofl_scatter_linear", flow, sign, point_precision, pmask, vals, C, vmask, H, W, query,
                              (size_t)H * W, false, out, valid, valid_rule, workspace, workspace_bytes, info_host, s);
}

int ofl_scatter_diag_bytes(int H, int W, size_t *bytes)
{
    if (!bytes || H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_scatter_diag_bytes: bad arguments");
    *bytes = diag_bytes(H, W);
    return OFL_OK;
}

int ofl_scatter_certify_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask, int H, int W,
                            void *workspace, size_t workspace_bytes, ofl_mesh_cert *cert_host, uint32_t *diag_bits, void *stream)
{
    OFL_TRY(need_device());
    if (!flow || !workspace || !cert_host || workspace_bytes < 256) return fail(OFL_E_INVALID, "ofl_scatter_certify: NULL pointer / workspace < 256 bytes");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 29)) return fail(OFL_E_INVALID, "ofl_scatter_certify: H*W must be in [1, 2^29)");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_scatter_certify: sign must be +1 or -1");
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "ofl_scatter_certify: bad point_precision");
    return certify_mesh(flow, point_precision ? 2 * sign : sign, pmask, H, W, workspace, cert_host, diag_bits, stream_of(stream));
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
    return scatter_grid_impl<double>("ofl_scatter_linear_f64", flow, sign, point_precision, pmask, vals, C, vmask, H, W, 0, H,
                                     out, valid, valid_rule, workspace, workspace_bytes, info_host, stream_of(stream));
}

int ofl_scatter_rows_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                         const float *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows,
                         float *out_rows, uint8_t *valid_rows, int valid_rule,
                         void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream)
{
    OFL_TRY(need_device());
    return scatter_grid_impl<float>("ofl_scatter_rows", flow, sign, point_precision, pmask, vals, C, vmask, H, W, row0, rows,
                                    out_rows, valid_rows, valid_rule, workspace, workspace_bytes, info_host, stream_of(stream));
}

// One row band per rank with the star passes sharded as well (include/ofl.h): step 1, the exchange, step 2.
int ofl_scatter_slab_stars_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask, int H, int W, int row0, int rows,
                               uint32_t *list, size_t list_bytes, void *workspace, size_t workspace_bytes, void *stream)
{
    OFL_TRY(need_device());
    if (!flow || !list || !workspace) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: NULL pointer");
    if (H <= 0 || W <= 0 || (long long)H * W >= (1ll << 27)) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: H*W must be in [1, 2^27)");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: sign must be +1 or -1");
    if (point_precision != 0 && point_precision != 1) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: bad point_precision");
    if (row0 < 0 || rows <= 0 || row0 + rows > H) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: rows [%d, %d) outside the %d-row grid", row0, row0 + rows, H);
    if (list_bytes % 16) return fail(OFL_E_INVALID, "ofl_scatter_slab_stars: list_bytes must be a multiple of 16");
    return exact_slab_stars(flow, point_precision ? 2 * sign : sign, pmask, H, W, row0, rows, list, list_bytes, workspace, workspace_bytes,
                            stream_of(stream));
}

int ofl_scatter_slab_finish_dev(const float *flow, int sign, int point_precision, const float *vals, int C, const uint8_t *vmask,
                                int H, int W, int row0, int rows, const uint32_t *lists, size_t list_bytes, int n_lists,
                                float *out_rows, uint8_t *valid_rows, int valid_rule,
                                void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream)
{
    OFL_TRY(need_device());
    if (!lists) return fail(OFL_E_INVALID, "ofl_scatter_slab_finish: NULL pointer");
    OFL_TRY(check_common("ofl_scatter_slab_finish", flow, sign, point_precision, C, vals, out_rows, valid_rows, valid_rule, H, W, workspace, workspace_bytes));
    if ((long long)H * W >= (1ll << 27)) return fail(OFL_E_INVALID, "ofl_scatter_slab_finish: H*W must be below 2^27");
    if (row0 < 0 || rows <= 0 || row0 + rows > H) return fail(OFL_E_INVALID, "ofl_scatter_slab_finish: rows [%d, %d) outside the %d-row grid", row0, row0 + rows, H);
    return exact_slab_finish(flow, sign, vals, C, vmask, H, W, row0, rows, lists, list_bytes, n_lists, out_rows, valid_rows,
                             valid_rule & ~OFL_SCATTER_UNCERTIFIED, workspace, workspace_bytes, info_host, stream_of(stream));
}

// Sparse queries (point tracking, utils.py:599-615): out[i][0..C) in float64, found[i] = 0 where griddata returns NaN.
int ofl_scatter_query_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                          const float *vals, int C, int H, int W,
                          const double *query_xy, size_t n_query, double *out, uint8_t *found,
                          void *workspace, size_t workspace_bytes, void *stream)
{
    OFL_TRY(need_device());
    if (C <= 0 || !vals || !out || !found || !query_xy) return fail(OFL_E_INVALID, "ofl_scatter_query: NULL pointer / C <= 0");
    // the counts are read back inside (the caller downloads the few results right away, so the synchronisation is free): a
    // point set without points, a refused one or an exceeded capacity is an ERROR here, not a list of "not found" flags
    uint64_t info[3];
    return scatter_query_impl("ofl_scatter_query", flow, sign, point_precision, pmask, vals, C, nullptr, H, W, query_xy,
                              n_query, true, out, found, 0, workspace, workspace_bytes, info, stream_of(stream));
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
