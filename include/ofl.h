/*
 * ofl.h -- C ABI of the MI355X-native optical-flow-field engine (libofl_hip.so).
 *
 * The reference (oflibnumpy 1.1.1, pure Python) has no FFI; its hot path funnels through ONE
 * Python seam, `apply_flow(flow, target, ref, mask)` (src/oflibnumpy/utils.py:199-261), plus the
 * expressions built on it in src/oflibnumpy/flow_class.py.  Every entry point below names the
 * reference lines it replaces.  A maintainer binds these with ctypes.
 *
 * Conventions
 *   - plain C types only: pointers, sizes, ints.  No torch / numpy types.
 *   - every function returns OFL_OK (0) or a negative OFL_E* code; ofl_last_error() gives the text.
 *   - *_dev entry points take DEVICE pointers (from ofl_malloc) and a stream handle (NULL = the
 *     library's default stream); they enqueue work and return without synchronising.
 *   - host entry points (no suffix) take caller-owned, C-contiguous HOST buffers, do
 *     H2D -> kernel -> D2H on the default stream and return when the result is in host memory.
 *     Nothing is retained after return.
 *   - layouts: flow vecs float32 [H][W][2] (channel 0 = x / horizontal, 1 = y / vertical),
 *     masks uint8 [H][W] with values 0/1, images [H][W][C] C-contiguous.
 *   - one process drives one GPU (ofl_init(device) once per process); the library is re-entrant
 *     across streams of that device.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry fails with
 *     OFL_E_NODEVICE.
 */
#ifndef OFL_H
#define OFL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFL_ABI_VERSION 4

/* status codes */
enum {
    OFL_OK          = 0,
    OFL_E_INVALID   = -1,   /* bad argument (shape, dtype, NULL pointer, unsupported combination) */
    OFL_E_NODEVICE  = -2,   /* no HIP device / ofl_init not called */
    OFL_E_HIP       = -3,   /* a HIP runtime call failed (text in ofl_last_error) */
    OFL_E_NOMEM     = -4,
    OFL_E_NOPOINTS  = -5,   /* scatter: no valid source points (qhull "No points given" in the reference) */
    OFL_E_RCCL      = -6
};

/* element types of warp targets (what cv2.remap accepts for INTER_LINEAR) */
enum { OFL_U8 = 0, OFL_I16 = 1, OFL_U16 = 2, OFL_F32 = 3, OFL_F64 = 4 };

/* sample-position quantisation of the bilinear gather */
enum {
    OFL_QUANT_OPENCV = 0,   /* cv2.remap semantics: coordinates snapped to 1/32 px (INTER_BITS = 5) */
    OFL_QUANT_EXACT  = 1    /* floor / fractional part in float32, no snapping (extension) */
};

/* arithmetic of the blend for 8-bit sources */
enum {
    OFL_ARITH_NATIVE    = 0,  /* u8: 15-bit fixed point, (acc + 2^14) >> 15; others float/double */
    OFL_ARITH_FLOAT_RNE = 1   /* u8 riding in an int16 concat (flow_class.py:615,644): float sum, round-half-even */
};

/* how "warped_mask == 1" (flow_class.py:668) evaluates for the dtype the reference's concat had */
enum {
    OFL_RULE_EQ1     = 0,   /* float concat: interpolated mask == 1 exactly */
    OFL_RULE_GE_HALF = 1,   /* uint8 concat (fixed point): interpolated >= 0.5 */
    OFL_RULE_GT_HALF = 2    /* int16 concat (round-half-even): interpolated > 0.5 */
};

/* flags OR-ed into the `valid_rule` argument of the scatter entries */
enum {
    OFL_SCATTER_ROUND  = 0x100,  /* out = rint(result): np.round of integer-typed targets, utils.py:256-257 */
    OFL_SCATTER_NEGATE = 0x200,  /* out = -result: the values are -vals (Flow.invert s->s = self.apply(-self), flow_class.py:746,
                                    without materialising -self; negation commutes exactly with the interpolation) */
    OFL_SCATTER_UNCERTIFIED = 0x400  /* the caller KNOWS the mesh cannot be certified (a certificate from ofl_scatter_certify_dev
                                    that says so, or a point mask with zeros): the entry skips its own certificate pass and
                                    read-back and takes the Delaunay path at once.  Never changes a result -- that path is the
                                    general one -- only saves the pass */
};

/* bits written by the zero-flow statistics (ofl_flow_stats_dev, and the fused compose kernel) */
enum {
    OFL_STAT_NONZERO_MASKED     = 1,  /* some vector component != 0 where mask   (Flow.is_zero(thresholded=False), flow_class.py:1244) */
    OFL_STAT_NONZERO_TH_MASKED  = 2,  /* some |component| >= 1e-3 where mask     (Flow.is_zero(thresholded=True)) */
    OFL_STAT_NONZERO            = 4,  /* some component != 0 anywhere            (is_zero_flow(thresholded=False), utils.py:527) */
    OFL_STAT_NONZERO_TH         = 8,  /* some |component| >= 1e-3 anywhere       (is_zero_flow(thresholded=True), utils.py:215) */
    OFL_STAT_NONFINITE          = 16, /* NaN / Inf present                       (validate_flow_array, utils.py:86) */
    OFL_STAT_MASK_HAS_ZERO      = 32  /* some mask byte is 0 (ofl_flow_stats only; lets callers pass NULL instead of an all-ones point mask) */
};

/* ------------------------------------------------------------------ runtime / device memory */
int         ofl_abi_version(void);
const char *ofl_last_error(void);
int         ofl_device_count(int *count);
int         ofl_init(int device);                 /* select device, create default stream */
int         ofl_device_name(char *buf, size_t buflen);
int         ofl_malloc(void **dptr, size_t bytes);
int         ofl_free(void *dptr);
int         ofl_memset(void *dptr, int value, size_t bytes, void *stream);
int         ofl_upload(void *dptr, const void *host, size_t bytes, void *stream);     /* async on stream; pinned staging not required */
int         ofl_download(void *host, const void *dptr, size_t bytes, void *stream);
int         ofl_download_async(void *host, const void *dptr, size_t bytes, void *stream);  /* no synchronisation: host must be pinned (ofl_host_alloc) for a true DMA */
int         ofl_copy_dev(void *dst, const void *src, size_t bytes, void *stream);
/* page-locked host memory: uploads / downloads from it are asynchronous DMA transfers that overlap with kernels
 * (on-disk formats are read straight into it: oflibnumpy_amd.device.load_sintel_device; utils.py:447-470) */
int         ofl_host_alloc(void **hptr, size_t bytes);
int         ofl_host_free(void *hptr);
int         ofl_stream_create(void **stream);
int         ofl_stream_destroy(void *stream);
int         ofl_stream_sync(void *stream);        /* NULL = default stream */
int         ofl_device_sync(void);
int         ofl_event_create(void **event);
int         ofl_event_destroy(void *event);
int         ofl_event_record(void *event, void *stream);
int         ofl_event_sync(void *event);
int         ofl_event_elapsed_ms(void *start, void *stop, float *ms);
int         ofl_mem_info(size_t *free_bytes, size_t *total_bytes);
/* host helper of the dataset loaders (load_kitti, load_sintel_mask; utils.py:426-490): PNG row un-filtering, types 0-4;
 * raw = inflated IDAT stream (height rows of 1 filter byte + stride bytes), out [height][stride]; no device needed */
int         ofl_png_unfilter(const uint8_t *raw, size_t raw_bytes, int height, int stride, int bpp, uint8_t *out);

/* ------------------------------------------------------------------ K2: fused mode-3 composition
 * Replaces Flow.combine_with(mode=3) numerics, flow_class.py:1412-1422 (+ Flow.apply :632-684,
 * apply_flow 't' utils.py:231-236, Flow.__add__ :332-334):
 *     out  = fb + B(fa; x + sign*fb)                 B = cv2.remap bilinear, 0 outside
 *     mout = mb & [B(ma; x + sign*fb) == 1]
 *   ref 't': fa/ma = self (f1), fb/mb = flow (f2), sign = -1
 *   ref 's': fa/ma = flow (f2), fb/mb = self (f1), sign = +1
 * batch fields are stored back to back ([batch][H][W][..]).  `stats` (device, uint32[batch][8] or
 * NULL) receives zero-flow flag WORDS computed from data the launch reads anyway (no extra traffic):
 * a word is set to 1 when its condition holds and is never cleared, so the caller zeroes the words
 * beforehand (plain idempotent stores -- no atomics on the hot path).
 *     stats[b][4 + k], k = 0..3 (the OFL_STAT_* bit index): EXACT predicates of fb/mb.
 *     stats[b][0], stats[b][1]: CERTIFICATES for fa/ma -- set when a gathered, masked vector of fa is
 *         non-zero / at or above the 1e-3 threshold, i.e. fa is certainly not (thresholded-)zero.  A clear
 *         word means "not observed": the caller confirms with ofl_flow_stats_dev before taking one of the
 *         reference's early exits (flow_class.py:1339-1354).  stats[b][2..3] are not written.
 * The host entry returns exact OFL_STAT_* bit masks: stats_host[2*b] for fa, stats_host[2*b+1] for fb.
 * out / mout must not alias the inputs.
 */
int ofl_compose3_dev(const float *fa, const uint8_t *ma, const float *fb, const uint8_t *mb,
                     int sign, int H, int W, int batch, float *out, uint8_t *mout,
                     uint32_t *stats, int quant, void *stream);
int ofl_compose3(const float *fa, const uint8_t *ma, const float *fb, const uint8_t *mb,
                 int sign, int H, int W, int batch, float *out, uint8_t *mout,
                 uint32_t *stats_host, int quant);

/* K2 with PACKED masks, for chains that stay on the device (SURVEY hard part H5: "masks as bytes, or bit-packed internally").
 * The three mask planes are one BIT per pixel -- bit (x & 31) of the uint32 word (x >> 5) of a row, rows padded to whole
 * words: [batch][H][(W + 31) / 32] words, allocated with ofl_mask_bits_bytes (16 bytes of slack after the last row) -- which
 * takes the masks from 3 of the 27 B/px of the contract to 0.4 (and from up to five times that, on a rotated sampling grid, in
 * partially used cache lines).  Same arithmetic, same flag words as ofl_compose3_dev (OFL_QUANT_OPENCV); the result equals
 * ofl_compose3_dev's bit for bit once unpacked.  W must be even.  ofl_mask_pack_dev / ofl_mask_unpack_dev convert between the
 * uint8 masks of every other entry (and of the host API) and the planes.
 */
int ofl_mask_bits_bytes(int H, int W, int batch, size_t *bytes);
int ofl_mask_pack_dev(const uint8_t *mask, int H, int W, int batch, uint32_t *bits, void *stream);
int ofl_mask_unpack_dev(const uint32_t *bits, int H, int W, int batch, uint8_t *mask, void *stream);
int ofl_compose3_bits_dev(const float *fa, const uint32_t *ma_bits, const float *fb, const uint32_t *mb_bits,
                          int sign, int H, int W, int batch, float *out, uint32_t *mout_bits,
                          uint32_t *stats, void *stream);

/* ------------------------------------------------------------------ K1: general bilinear gather
 * Replaces apply_flow(flow, target, 't') utils.py:231-236 and the mask handling around it in
 * Flow.apply flow_class.py:632-695, valid_target :1148-1150, valid_source :1179-1183:
 *     dst[y][x][c] = B(src[..][c]; (x, y) + sign * flow[y - pad_top][x - pad_left])
 *     valid[y][x]  = rule(B(smask; ...)) [& fmask[y - pad_top][x - pad_left], 0 outside the flow area]
 *   src/dst [H][W][C] of `dtype`; flow [fH][fW][2] located at (pad_top, pad_left) inside the
 *   H x W target with ZERO flow elsewhere (Flow.pad mode 'constant', flow_class.py:652-659);
 *   smask [H][W] or NULL (all ones); valid [H][W] or NULL; fmask [fH][fW] or NULL.
 *   C == 0 with src = dst = NULL computes `valid` only (warp of an all-ones / smask image).
 */
int ofl_gather_bilinear_dev(const void *src, int dtype, int C, int H, int W,
                            const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                            const uint8_t *smask, const uint8_t *fmask,
                            void *dst, uint8_t *valid,
                            int quant, int arith, int rule, void *stream);
/* K1 over a BATCH of fields in one launch (Flow.apply for a stack of independent warps: BASELINE config 2-sized jobs are one
 * generation of waves per launch when launched alone, utils.py:231-236 / flow_class.py:632-695 per field).  Field b of the
 * batch reads flow [b][fH][fW][2], fmask [b][fH][fW] (or NULL), writes dst [b][H][W][C] and valid [b][H][W] (or NULL);
 * src is [b][H][W][C] -- or ONE image [H][W][C] warped by every flow when src_shared != 0 -- and smask likewise
 * ([b][H][W], one shared [H][W] when smask_shared != 0, or NULL).  Everything else as ofl_gather_bilinear_dev, whose results
 * the fields equal bit for bit.  batch in [1, 65535].
 */
int ofl_gather_bilinear_batch_dev(const void *src, int src_shared, int dtype, int C, int H, int W, int batch,
                                  const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                                  const uint8_t *smask, int smask_shared, const uint8_t *fmask,
                                  void *dst, uint8_t *valid, int quant, int arith, int rule, void *stream);

/* One row band of the same result, for a field split over several GPUs: the source
 * image (and smask) are replicated, each rank holds rows [row0, row0 + rows) of the flow / its mask and
 * produces the same rows of dst / valid:
 *     dst_rows[y][x][c] = B(src[..][c]; (x, row0 + y) + sign * flow_rows[y][x]),  0 <= y < rows
 * flow_rows [rows][W][2], fmask_rows [rows][W] or NULL, dst_rows [rows][W][C], valid_rows [rows][W] or NULL.
 * Bands are disjoint, so there is no exchange step after the launch.
 */
int ofl_gather_rows_dev(const void *src, int dtype, int C, int H, int W, int row0, int rows,
                        const float *flow_rows, int sign, const uint8_t *smask, const uint8_t *fmask_rows,
                        void *dst_rows, uint8_t *valid_rows, int quant, int arith, int rule, void *stream);
int ofl_gather_bilinear(const void *src, int dtype, int C, int H, int W,
                        const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                        const uint8_t *smask, const uint8_t *fmask,
                        void *dst, uint8_t *valid,
                        int quant, int arith, int rule);

/* ------------------------------------------------------------------ K4: zero-flow / finite statistics
 * Replaces is_zero_flow utils.py:527-544, threshold_vectors :298-316, Flow.is_zero
 * flow_class.py:1230-1245 and the finite check of validate_flow_array utils.py:86.
 * stats: one uint32 (device for _dev, host otherwise), OR of OFL_STAT_* bits; zeroed by the callee.
 */
int ofl_flow_stats_dev(const float *flow, const uint8_t *mask, size_t n_px, float threshold,
                       uint32_t *stats, void *stream);
int ofl_flow_stats(const float *flow, const uint8_t *mask, size_t n_px, float threshold,
                   uint32_t *stats_host);

/* ------------------------------------------------------------------ K5: element-wise epilogues
 * Flow.__add__/__sub__/__neg__ flow_class.py:310-375, 479-489:  out = a + alpha*b, mout = ma & mb.
 * b/mb may be NULL (then out = alpha * a, mout = ma: negation / scaling).
 */
int ofl_axpy_dev(const float *a, const uint8_t *ma, const float *b, const uint8_t *mb, float alpha,
                 size_t n_px, float *out, uint8_t *mout, void *stream);

/* ------------------------------------------------------------------ K3: scattered -> regular grid
 * Replaces apply_flow(flow, target, 's', mask) utils.py:237-258 (scipy.interpolate.griddata 'linear',
 * NaN -> 0) and the inline griddata of mode 2 / ref 't' flow_class.py:1398-1410.
 * Source point i sits at (x, y) + sign*flow[i] -- evaluated in float64 (point_precision 0, utils.py:242)
 * or rounded to float32 first (point_precision 1, flow_class.py:1398-1400) -- and carries
 * vals[i][0..C) (float32) plus, optionally, a mask value vmask[i]; points with pmask[i] == 0 are
 * dropped (pmask NULL = keep all, utils.py:249-251), and so are points whose position is not finite (the reference's Flow
 * refuses NaN / Inf vectors before it gets here).  The result is linear on the Delaunay triangulation of the kept
 * points (what griddata builds with Qhull):
 *     out[H][W][C]  piecewise-linear interpolation (float64 barycentric, stored as float32),
 *                   0 where no triangle covers the node;
 *     valid[H][W]   valid_rule 0: float32(interpolated vmask) == 1   (flow_class.py:668)
 *                   valid_rule 1: interpolated vmask > 0.99           (flow_class.py:1410)
 *                   valid_rule 2: rint(interpolated vmask) == 1 -- integer-typed targets, whose concatenated mask channel
 *                                 the reference rounds before the comparison (utils.py:256-257, flow_class.py:668)
 *                   valid_rule | OFL_SCATTER_ROUND: additionally out = float32(rint(float64 result)), the
 *                   np.round the reference applies to integer-typed targets (utils.py:256-257)
 *                   (vmask NULL = all ones, i.e. valid == "covered by a triangle").
 * query == NULL evaluates at the regular grid nodes; otherwise query [H][W][2] holds absolute (x, y)
 * positions (mode 2 't').  C may be 0 (validity only).  `workspace` (device) must hold
 * ofl_scatter_workspace_bytes() bytes.  info_host (host uint64[3] or NULL): [0] kept points (exact duplicates of a
 * site included, although only the smallest index of a location is a site of the triangulation); on the Delaunay path
 * [1] points whose star neither the mesh-fan pass nor the per-thread ring search could finish and [2] those of them left
 * for the workgroup pass (hull points, fan apexes of border pockets); 0 / 0 on the certified path.
 * OFL_E_NOPOINTS when no point is kept (qhull's "No points given") -- on the Delaunay path this and the capacity errors are
 * known only after the fact and reported to callers that pass info_host (one read-back at the end of the call); with
 * info_host == NULL that path only enqueues work and a field without kept points gives an all-invalid result.
 * Grid nodes take one of two paths: a field whose cell-wise mesh is certified to BE the Delaunay triangulation
 * (ofl_scatter_certify_dev) is resolved by one kernel; every other field -- folds, dropped points, curved borders,
 * sheared cells -- gets a real Delaunay triangulation of the kept points on the GPU.
 * valid_rule | OFL_SCATTER_UNCERTIFIED skips the entry's own certificate pass (and its read-back) for callers that know
 * the answer is "not certified"; the result is the same either way.
 * A certificate says that the mesh is the triangulation, not that the one-kernel path finds every node's triangle in it
 * (a certified field may squeeze a third of the image into a sliver): that kernel counts the nodes it could not locate,
 * the entry -- which has synchronised for the certificate anyway -- reads the count back and, if it is not zero, computes
 * the call on the Delaunay path instead.  The results are SciPy's either way.
 * Degenerate point sets: exact duplicates are ONE site (Qhull's Qc; a flow may collapse a whole image block onto one pixel:
 * buckets of thousands of coincident points are reduced through a hash table of positions).  What remains degenerate is
 * refused like Qhull refuses a flat initial simplex (the reference raises QhullError): all points on one spot or one
 * axis-parallel line, more than 65 536 distinct sites crowded into buckets of more than 4 096, more than 2^20 unfinished
 * or 2^18 unbounded cells (bulk collinearity) -- OFL_E_INVALID for callers that pass info_host, an all-zero / all-invalid
 * result for everybody (also after any capacity error: never a partial warp).
 */
int ofl_scatter_linear_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                           const float *vals, int C, const uint8_t *vmask, int H, int W,
                           const float *query, float *out, uint8_t *valid, int valid_rule,
                           void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream);
/* The same for float64 values at the grid nodes (no query positions): griddata interpolates in float64 and
 * apply_flow returns result.astype(target.dtype) (utils.py:253-258), so a float64 image keeps its precision.
 */
int ofl_scatter_linear_f64_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                               const double *vals, int C, const uint8_t *vmask, int H, int W,
                               double *out, uint8_t *valid, int valid_rule,
                               void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream);

/* One row band of the grid result (SURVEY 8e, config 5 as loaded, ref 's'): every rank holds the full inputs
 * (flow, masks, values are replicated), rasterises only the triangles that reach rows
 * [row0 - 16, row0 + rows + 16) and resolves rows [row0, row0 + rows): out_rows [rows][W][C],
 * valid_rows [rows][W].  The concatenation of the bands equals ofl_scatter_linear_dev bit for bit; bands are
 * disjoint, nothing is exchanged afterwards.  Same workspace size as the full call.
 */
int ofl_scatter_rows_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                         const float *vals, int C, const uint8_t *vmask, int H, int W, int row0, int rows,
                         float *out_rows, uint8_t *valid_rows, int valid_rule,
                         void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream);
int ofl_scatter_workspace_bytes(int H, int W, int C, size_t *bytes);

/* The same band with the STAR passes sharded too (SURVEY 8e, config 5 'ref s' -- a field the certificate refuses, whose
 * Delaunay stars are most of the time): two calls per rank around ONE exchange.
 *   1. ofl_scatter_slab_stars_dev: bins all sites (replicated inputs, as above), builds the stars of the sites within a
 *      margin of the band's rows only, and writes the sites it could not finish -- rims of holes, hull sites: those of
 *      the rank's own rows, the first / last band open-ended -- to `list` (device memory, list_bytes a multiple of 16):
 *      uint32 { entries, error bits, 0, 0 } followed by one 64-byte record per site.  64 bytes per expected site; a
 *      list that overflows is reported by step 2 on every rank (OFL_E_INVALID, err flag 32) -- retry with a larger one
 *      or fall back to ofl_scatter_rows_dev.
 *   2. all-gather the lists (ofl_comm_allgather: `list_bytes` per rank, in rank order or any other).
 *   3. ofl_scatter_slab_finish_dev with the gathered buffer (`n_lists` lists of `list_bytes` each, this rank's among
 *      them) and THE SAME workspace, untouched since step 1 (checked: step 1 stamps its band into the workspace, a
 *      workspace without that stamp -- no step 1, another band's, any other scatter call in between, a second step 2 --
 *      is refused with OFL_E_INVALID before any kernel runs): every rank finishes all unfinished stars (they can reach
 *      any band), rasterises and resolves its rows.  out_rows / valid_rows / valid_rule / info_host as above; the call
 *      synchronises (it reads the error bits back: an error on one rank blanks every band).
 * The concatenation of the bands equals ofl_scatter_linear_dev with OFL_SCATTER_UNCERTIFIED bit for bit.  A field whose
 * mesh certifies needs none of this: ofl_scatter_certified_dev shards by rows without any exchange.
 */
int ofl_scatter_slab_stars_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask, int H, int W,
                               int row0, int rows, uint32_t *list, size_t list_bytes,
                               void *workspace, size_t workspace_bytes, void *stream);
int ofl_scatter_slab_finish_dev(const float *flow, int sign, int point_precision, const float *vals, int C,
                                const uint8_t *vmask, int H, int W, int row0, int rows,
                                const uint32_t *lists, size_t list_bytes, int n_lists,
                                float *out_rows, uint8_t *valid_rows, int valid_rule,
                                void *workspace, size_t workspace_bytes, uint64_t *info_host, void *stream);

/* Certificate of the warped grid as a triangulation (one pass over the flow, one 144-byte read-back; synchronises).
 * SciPy's griddata (utils.py:253) triangulates the points with Qhull; the cell-wise mesh of the warped grid IS that
 * Delaunay triangulation when no triangle is folded, every interior edge passes the local Delaunay test, no point is
 * dropped and the mesh border is straight between the warped image corners (then the convex hull is the border's).
 * `certified` = 1 states exactly that; such a field takes ofl_scatter_certified_dev -- one kernel, no owner map, no
 * atomics, no synchronisation.  The scatter entries above certify internally on every call; callers that warp with
 * the same field repeatedly certify once and keep the record (it depends on flow, sign, point_precision, pmask only).
 */
typedef struct ofl_mesh_cert {
    uint32_t certified;        /* 1: cell-wise mesh == Delaunay triangulation of the warped points (up to co-circular cells) */
    uint32_t folded_cells;     /* cells whose two triangles are not both positively oriented (counted exactly up to ~65 000, a lower bound beyond) */
    uint32_t bad_edges;        /* interior mesh edges failing the local Delaunay (in-circle) test beyond rounding (likewise) */
    uint32_t dropped;          /* 1: pmask drops points */
    double   border_dev;       /* px: largest distance of a border point from the straight side between its corners */
    double   corner[4][2];     /* warped image corners (x, y): (0,0), (W-1,0), (W-1,H-1), (0,H-1) */
    const uint32_t *diag_bits; /* the diag_bits buffer given to ofl_scatter_certify_dev (device memory the CALLER owns and keeps
                                  alive as long as the certificate is used), or NULL */
} ofl_mesh_cert;
/* diag_bits (device, ofl_scatter_diag_bytes(H, W) bytes, or NULL): receives the Delaunay diagonal of every grid cell, one bit
 * per cell, rows padded to 32-bit words.  ofl_scatter_certified_dev then reads a bit where it would otherwise evaluate the
 * cell's float64 in-circle determinant for every node -- same predicate, same numbers, same results, fewer instructions. */
int ofl_scatter_diag_bytes(int H, int W, size_t *bytes);
int ofl_scatter_certify_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask, int H, int W,
                            void *workspace, size_t workspace_bytes, ofl_mesh_cert *cert_host, uint32_t *diag_bits, void *stream);
/* Rows [row0, row0 + rows) of the grid result for a CERTIFIED field (cert->certified must be 1, no point mask):
 * out_rows [rows][W][C], valid_rows [rows][W] as in ofl_scatter_linear_dev; asynchronous.  fail_count_dev (device
 * uint32, may be NULL; the caller zeroes it) counts nodes well inside the hull for which the kernel found no triangle
 * (written as 0 / invalid).  It is 0 for every affine field; a caller that keeps a certificate should check it once per
 * (field, sign) -- which nodes are found depends on nothing else -- and send a field that loses nodes through
 * ofl_scatter_linear_dev with OFL_SCATTER_UNCERTIFIED (the Python layer does exactly that). */
int ofl_scatter_certified_dev(const float *flow, int sign, int point_precision, const float *vals, int C,
                              const uint8_t *vmask, int H, int W, int row0, int rows, float *out_rows,
                              uint8_t *valid_rows, int valid_rule, const ofl_mesh_cert *cert,
                              uint32_t *fail_count_dev, void *stream);
int ofl_scatter_linear(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                       const float *vals, int C, const uint8_t *vmask, int H, int W,
                       const float *query, float *out, uint8_t *valid, int valid_rule);

/* ------------------------------------------------------------------ sparse point tracking (next tier)
 * track_pts, utils.py:547-622:
 *   ofl_sample_points_dev   'ref s' default: bilinear_interpolation(flow[..., ::-1], pts) utils.py:161-196, float64,
 *                           pts_rc / out_rc are [n][2] in (row, col) order; points must lie inside the field
 *   ofl_scatter_query_dev   'ref t' (utils.py:610-615) and s_exact_mode (:599-603): griddata(points, values, pts):
 *                           same triangulation as ofl_scatter_linear_dev, evaluated at n_query float64 points
 *                           query_xy [n][2] = (x, y); out float64 [n][C]; found[i] = 0 where griddata gives NaN.
 *                           Synchronises: the point counts and error flags of the triangulation are read back inside, so
 *                           "no point kept" (OFL_E_NOPOINTS) and a refused / overflowing point set (OFL_E_INVALID) are
 *                           ERRORS here, never a list of found = 0 that looks like "outside the hull".
 */
int ofl_sample_points_dev(const float *flow, int H, int W, const double *pts_rc, size_t n, double *out_rc, void *stream);
int ofl_scatter_query_dev(const float *flow, int sign, int point_precision, const uint8_t *pmask,
                          const float *vals, int C, int H, int W,
                          const double *query_xy, size_t n_query, double *out, uint8_t *found,
                          void *workspace, size_t workspace_bytes, void *stream);

/* small device helpers of the flow algebra:
 *   ofl_mask_and_dev     out = a & b                      (flow_class.py:643)
 *   ofl_grid_offset_dev  out[y][x] = float32((x, y) + sign * vecs[y][x])   (flow_class.py:1398-1406)
 */
int ofl_mask_and_dev(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, void *stream);
/*   ofl_convert_dev      element-wise dtype conversion on the device, n elements: any OFL_* dtype -> OFL_F32 (the
 *                        values the scatter kernel interpolates, utils.py:253) and OFL_F32 -> any dtype with a plain
 *                        C cast (result.astype(target.dtype), utils.py:258; integer destinations expect integral,
 *                        in-range values -- the scatter entries with OFL_SCATTER_ROUND produce them)
 */
int ofl_convert_dev(const void *src, int src_dtype, void *dst, int dst_dtype, size_t n, void *stream);
/*   ofl_flow_extent_dev  extent (device float32[4]) = { min y, max y, min x, max x } of the positions
 *                        float32((x, y) + sign * threshold_vectors(vecs)[y][x]) over the masked pixels
 *                        (Flow.get_padding, flow_class.py:1214-1226: sign -1 for ref 't', +1 for 's');
 *                        { +inf, -inf, +inf, -inf } when no pixel is masked
 */
int ofl_flow_extent_dev(const float *vecs, const uint8_t *mask, int H, int W, int sign, float threshold,
                        float *extent, void *stream);
int ofl_grid_offset_dev(const float *vecs, int sign, int H, int W, float *out, void *stream);

/* ------------------------------------------------------------------ K4: bilinear resize of a flow field
 * Replaces resize_flow (src/oflibnumpy/utils.py:493-525: cv2.resize(flow, None, fx, fy), INTER_LINEAR, then
 * vecs[..., 0] *= fx, vecs[..., 1] *= fy) and the mask half of Flow.resize (flow_class.py:501-506:
 * np.round(cv2.resize(mask.astype('f'), ..))).  The caller supplies the output size Ho = cvRound(H * fy),
 * Wo = cvRound(W * fx), the inverse scales scale_y = 1 / fy, scale_x = 1 / fx (doubles, as OpenCV derives
 * them) and the float32 channel factors mul_u = float32(fx), mul_v = float32(fy).
 * vecs float32[H][W][2] -> out float32[Ho][Wo][2];  mask uint8[H][W] -> mout uint8[Ho][Wo] (both or neither).
 */
int ofl_resize_flow(const float *vecs, const uint8_t *mask, int H, int W, int Ho, int Wo,
                    double scale_y, double scale_x, float mul_u, float mul_v, float *out, uint8_t *mout);
int ofl_resize_flow_dev(const float *vecs, const uint8_t *mask, int H, int W, int Ho, int Wo,
                        double scale_y, double scale_x, float mul_u, float mul_v,
                        float *out, uint8_t *mout, void *stream);

/* ------------------------------------------------------------------ C1: the exchange steps (RCCL)
 * Two exchange steps exist in the sharded workload: one broadcast of a shared source image / flow from rank `root`
 * to all ranks over xGMI, and -- for one huge field warped with ref 's' in slab mode (above) -- one all-gather of the
 * ranks' lists of unfinished sites (`bytes` from every rank into recv[rank * bytes ...], send may alias its own slot).
 * The 128-byte unique id is created on rank 0 with ofl_comm_unique_id and distributed by the launcher
 * (torch.distributed store / gloo).
 */
int ofl_comm_unique_id(void *id128);
int ofl_comm_init(const void *id128, int rank, int world);
int ofl_comm_broadcast(void *dptr, size_t bytes, int root, void *stream);
int ofl_comm_allgather(const void *send, void *recv, size_t bytes, void *stream);
int ofl_comm_size(int *world);              /* ranks of the live communicator (ncclCommCount), 0 without one */
int ofl_comm_destroy(void);

#ifdef __cplusplus
}
#endif
#endif /* OFL_H */
