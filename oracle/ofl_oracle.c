/*
 * ofl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the oflibnumpy hot path that goes through cv2.remap
 * (the 't'-reference gather) plus the mode-3 composition built on it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (oflibnumpy_amd) never does.
 *
 * What it follows (file:line into /root/reference):
 *   - sample-map construction      src/oflibnumpy/utils.py:231-235
 *   - cv2.remap(INTER_LINEAR)      src/oflibnumpy/utils.py:236
 *   - mask channel / "== 1" rule   src/oflibnumpy/flow_class.py:632-644, 667-680
 *   - mode-3 closed forms          src/oflibnumpy/flow_class.py:1412-1422
 *   - zero-flow predicates         src/oflibnumpy/utils.py:298-316, 527-544;
 *                                  src/oflibnumpy/flow_class.py:1230-1245
 *
 * cv2.remap itself lives in a third-party dependency that is absent from
 * /root/reference and from this image: opencv-python (unpinned in
 * setup.py:52-56; docs/requirements.txt:2 pins 4.2.0.34).  Its published
 * algorithm (modules/imgproc/src/imgwarp.cpp: RemapInvoker + remapBilinear,
 * INTER_BITS = 5, INTER_TAB_SIZE = 32, BORDER_CONSTANT value 0) is restated
 * here:
 *     sx = cvRound(px * 32), sy = cvRound(py * 32)         (round-half-even)
 *     ix = sat_s16(sx >> 5), iy = sat_s16(sy >> 5), ax = sx & 31, ay = sy & 31
 *     w  = {(1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx}, fx = ax/32, fy = ay/32
 *     dst = v00*w0 + v01*w1 + v10*w2 + v11*w3   (left to right, no FMA),
 *     taps outside the source contribute 0.
 *   float / 16-bit sources accumulate in float, double sources in double,
 *   8-bit sources use the 15-bit fixed-point table ((sum + 2^14) >> 15).
 *
 * PARITY STATUS: pinned by the reference's own known-answer tests for this
 * path (tests/test_flow_class.py:852-980 7x7 masks, tests/test_utils.py:277-283
 * integer translation, tests/test_flow_class.py:1050-1057 analytic mode 3);
 * sub-1/32-px behaviour of cv2.remap is "parity unpinned" -- no reference test
 * discriminates it and OpenCV cannot be run here.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ORC_U8 = 0, ORC_I16 = 1, ORC_U16 = 2, ORC_F32 = 3, ORC_F64 = 4 };
enum { ORC_QUANT_OPENCV = 0, ORC_QUANT_EXACT = 1 };
enum { ORC_ARITH_NATIVE = 0, ORC_ARITH_FLOAT_RNE = 1 };
enum { ORC_RULE_EQ1 = 0, ORC_RULE_GE_HALF = 1, ORC_RULE_GT_HALF = 2 };

int orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* cvRound(float): round-half-even to int32; out of range -> INT_MIN like
 * cvtss2si.  (The saturated ix below makes both conventions sample nothing.) */
static inline int cv_round_f(float v)
{
    float r = nearbyintf(v);
    if (!(r >= -2147483648.0f && r < 2147483648.0f)) return INT_MIN;
    return (int)r;
}

static inline int sat_s16(int v)
{
    return v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
}

/* Sample position of one pixel: utils.py:231-235.  NumPy evaluates the in-place
 * "float32 += int64" in float64 and rounds once to float32. */
static inline float map_coord(int grid, float flow, int sign)
{
    double f = (double)flow;
    return (float)(sign >= 0 ? (double)grid + f : (double)grid - f);
}

typedef struct {
    int ix, iy;        /* top-left tap */
    float w[4];        /* float table weights */
    int wi[4];         /* 15-bit fixed-point weights (8-bit sources) */
} tap_t;

static inline void make_tap(float px, float py, int quant, tap_t *t)
{
    if (quant == ORC_QUANT_OPENCV) {
        int sx = cv_round_f(px * 32.0f);
        int sy = cv_round_f(py * 32.0f);
        int ax = sx & 31, ay = sy & 31;
        t->ix = sat_s16(sx >> 5);
        t->iy = sat_s16(sy >> 5);
        float fx = (float)ax * (1.0f / 32.0f), fy = (float)ay * (1.0f / 32.0f);
        float x0 = 1.0f - fx, y0 = 1.0f - fy;
        t->w[0] = y0 * x0; t->w[1] = y0 * fx; t->w[2] = fy * x0; t->w[3] = fy * fx;
        t->wi[0] = (32 - ay) * (32 - ax) * 32; t->wi[1] = (32 - ay) * ax * 32;
        t->wi[2] = ay * (32 - ax) * 32;        t->wi[3] = ay * ax * 32;
    } else {
        float flx = floorf(px), fly = floorf(py);
        float fx = px - flx, fy = py - fly;
        /* clamp to the int16 range the snapped path uses */
        flx = flx < -32768.0f ? -32768.0f : (flx > 32767.0f ? 32767.0f : flx);
        fly = fly < -32768.0f ? -32768.0f : (fly > 32767.0f ? 32767.0f : fly);
        t->ix = (int)flx; t->iy = (int)fly;
        float x0 = 1.0f - fx, y0 = 1.0f - fy;
        t->w[0] = y0 * x0; t->w[1] = y0 * fx; t->w[2] = fy * x0; t->w[3] = fy * fx;
        for (int k = 0; k < 4; ++k) t->wi[k] = (int)lrintf(t->w[k] * 32768.0f);
    }
}

static inline int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

static inline double load_as_double(const void *src, int dtype, size_t idx)
{
    switch (dtype) {
    case ORC_U8:  return (double)((const uint8_t *)src)[idx];
    case ORC_I16: return (double)((const int16_t *)src)[idx];
    case ORC_U16: return (double)((const uint16_t *)src)[idx];
    case ORC_F32: return (double)((const float *)src)[idx];
    default:      return ((const double *)src)[idx];
    }
}

/* One destination element.  v[] are the four taps (0 where out of bounds). */
static inline void blend_store(void *dst, int dtype, int arith, size_t idx,
                               const double v[4], const tap_t *t)
{
    if (dtype == ORC_U8 && arith == ORC_ARITH_NATIVE) {
        int acc = (int)v[0] * t->wi[0] + (int)v[1] * t->wi[1] +
                  (int)v[2] * t->wi[2] + (int)v[3] * t->wi[3];
        ((uint8_t *)dst)[idx] = (uint8_t)sat_u8((acc + (1 << 14)) >> 15);
        return;
    }
    if (dtype == ORC_F64) {
        double s = v[0] * (double)t->w[0];
        s = s + v[1] * (double)t->w[1];
        s = s + v[2] * (double)t->w[2];
        s = s + v[3] * (double)t->w[3];
        ((double *)dst)[idx] = s;
        return;
    }
    float s = (float)v[0] * t->w[0];
    s = s + (float)v[1] * t->w[1];
    s = s + (float)v[2] * t->w[2];
    s = s + (float)v[3] * t->w[3];
    switch (dtype) {
    case ORC_F32: ((float *)dst)[idx] = s; break;
    case ORC_U8: {  /* uint8 image riding in an int16 concat: cvRound, saturate_cast<short>, astype(uint8) */
        int r = cv_round_f(s);
        r = sat_s16(r);
        ((uint8_t *)dst)[idx] = (uint8_t)r;
        break; }
    case ORC_I16: ((int16_t *)dst)[idx] = (int16_t)sat_s16(cv_round_f(s)); break;
    case ORC_U16: { int r = cv_round_f(s); r = r < 0 ? 0 : (r > 65535 ? 65535 : r);
                    ((uint16_t *)dst)[idx] = (uint16_t)r; break; }
    default: break;
    }
}

/* Validity of one pixel = what "warped_mask_channel == 1" evaluates to
 * (flow_class.py:668) for the dtype the concatenated array had. */
static inline uint8_t valid_of(const uint8_t m[4], const tap_t *t, int rule)
{
    if (rule == ORC_RULE_EQ1) {
        float s = (float)m[0] * t->w[0];
        s = s + (float)m[1] * t->w[1];
        s = s + (float)m[2] * t->w[2];
        s = s + (float)m[3] * t->w[3];
        return s == 1.0f;
    }
    if (rule == ORC_RULE_GE_HALF) {   /* uint8 fixed point: (acc + 2^14) >> 15 == 1 */
        int acc = m[0] * t->wi[0] + m[1] * t->wi[1] + m[2] * t->wi[2] + m[3] * t->wi[3];
        return ((acc + (1 << 14)) >> 15) == 1;
    }
    /* int16 concat: cvRound(float sum) == 1 */
    float s = (float)m[0] * t->w[0];
    s = s + (float)m[1] * t->w[1];
    s = s + (float)m[2] * t->w[2];
    s = s + (float)m[3] * t->w[3];
    return cv_round_f(s) == 1;
}

/*
 * General gather: dst[y,x,:] = B(src; (x,y) + sign*flow[y,x]).
 *   src/dst  [H,W,C] of dtype; flow [fH,fW,2] f32 placed at (pad_top,pad_left)
 *   inside the H x W target (zero flow elsewhere: Flow.pad 'constant',
 *   flow_class.py:652-659); smask [H,W] u8 or NULL (= all ones); valid [H,W]
 *   u8 or NULL.  Returns 0.
 */
int orc_gather_bilinear(const void *src, int dtype, int C, int H, int W,
                        const float *flow, int fH, int fW, int pad_top, int pad_left,
                        int sign, const uint8_t *smask, void *dst, uint8_t *valid,
                        int quant, int arith, int rule)
{
    if (H <= 0 || W <= 0 || C <= 0) return 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            float fu = 0.0f, fv = 0.0f;
            int fy_ = y - pad_top, fx_ = x - pad_left;
            if (fy_ >= 0 && fy_ < fH && fx_ >= 0 && fx_ < fW) {
                fu = flow[((size_t)fy_ * fW + fx_) * 2 + 0];
                fv = flow[((size_t)fy_ * fW + fx_) * 2 + 1];
            }
            tap_t t;
            make_tap(map_coord(x, fu, sign), map_coord(y, fv, sign), quant, &t);
            int xs[2] = { t.ix, t.ix + 1 }, ys[2] = { t.iy, t.iy + 1 };
            int inb[4];
            size_t off[4];
            for (int k = 0; k < 4; ++k) {
                int yy = ys[k >> 1], xx = xs[k & 1];
                inb[k] = (xx >= 0 && xx < W && yy >= 0 && yy < H);
                off[k] = inb[k] ? ((size_t)yy * W + xx) : 0;
            }
            size_t o = (size_t)y * W + x;
            for (int c = 0; c < C; ++c) {
                double v[4];
                for (int k = 0; k < 4; ++k)
                    v[k] = inb[k] ? load_as_double(src, dtype, off[k] * C + c) : 0.0;
                blend_store(dst, dtype, arith, o * C + c, v, &t);
            }
            if (valid) {
                uint8_t m[4];
                for (int k = 0; k < 4; ++k)
                    m[k] = inb[k] ? (smask ? (smask[off[k]] != 0) : 1) : 0;
                valid[o] = valid_of(m, &t, rule);
            }
        }
    }
    return 0;
}

/*
 * Mode-3 composition (flow_class.py:1412-1422), no early exits:
 *   ref 't' (sign=-1): out = fb + B(fa; x - fb), mout = mb & [B(ma; x - fb) == 1]
 *   ref 's' (sign=+1): out = fb + B(fa; x + fb), mout = mb & [B(ma; x + fb) == 1]
 * fa/ma = the field that is SAMPLED, fb/mb = the field that supplies the sample
 * positions and the addend:  't': fa = self(f1), fb = flow(f2);
 *                            's': fa = flow(f2), fb = self(f1).
 */
int orc_compose3(const float *fa, const uint8_t *ma, const float *fb, const uint8_t *mb,
                 int sign, int H, int W, float *out, uint8_t *mout, int quant)
{
    if (H <= 0 || W <= 0) return 1;
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y) {
        for (int x = 0; x < W; ++x) {
            size_t o = (size_t)y * W + x;
            float bu = fb[o * 2], bv = fb[o * 2 + 1];
            tap_t t;
            make_tap(map_coord(x, bu, sign), map_coord(y, bv, sign), quant, &t);
            int xs[2] = { t.ix, t.ix + 1 }, ys[2] = { t.iy, t.iy + 1 };
            float su = 0.f, sv = 0.f;
            uint8_t m[4];
            float vu[4], vv[4];
            for (int k = 0; k < 4; ++k) {
                int yy = ys[k >> 1], xx = xs[k & 1];
                int in = (xx >= 0 && xx < W && yy >= 0 && yy < H);
                size_t s = in ? ((size_t)yy * W + xx) : 0;
                vu[k] = in ? fa[s * 2] : 0.f;
                vv[k] = in ? fa[s * 2 + 1] : 0.f;
                m[k] = in ? (ma[s] != 0) : 0;
            }
            su = vu[0] * t.w[0]; su = su + vu[1] * t.w[1]; su = su + vu[2] * t.w[2]; su = su + vu[3] * t.w[3];
            sv = vv[0] * t.w[0]; sv = sv + vv[1] * t.w[1]; sv = sv + vv[2] * t.w[2]; sv = sv + vv[3] * t.w[3];
            out[o * 2] = bu + su;
            out[o * 2 + 1] = bv + sv;
            mout[o] = (uint8_t)((mb[o] != 0) & valid_of(m, &t, ORC_RULE_EQ1));
        }
    }
    return 0;
}

/*
 * Zero-flow predicates.
 *   masked==0, thresholded==1: is_zero_flow(flow, True)  utils.py:527-544 with
 *       threshold_vectors utils.py:310-315 (per component, strict |v| < 1e-3).
 *   masked==1: Flow.is_zero  flow_class.py:1244 -- only vectors where mask.
 * Returns 1 if zero, 0 otherwise.
 */
int orc_is_zero(const float *flow, const uint8_t *mask, size_t n_px, int thresholded, double threshold)
{
    int zero = 1;
    /* NumPy compares the float32 array against the Python float in float32
     * (weak scalar), so the threshold is rounded to float32 first. */
    const float th = (float)threshold;
#pragma omp parallel for schedule(static) reduction(&& : zero)
    for (long long i = 0; i < (long long)n_px; ++i) {
        if (mask && !mask[i]) continue;
        for (int c = 0; c < 2; ++c) {
            float v = flow[i * 2 + c];
            int z = thresholded ? ((v < th) && (v > -th)) : (v == 0.0f);
            zero = zero && z;
        }
    }
    return zero;
}

/*
 * Flow.resize / resize_flow  (utils.py:519-523, flow_class.py:501-506): cv2.resize(.., None, fx, fy) with the
 * default INTER_LINEAR on the float32 vectors and on mask.astype('f'), per-channel vector scaling, np.round
 * of the mask.  OpenCV is not vendored by the reference (parity unpinned below float rounding); this follows
 * the published algorithm of opencv 4.2 modules/imgproc/src/resize.cpp (resize() -> resizeGeneric_ with
 * HResizeLinear / VResizeLinear for float): per output column
 *     fx = (float)((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx;
 *     sx < 0 -> (sx, fx) = (0, 0);  sx >= W-1 -> (sx, fx) = (W-1, 0)
 * rows alike but with the row INDICES clipped to [0, H-1] and the weights kept; horizontal pass
 * S[sx]*(1-fx) + S[sx+1]*fx first, then vertical R0*(1-fy) + R1*fy, all float32, no contraction.
 * (For fx = fy = 0.5 exactly OpenCV dispatches to its 2x2 area average, the same value up to the
 * summation order of four floats.)  scale_x, scale_y = 1/fx, 1/fy as doubles; Ho, Wo = cvRound(H*fy), ..
 * mul_u, mul_v: float32 factors applied to channel 0 / 1 (utils.py:521-522).
 * Mask: interpolated as float32 0/1, np.round half-to-even -> valid iff value > 0.5 (flow_class.py:504-506).
 */
static void orc_resize_coef(int d, double scale, int n, int clamp_weight, int *s0, int *s1, float *w0, float *w1)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (clamp_weight) {             /* columns: the tap pair collapses on the border sample */
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= n - 1) { s = n - 1; f = 0.f; }
        *s0 = s;
        *s1 = s + 1 < n ? s + 1 : n - 1;
    } else {                        /* rows: indices clipped, weights kept */
        *s0 = s < 0 ? 0 : (s > n - 1 ? n - 1 : s);
        *s1 = s + 1 < 0 ? 0 : (s + 1 > n - 1 ? n - 1 : s + 1);
    }
    *w0 = 1.f - f;
    *w1 = f;
}

int orc_resize_flow(const float *vecs, const uint8_t *mask, int H, int W, int Ho, int Wo,
                    double scale_y, double scale_x, float mul_u, float mul_v, float *out, uint8_t *mout)
{
    if (!vecs || !out || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return 1;
#pragma omp parallel for schedule(static)
    for (int dy = 0; dy < Ho; ++dy) {
        int y0, y1;
        float b0, b1;
        orc_resize_coef(dy, scale_y, H, 0, &y0, &y1, &b0, &b1);
        for (int dx = 0; dx < Wo; ++dx) {
            int x0, x1;
            float a0, a1;
            orc_resize_coef(dx, scale_x, W, 1, &x0, &x1, &a0, &a1);
            const size_t i00 = (size_t)y0 * W + x0, i01 = (size_t)y0 * W + x1;
            const size_t i10 = (size_t)y1 * W + x0, i11 = (size_t)y1 * W + x1;
            const size_t o = (size_t)dy * Wo + dx;
            for (int c = 0; c < 2; ++c) {
                float r0 = vecs[i00 * 2 + c] * a0; r0 = r0 + vecs[i01 * 2 + c] * a1;
                float r1 = vecs[i10 * 2 + c] * a0; r1 = r1 + vecs[i11 * 2 + c] * a1;
                float v = r0 * b0; v = v + r1 * b1;
                out[o * 2 + c] = v * (c == 0 ? mul_u : mul_v);
            }
            if (mask && mout) {
                float r0 = (float)(mask[i00] != 0) * a0; r0 = r0 + (float)(mask[i01] != 0) * a1;
                float r1 = (float)(mask[i10] != 0) * a0; r1 = r1 + (float)(mask[i11] != 0) * a1;
                float v = r0 * b0; v = v + r1 * b1;
                mout[o] = (uint8_t)(v > 0.5f);
            }
        }
    }
    return 0;
}
