// K4: bilinear resize of a flow field and its mask (Flow.resize / resize_flow; reference
// src/oflibnumpy/utils.py:519-523 and flow_class.py:501-506, which call cv2.resize(.., None, fx, fy) with the
// default INTER_LINEAR).  One launch resamples the vectors, scales the two channels and rounds the mask.
// HBM-bound: every source line that is touched is read once (neighbouring output pixels share taps through
// the vector L1 / L2), every output pixel is written once with 16-byte stores.
#include "ofl_common.h"
#include <stdlib.h>

#pragma clang fp contract(off)

using namespace ofl;

namespace {

#ifndef OFL_RS_NT
#define OFL_RS_NT 1            // non-temporal output stores
#endif
typedef float v4f __attribute__((ext_vector_type(4)));

struct ResizeCoef { int s0, s1; float w0, w1; };

// opencv resize.cpp: fx = (float)((dx + 0.5) * scale - 0.5); sx = floor(fx); fx -= sx.  Columns collapse the tap
// pair onto the border sample, rows clip the indices and keep the weights.
__device__ __forceinline__ ResizeCoef resize_coef(int d, double scale, int n, bool clamp_weight)
{
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    ResizeCoef c;
    if (clamp_weight) {
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= n - 1) { s = n - 1; f = 0.f; }
        c.s0 = s;
        c.s1 = min(s + 1, n - 1);
    } else {
        c.s0 = min(max(s, 0), n - 1);
        c.s1 = min(max(s + 1, 0), n - 1);
    }
    c.w0 = 1.f - f;
    c.w1 = f;
    return c;
}

__device__ __forceinline__ float resize_blend(float v00, float v01, float v10, float v11, const ResizeCoef &cx, const ResizeCoef &cy)
{
    float r0 = v00 * cx.w0; r0 = r0 + v01 * cx.w1;
    float r1 = v10 * cx.w0; r1 = r1 + v11 * cx.w1;
    float v = r0 * cy.w0;
    return v + r1 * cy.w1;
}

__device__ __forceinline__ float2 resize_px(const float2 *__restrict__ src, const uint8_t *__restrict__ mask, int W,
                                            const ResizeCoef &cx, const ResizeCoef &cy, float mul_u, float mul_v, uint8_t &m)
{
    const size_t i00 = (size_t)cy.s0 * W + cx.s0, i10 = (size_t)cy.s1 * W + cx.s0;
    float2 a, b, c, d;
    uint32_t m0, m1;                                  // mask bytes of the two taps of each row
    if (cx.s1 == cx.s0 + 1) {
        // the two taps of a row are neighbours: ONE 16-byte load (8-byte aligned) and one 2-byte mask load per row
        const float4 r0 = *reinterpret_cast<const float4 *>(src + i00), r1 = *reinterpret_cast<const float4 *>(src + i10);
        a = make_float2(r0.x, r0.y); b = make_float2(r0.z, r0.w);
        c = make_float2(r1.x, r1.y); d = make_float2(r1.z, r1.w);
        m0 = mask ? *reinterpret_cast<const uint16_t *>(mask + i00) : 0u;
        m1 = mask ? *reinterpret_cast<const uint16_t *>(mask + i10) : 0u;
    } else {                                          // collapsed onto the border column
        a = b = src[i00];
        c = d = src[i10];
        m0 = mask ? mask[i00] * 0x101u : 0u;
        m1 = mask ? mask[i10] * 0x101u : 0u;
    }
    if (mask) {
        const float v = resize_blend((float)((m0 & 0xffu) != 0), (float)((m0 & 0xff00u) != 0), (float)((m1 & 0xffu) != 0),
                                     (float)((m1 & 0xff00u) != 0), cx, cy);
        m = (uint8_t)(v > 0.5f);                     // np.round (half to even) of a value in [0, 1]
    }
    return make_float2(resize_blend(a.x, b.x, c.x, d.x, cx, cy) * mul_u, resize_blend(a.y, b.y, c.y, d.y, cx, cy) * mul_v);
}

// block (64, 4): a wave covers 128 consecutive output pixels of one row (two per lane -> 1 KiB per store instruction)
__global__ __launch_bounds__(256)
void resize_flow_kernel(const float2 *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, int Ho, int Wo,
                        double scale_y, double scale_x, float mul_u, float mul_v,
                        float2 *__restrict__ out, uint8_t *__restrict__ mout)
{
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx = (blockIdx.x * 64 + threadIdx.x) * 2;
    if (dy >= Ho || dx >= Wo) return;
    const ResizeCoef cy = resize_coef(dy, scale_y, H, false);
    const ResizeCoef cx0 = resize_coef(dx, scale_x, W, true);
    const size_t o = (size_t)dy * Wo + dx;
    uint8_t m0 = 0, m1 = 0;
    const float2 p0 = resize_px(src, mask, W, cx0, cy, mul_u, mul_v, m0);
    if (dx + 1 < Wo) {
        const ResizeCoef cx1 = resize_coef(dx + 1, scale_x, W, true);
        const float2 p1 = resize_px(src, mask, W, cx1, cy, mul_u, mul_v, m1);
        if ((o & 1) == 0) {                          // 16-byte aligned pair (always when Wo is even)
#if OFL_RS_NT
            const v4f v = { p0.x, p0.y, p1.x, p1.y };
            __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(out) + (o >> 1));
            if (mout) __builtin_nontemporal_store((uint16_t)(m0 | (m1 << 8)), reinterpret_cast<uint16_t *>(mout + o));
#else
            reinterpret_cast<float4 *>(out)[o >> 1] = make_float4(p0.x, p0.y, p1.x, p1.y);
            if (mout) *reinterpret_cast<uint16_t *>(mout + o) = (uint16_t)(m0 | (m1 << 8));
#endif
        } else {
            out[o] = p0; out[o + 1] = p1;
            if (mout) { mout[o] = m0; mout[o + 1] = m1; }
        }
    } else {
        out[o] = p0;
        if (mout) mout[o] = m0;
    }
}

// Up-scaling (scale_x <= 1 source px per output px): four adjacent output pixels of a lane read at most SIX adjacent source
// columns, so the taps of all four come from two or three 16-byte loads (and one or two mask loads) per source row instead
// of one load pair per pixel -- 9 to 13 memory instructions per four pixels instead of 20; the values are picked from the
// loaded run by index.  Same arithmetic per pixel as resize_px: bit-identical output.
struct __attribute__((aligned(8))) Pair2f { float2 lo, hi; };
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) U16u { uint16_t v; };

__device__ __forceinline__ float2 pick6(int i, const float2 (&v)[6])
{
    float2 r = v[0];
    r = i == 1 ? v[1] : r; r = i == 2 ? v[2] : r; r = i == 3 ? v[3] : r; r = i == 4 ? v[4] : r; r = i == 5 ? v[5] : r;
    return r;
}

template <bool X2>      // X2: scale_x == 0.5 (the x2 up-scaling of pyramids): the tap columns of the four pixels sit at FIXED
__global__ __launch_bounds__(256)      // places of the run -- (0,1) (1,2) (1,2) (2,3) -- away from the left and right borders
void resize_flow4_kernel(const float2 *__restrict__ src, const uint8_t *__restrict__ mask, int H, int W, int Ho, int Wo,
                         double scale_y, double scale_x, float mul_u, float mul_v,
                         float2 *__restrict__ out, uint8_t *__restrict__ mout)
{
    const int dy = blockIdx.y * 4 + threadIdx.y;
    const int dx = (blockIdx.x * 64 + threadIdx.x) * 4;          // Wo % 4 == 0, W >= 6 (host checks)
    const bool act = dy < Ho && dx < Wo;
    const ResizeCoef cy = resize_coef(act ? dy : 0, scale_y, H, false);
    ResizeCoef cx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) cx[j] = resize_coef(act ? dx + j : j, scale_x, W, true);
    const int base = min(cx[0].s0, W - 6);                       // columns base .. base + 5 cover every tap of the four pixels
    const bool wide = __any(act && cx[3].s1 - base > 3);         // wave-uniform: is the third pair of columns needed at all?
    float2 r0[6], r1[6];
    uint32_t m0 = 0, m1 = 0;                                     // mask bytes of columns base .. base + 3 (+ 4, 5 in the high half of mh)
    uint32_t mh0 = 0, mh1 = 0;
    if (act) {
        const size_t i0 = (size_t)cy.s0 * W + base, i1 = (size_t)cy.s1 * W + base;
        const Pair2f a0 = *reinterpret_cast<const Pair2f *>(src + i0), a1 = *reinterpret_cast<const Pair2f *>(src + i0 + 2);
        const Pair2f b0 = *reinterpret_cast<const Pair2f *>(src + i1), b1 = *reinterpret_cast<const Pair2f *>(src + i1 + 2);
        r0[0] = a0.lo; r0[1] = a0.hi; r0[2] = a1.lo; r0[3] = a1.hi;
        r1[0] = b0.lo; r1[1] = b0.hi; r1[2] = b1.lo; r1[3] = b1.hi;
        r0[4] = r0[5] = r1[4] = r1[5] = make_float2(0.f, 0.f);
        if (wide) {
            const Pair2f a2 = *reinterpret_cast<const Pair2f *>(src + i0 + 4), b2 = *reinterpret_cast<const Pair2f *>(src + i1 + 4);
            r0[4] = a2.lo; r0[5] = a2.hi; r1[4] = b2.lo; r1[5] = b2.hi;
        }
        if (mask) {
            m0 = reinterpret_cast<const U32u *>(mask + i0)->v;
            m1 = reinterpret_cast<const U32u *>(mask + i1)->v;
            if (wide) {
                mh0 = reinterpret_cast<const U16u *>(mask + i0 + 4)->v;
                mh1 = reinterpret_cast<const U16u *>(mask + i1 + 4)->v;
            }
        }
    }
    float2 p[4];
    uint32_t mo = 0;
    // x2: the picks by index (a third of this kernel's instructions, and it is instruction-bound) are spared wherever the
    // whole wave has the regular pattern; the arithmetic per pixel is the same
    bool regular = false;
    if constexpr (X2)
        regular = __all(!act || (cx[0].s0 == base && cx[0].s1 == base + 1 && cx[1].s0 == base + 1 && cx[1].s1 == base + 2 &&
                                 cx[2].s0 == base + 1 && cx[2].s1 == base + 2 && cx[3].s0 == base + 2 && cx[3].s1 == base + 3));
    if (!act) return;
    if (X2 && regular) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            constexpr int kk[4] = { 0, 1, 1, 2 };
            const int k0 = kk[j];
            const float2 a = r0[k0], b = r0[k0 + 1], c = r1[k0], d = r1[k0 + 1];
            p[j] = make_float2(resize_blend(a.x, b.x, c.x, d.x, cx[j], cy) * mul_u, resize_blend(a.y, b.y, c.y, d.y, cx[j], cy) * mul_v);
            if (mask) {
                const float q00 = (float)(((m0 >> (8 * k0)) & 0xffu) != 0), q01 = (float)(((m0 >> (8 * k0 + 8)) & 0xffu) != 0);
                const float q10 = (float)(((m1 >> (8 * k0)) & 0xffu) != 0), q11 = (float)(((m1 >> (8 * k0 + 8)) & 0xffu) != 0);
                const float v = resize_blend(q00, q01, q10, q11, cx[j], cy);
                mo |= (v > 0.5f ? 1u : 0u) << (8 * j);
            }
        }
    } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k0 = cx[j].s0 - base, k1 = cx[j].s1 - base;
        const float2 a = pick6(k0, r0), b = pick6(k1, r0), c = pick6(k0, r1), d = pick6(k1, r1);
        p[j] = make_float2(resize_blend(a.x, b.x, c.x, d.x, cx[j], cy) * mul_u, resize_blend(a.y, b.y, c.y, d.y, cx[j], cy) * mul_v);
        if (mask) {
            const unsigned long long w0 = (unsigned long long)m0 | ((unsigned long long)mh0 << 32), w1 = (unsigned long long)m1 | ((unsigned long long)mh1 << 32);
            const float q00 = (float)(((w0 >> (8 * k0)) & 0xffu) != 0), q01 = (float)(((w0 >> (8 * k1)) & 0xffu) != 0);
            const float q10 = (float)(((w1 >> (8 * k0)) & 0xffu) != 0), q11 = (float)(((w1 >> (8 * k1)) & 0xffu) != 0);
            const float v = resize_blend(q00, q01, q10, q11, cx[j], cy);
            mo |= (v > 0.5f ? 1u : 0u) << (8 * j);
        }
    }
    }
    const size_t o = (size_t)dy * Wo + dx;
#if OFL_RS_NT
    const v4f v0 = { p[0].x, p[0].y, p[1].x, p[1].y }, v1 = { p[2].x, p[2].y, p[3].x, p[3].y };
    __builtin_nontemporal_store(v0, reinterpret_cast<v4f *>(out) + (o >> 1));
    __builtin_nontemporal_store(v1, reinterpret_cast<v4f *>(out) + (o >> 1) + 1);
    if (mout) __builtin_nontemporal_store(mo, reinterpret_cast<uint32_t *>(mout + o));
#else
    reinterpret_cast<float4 *>(out)[o >> 1] = make_float4(p[0].x, p[0].y, p[1].x, p[1].y);
    reinterpret_cast<float4 *>(out)[(o >> 1) + 1] = make_float4(p[2].x, p[2].y, p[3].x, p[3].y);
    if (mout) *reinterpret_cast<uint32_t *>(mout + o) = mo;
#endif
}

}  // namespace

extern "C" {

int ofl_resize_flow_dev(const float *vecs, const uint8_t *mask, int H, int W, int Ho, int Wo,
                        double scale_y, double scale_x, float mul_u, float mul_v,
                        float *out, uint8_t *mout, void *stream)
{
    OFL_TRY(need_device());
    if (!vecs || !out) return fail(OFL_E_INVALID, "ofl_resize_flow: NULL pointer");
    if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return fail(OFL_E_INVALID, "ofl_resize_flow: sizes must be positive");
    if (!(scale_y > 0.0) || !(scale_x > 0.0)) return fail(OFL_E_INVALID, "ofl_resize_flow: scales must be positive");
    if ((mask == nullptr) != (mout == nullptr)) return fail(OFL_E_INVALID, "ofl_resize_flow: mask and mout go together");
    const unsigned gy = (unsigned)((Ho + 3) / 4), gx = (unsigned)((Wo + 127) / 128);
    if (gy > 65535u) return fail(OFL_E_INVALID, "ofl_resize_flow: output too tall");
    static const bool no4 = OFL_KNOB_SET("OFL_RS_NO4");              // development knob (A/B; experiments build only)
    if (!no4 && scale_x <= 1.0 && (Wo & 3) == 0 && W >= 6) {
        if (scale_x == 0.5)
            hipLaunchKernelGGL(resize_flow4_kernel<true>, dim3((unsigned)((Wo + 255) / 256), gy), dim3(64, 4), 0, stream_of(stream),
                               reinterpret_cast<const float2 *>(vecs), mask, H, W, Ho, Wo, scale_y, scale_x, mul_u, mul_v,
                               reinterpret_cast<float2 *>(out), mout);
        else
            hipLaunchKernelGGL(resize_flow4_kernel<false>, dim3((unsigned)((Wo + 255) / 256), gy), dim3(64, 4), 0, stream_of(stream),
                               reinterpret_cast<const float2 *>(vecs), mask, H, W, Ho, Wo, scale_y, scale_x, mul_u, mul_v,
                               reinterpret_cast<float2 *>(out), mout);
        OFL_HIP(hipGetLastError());
        return OFL_OK;
    }
    hipLaunchKernelGGL(resize_flow_kernel, dim3(gx, gy), dim3(64, 4), 0, stream_of(stream),
                       reinterpret_cast<const float2 *>(vecs), mask, H, W, Ho, Wo, scale_y, scale_x, mul_u, mul_v,
                       reinterpret_cast<float2 *>(out), mout);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_resize_flow(const float *vecs, const uint8_t *mask, int H, int W, int Ho, int Wo,
                    double scale_y, double scale_x, float mul_u, float mul_v, float *out, uint8_t *mout)
{
    OFL_TRY(need_device());
    if (!vecs || !out) return fail(OFL_E_INVALID, "ofl_resize_flow: NULL pointer");
    if (H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return fail(OFL_E_INVALID, "ofl_resize_flow: sizes must be positive");
    const size_t n_in = (size_t)H * W, n_out = (size_t)Ho * Wo;
    hipStream_t s = rt().stream;
    void *dv = nullptr, *dm = nullptr, *dout = nullptr, *dmo = nullptr;
    int rc = OFL_OK;
    hipError_t e = hipSuccess;
    do {
        if ((e = hipMalloc(&dv, n_in * 8)) != hipSuccess) break;
        if ((e = hipMalloc(&dout, n_out * 8 + 16)) != hipSuccess) break;
        if (mask && (e = hipMalloc(&dm, n_in)) != hipSuccess) break;
        if (mask && (e = hipMalloc(&dmo, n_out + 16)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(dv, vecs, n_in * 8, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if (mask && (e = hipMemcpyAsync(dm, mask, n_in, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((rc = ofl_resize_flow_dev((const float *)dv, (const uint8_t *)dm, H, W, Ho, Wo, scale_y, scale_x, mul_u, mul_v,
                                      (float *)dout, (uint8_t *)dmo, s)) != OFL_OK) break;
        if ((e = hipMemcpyAsync(out, dout, n_out * 8, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if (mask && (e = hipMemcpyAsync(mout, dmo, n_out, hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    if (e != hipSuccess) rc = hip_fail(e, "ofl_resize_flow");
    if (dv) (void)hipFree(dv);
    if (dm) (void)hipFree(dm);
    if (dout) (void)hipFree(dout);
    if (dmo) (void)hipFree(dmo);
    return rc;
}

}  // extern "C"
