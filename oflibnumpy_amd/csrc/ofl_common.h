// ofl_common.h -- shared host/device helpers of libofl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ofl.h"

// ----------------------------------------------------------------------------- host side state
namespace ofl {

struct Runtime {
    bool        ready   = false;
    int         device  = -1;
    hipStream_t stream  = nullptr;   // library default stream (non-blocking)
    int         n_cu    = 256;
};

Runtime &rt();
int  fail(int code, const char *fmt, ...);          // records the message, returns `code`
int  hip_fail(hipError_t e, const char *what);      // OFL_E_HIP (or OFL_E_NOMEM) with the HIP text
int  need_device();                                  // OFL_OK or OFL_E_NODEVICE

inline hipStream_t stream_of(void *s) { return s ? (hipStream_t)s : rt().stream; }

}  // namespace ofl

#define OFL_HIP(call)                                                      \
    do {                                                                   \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) return ofl::hip_fail(e__, #call);           \
    } while (0)

// Tuning / test knobs read from the environment exist only in the EXPERIMENTS build (-DOFL_EXPERIMENTS, linked as
// libofl_hip_exp.so by build_native.build_experiments() for tools/ and for the tests of the non-default routes);
// the shipped library evaluates them to their defaults at compile time and never calls getenv.
#ifdef OFL_EXPERIMENTS
#include <stdlib.h>
#define OFL_KNOB_INT(name, dflt)    (getenv(name) ? atoi(getenv(name)) : (dflt))
#define OFL_KNOB_DOUBLE(name, dflt) (getenv(name) ? atof(getenv(name)) : (dflt))
#define OFL_KNOB_SET(name)          (getenv(name) != nullptr)
#else
#define OFL_KNOB_INT(name, dflt)    (dflt)
#define OFL_KNOB_DOUBLE(name, dflt) (dflt)
#define OFL_KNOB_SET(name)          (false)
#endif

#define OFL_TRY(call)                                                      \
    do {                                                                   \
        int rc__ = (call);                                                 \
        if (rc__ != OFL_OK) return rc__;                                   \
    } while (0)

// ----------------------------------------------------------------------------- device side helpers
// All float arithmetic that must agree bit-for-bit with the CPU oracle goes through the
// explicitly rounded intrinsics below so that no FMA contraction can change a result.
namespace ofl {

struct Tap {
    int   ix, iy;     // top-left tap (already saturated to the int16 range cv2.remap uses)
    float w0, w1, w2, w3;
    int   ax, ay;     // 1/32-px fractions (opencv quantisation only)
};

// Sample coordinate of one pixel, utils.py:231-235: NumPy evaluates "float32 += int64" in float64
// and rounds once to float32.
__device__ __forceinline__ float map_coord(int grid, float flow, int sign)
{
    double f = (double)flow;
    return (float)(sign >= 0 ? (double)grid + f : (double)grid - f);
}

__device__ __forceinline__ int sat_s16(int v) { return min(max(v, -32768), 32767); }

// cvRound(float): round-half-even.  v_cvt_i32_f32 saturates where cvtss2si yields INT_MIN; both
// ends land outside every admissible image (dims <= 32766) after the int16 saturation.
__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }

template <int QUANT>
__device__ __forceinline__ Tap make_tap(float px, float py)
{
    Tap t;
    if (QUANT == OFL_QUANT_OPENCV) {
        int sx = cv_round(__fmul_rn(px, 32.0f));
        int sy = cv_round(__fmul_rn(py, 32.0f));
        t.ax = sx & 31;
        t.ay = sy & 31;
        t.ix = sat_s16(sx >> 5);
        t.iy = sat_s16(sy >> 5);
        float fx = (float)t.ax * (1.0f / 32.0f), fy = (float)t.ay * (1.0f / 32.0f);
        float x0 = 1.0f - fx, y0 = 1.0f - fy;
        t.w0 = __fmul_rn(y0, x0); t.w1 = __fmul_rn(y0, fx);
        t.w2 = __fmul_rn(fy, x0); t.w3 = __fmul_rn(fy, fx);
    } else {
        float flx = floorf(px), fly = floorf(py);
        float fx = __fsub_rn(px, flx), fy = __fsub_rn(py, fly);
        flx = fminf(fmaxf(flx, -32768.0f), 32767.0f);
        fly = fminf(fmaxf(fly, -32768.0f), 32767.0f);
        t.ix = (int)flx; t.iy = (int)fly;
        t.ax = t.ay = 0;
        float x0 = __fsub_rn(1.0f, fx), y0 = __fsub_rn(1.0f, fy);
        t.w0 = __fmul_rn(y0, x0); t.w1 = __fmul_rn(y0, fx);
        t.w2 = __fmul_rn(fy, x0); t.w3 = __fmul_rn(fy, fx);
    }
    return t;
}

// v00*w0 + v01*w1 + v10*w2 + v11*w3, left to right, one rounding per operation (remapBilinear).
__device__ __forceinline__ float blend4(float v00, float v01, float v10, float v11, const Tap &t)
{
    float s = __fmul_rn(v00, t.w0);
    s = __fadd_rn(s, __fmul_rn(v01, t.w1));
    s = __fadd_rn(s, __fmul_rn(v10, t.w2));
    s = __fadd_rn(s, __fmul_rn(v11, t.w3));
    return s;
}

__device__ __forceinline__ double blend4d(double v00, double v01, double v10, double v11, const Tap &t)
{
    double s = __dmul_rn(v00, (double)t.w0);
    s = __dadd_rn(s, __dmul_rn(v01, (double)t.w1));
    s = __dadd_rn(s, __dmul_rn(v10, (double)t.w2));
    s = __dadd_rn(s, __dmul_rn(v11, (double)t.w3));
    return s;
}

// zero-flow statistic bits of one vector (OFL_STAT_*), th = float32(1e-3): utils.py:315 compares
// the float32 array with the weak Python scalar in float32.
__device__ __forceinline__ uint32_t stat_bits(float u, float v, bool m, float th)
{
    bool nz  = (u != 0.0f) | (v != 0.0f);
    bool nzt = !((u < th) & (u > -th) & (v < th) & (v > -th));
    return ((nz && m) ? 1u : 0u) | ((nzt && m) ? 2u : 0u) | (nz ? 4u : 0u) | (nzt ? 8u : 0u);
}

}  // namespace ofl
