// ofl_gather.hip -- K1 general bilinear gather and K2 fused mode-3 composition for gfx950.
//
// Both kernels are HBM-bound byte movers (no MFMA): every output pixel streams its flow vector
// (coalesced, 16 B per lane), derives the cv2.remap sample position, and gathers the 2x2 source
// neighbourhood.  The two horizontally adjacent taps of one source row are fetched with ONE
// 8-byte-aligned 16-byte load; reuse between neighbouring pixels is served by the per-CU L1 and the
// per-XCD L2, which is why workgroups own compact 2-D tiles (64 x 16 px) and consecutive tiles of
// one XCD are neighbours (blocks b and b+8 share an XCD on MI355X).
//
// Numerics follow oracle/ofl_oracle.c operation for operation (contraction disabled) so that the
// float results and the validity masks are bit-identical to the CPU restatement.
#include "ofl_common.h"

#pragma clang fp contract(off)

using namespace ofl;

namespace {

constexpr int kTileW = 64;   // pixels per tile row  (16 lanes x 4 px)
constexpr int kTileH = 16;   // tile rows            (256 threads / 16 lanes)
constexpr int kPx    = 4;    // pixels per thread along x

struct __attribute__((aligned(8))) Pair2 { float lo_u, lo_v, hi_u, hi_v; };   // taps (ix, ix+1) of a float2 field

// MI355X deals consecutive workgroups round-robin over its 8 XCDs.  Give every XCD a contiguous
// run of tiles so that neighbouring tiles (which share gather halos) meet in the same L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks)
{
    constexpr int kXcd = 8;
    int per = nblocks / kXcd;
    int main = per * kXcd;
    if (bid >= main) return bid;            // ragged tail keeps its natural position
    return (bid % kXcd) * per + bid / kXcd;
}

// Tap selection for a row pair loaded at the clamped column ixc: d = ix - ixc is 0 in the interior,
// -1 when the left tap hangs over the left edge, +1 when the right tap hangs over the right edge;
// anything else means both taps are outside.
__device__ __forceinline__ void select_pair(const Pair2 &p, int d, bool row_ok,
                                            float &u0, float &v0, float &u1, float &v1)
{
    bool lo0 = row_ok & (d == 0), hi0 = row_ok & (d == 1);
    bool hi1 = row_ok & (d == 0), lo1 = row_ok & (d == -1);
    u0 = lo0 ? p.lo_u : (hi0 ? p.hi_u : 0.0f);
    v0 = lo0 ? p.lo_v : (hi0 ? p.hi_v : 0.0f);
    u1 = hi1 ? p.hi_u : (lo1 ? p.lo_u : 0.0f);
    v1 = hi1 ? p.hi_v : (lo1 ? p.lo_v : 0.0f);
}

__device__ __forceinline__ void select_mask(uint32_t lo, uint32_t hi, int d, bool row_ok, float &m0, float &m1)
{
    bool lo0 = row_ok & (d == 0), hi0 = row_ok & (d == 1);
    bool hi1 = row_ok & (d == 0), lo1 = row_ok & (d == -1);
    m0 = (lo0 ? (lo != 0) : (hi0 ? (hi != 0) : false)) ? 1.0f : 0.0f;
    m1 = (hi1 ? (hi != 0) : (lo1 ? (lo != 0) : false)) ? 1.0f : 0.0f;
}

// ------------------------------------------------------------------------------------ K2
template <int QUANT, bool STATS>
__global__ __launch_bounds__(256)
void compose3_kernel(const float *__restrict__ fa, const uint8_t *__restrict__ ma,
                     const float *__restrict__ fb, const uint8_t *__restrict__ mb,
                     int sign, int H, int W, int tiles_x, int tiles_per_field, int nblocks,
                     float *__restrict__ out, uint8_t *__restrict__ mout,
                     uint32_t *__restrict__ stats, float th)
{
    const int tile = xcd_swizzle(blockIdx.x, nblocks);
    const int b    = tile / tiles_per_field;
    const int t    = tile - b * tiles_per_field;
    const int ty   = t / tiles_x, tx = t - ty * tiles_x;
    const int lx   = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x0   = tx * kTileW + lx * kPx;
    const int y    = ty * kTileH + ly;

    const size_t field = (size_t)b * H * W;
    fa += field * 2; fb += field * 2; out += field * 2;
    ma += field;     mb += field;     mout += field;

    uint32_t bits_a = 0, bits_b = 0;
    if (y < H && x0 < W) {
        const size_t o = (size_t)y * W + x0;
        const float4   b01 = *reinterpret_cast<const float4 *>(fb + 2 * o);
        const float4   b23 = *reinterpret_cast<const float4 *>(fb + 2 * o + 4);
        const uint32_t mb4 = *reinterpret_cast<const uint32_t *>(mb + o);
        const float bu[kPx] = { b01.x, b01.z, b23.x, b23.z };
        const float bv[kPx] = { b01.y, b01.w, b23.y, b23.w };

        float    ou[kPx], ov[kPx];
        uint32_t mo = 0;
#pragma unroll
        for (int j = 0; j < kPx; ++j) {
            const Tap tp = make_tap<QUANT>(map_coord(x0 + j, bu[j], sign), map_coord(y, bv[j], sign));
            const int  ixc = min(max(tp.ix, 0), W - 2);
            const int  d   = tp.ix - ixc;
            const bool r0  = (unsigned)tp.iy < (unsigned)H;
            const bool r1  = (unsigned)(tp.iy + 1) < (unsigned)H;
            const int  y0c = min(max(tp.iy, 0), H - 1);
            const int  y1c = min(max(tp.iy + 1, 0), H - 1);
            const size_t s0 = (size_t)y0c * W + ixc, s1 = (size_t)y1c * W + ixc;

            const Pair2 p0 = *reinterpret_cast<const Pair2 *>(fa + 2 * s0);
            const Pair2 p1 = *reinterpret_cast<const Pair2 *>(fa + 2 * s1);
            const uint32_t m00 = ma[s0], m01 = ma[s0 + 1], m10 = ma[s1], m11 = ma[s1 + 1];

            float u00, v00, u01, v01, u10, v10, u11, v11, a00, a01, a10, a11;
            select_pair(p0, d, r0, u00, v00, u01, v01);
            select_pair(p1, d, r1, u10, v10, u11, v11);
            select_mask(m00, m01, d, r0, a00, a01);
            select_mask(m10, m11, d, r1, a10, a11);

            ou[j] = __fadd_rn(bu[j], blend4(u00, u01, u10, u11, tp));     // Flow.__add__, flow_class.py:332
            ov[j] = __fadd_rn(bv[j], blend4(v00, v01, v10, v11, tp));
            const bool mbit = ((mb4 >> (8 * j)) & 0xffu) != 0;
            const bool ok   = (blend4(a00, a01, a10, a11, tp) == 1.0f) & mbit;   // flow_class.py:668,680,333
            mo |= (ok ? 1u : 0u) << (8 * j);
            if (STATS) bits_b |= stat_bits(bu[j], bv[j], mbit, th);
        }
        *reinterpret_cast<float4 *>(out + 2 * o)     = make_float4(ou[0], ov[0], ou[1], ov[1]);
        *reinterpret_cast<float4 *>(out + 2 * o + 4) = make_float4(ou[2], ov[2], ou[3], ov[3]);
        *reinterpret_cast<uint32_t *>(mout + o)      = mo;

        if (STATS) {   // stream the sampled field once for its own early-exit predicate
            const float4   a01 = *reinterpret_cast<const float4 *>(fa + 2 * o);
            const float4   a23 = *reinterpret_cast<const float4 *>(fa + 2 * o + 4);
            const uint32_t ma4 = *reinterpret_cast<const uint32_t *>(ma + o);
            bits_a |= stat_bits(a01.x, a01.y, (ma4 & 0xffu) != 0, th);
            bits_a |= stat_bits(a01.z, a01.w, (ma4 & 0xff00u) != 0, th);
            bits_a |= stat_bits(a23.x, a23.y, (ma4 & 0xff0000u) != 0, th);
            bits_a |= stat_bits(a23.z, a23.w, (ma4 & 0xff000000u) != 0, th);
        }
    }
    if (STATS) {
        // wave-level OR, then idempotent plain stores of the flag words (no atomics: thousands of
        // waves hitting one address with atomics would serialise at the memory side).
        uint32_t wa = 0, wb = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wa |= (__ballot((bits_a >> k) & 1u) != 0ull) ? (1u << k) : 0u;
            wb |= (__ballot((bits_b >> k) & 1u) != 0ull) ? (1u << k) : 0u;
        }
        if ((threadIdx.x & 63) == 0) {
            uint32_t *s = stats + (size_t)b * 8;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (((wa >> k) & 1u) && __hip_atomic_load(s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) s[k] = 1u;
                if (((wb >> k) & 1u) && __hip_atomic_load(s + 4 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) s[4 + k] = 1u;
            }
        }
    }
}

// Generic-shape fallback (any W >= 1, one pixel per thread, no vector accesses).
template <int QUANT>
__global__ __launch_bounds__(256)
void compose3_generic_kernel(const float *__restrict__ fa, const uint8_t *__restrict__ ma,
                             const float *__restrict__ fb, const uint8_t *__restrict__ mb,
                             int sign, int H, int W, size_t n_total,
                             float *__restrict__ out, uint8_t *__restrict__ mout,
                             uint32_t *__restrict__ stats, float th)
{
    const size_t hw = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / hw, o = i - b * hw;
        const int y = (int)(o / W), x = (int)(o - (size_t)y * W);
        const float *fab = fa + b * hw * 2;
        const uint8_t *mab = ma + b * hw;
        const float bu = fb[2 * i], bv = fb[2 * i + 1];
        const Tap tp = make_tap<QUANT>(map_coord(x, bu, sign), map_coord(y, bv, sign));
        float u[4], v[4], a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = tp.iy + (k >> 1), xx = tp.ix + (k & 1);
            const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            const size_t s = in ? (size_t)yy * W + xx : 0;
            u[k] = in ? fab[2 * s] : 0.0f;
            v[k] = in ? fab[2 * s + 1] : 0.0f;
            a[k] = (in && mab[s] != 0) ? 1.0f : 0.0f;
        }
        out[2 * i]     = __fadd_rn(bu, blend4(u[0], u[1], u[2], u[3], tp));
        out[2 * i + 1] = __fadd_rn(bv, blend4(v[0], v[1], v[2], v[3], tp));
        const bool mbit = mb[i] != 0;
        mout[i] = (uint8_t)((blend4(a[0], a[1], a[2], a[3], tp) == 1.0f) & mbit);
        if (stats) {
            uint32_t sb = stat_bits(bu, bv, mbit, th);
            uint32_t sa = stat_bits(fa[2 * i], fa[2 * i + 1], ma[i] != 0, th);
            uint32_t *s = stats + b * 8;
            for (int k = 0; k < 4; ++k) {
                if ((sa >> k) & 1u) s[k] = 1u;
                if ((sb >> k) & 1u) s[4 + k] = 1u;
            }
        }
    }
}

// ------------------------------------------------------------------------------------ K1
template <typename T> struct Acc { typedef float type; };
template <> struct Acc<double> { typedef double type; };

template <typename T> __device__ __forceinline__ T finish(float s, int arith);
template <> __device__ __forceinline__ float    finish<float>(float s, int) { return s; }
template <> __device__ __forceinline__ int16_t  finish<int16_t>(float s, int) { return (int16_t)sat_s16(cv_round(s)); }
template <> __device__ __forceinline__ uint16_t finish<uint16_t>(float s, int) { return (uint16_t)min(max(cv_round(s), 0), 65535); }
template <> __device__ __forceinline__ uint8_t  finish<uint8_t>(float s, int) { return (uint8_t)sat_s16(cv_round(s)); }

template <typename T, int CT>
__global__ __launch_bounds__(256)
void gather_kernel(const T *__restrict__ src, int Crt, int H, int W,
                   const float *__restrict__ flow, int fH, int fW, int pad_top, int pad_left, int sign,
                   const uint8_t *__restrict__ smask, const uint8_t *__restrict__ fmask,
                   T *__restrict__ dst, uint8_t *__restrict__ valid,
                   int quant, int arith, int rule, int tiles_x, int nblocks)
{
    const int C    = CT > 0 ? CT : Crt;
    const int tile = xcd_swizzle(blockIdx.x, nblocks);
    const int ty   = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x    = tx * 32 + (threadIdx.x & 31);
    const int y    = ty * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;

    float fu = 0.0f, fv = 0.0f;
    const int  fy = y - pad_top, fx = x - pad_left;
    const bool in_flow = (unsigned)fy < (unsigned)fH && (unsigned)fx < (unsigned)fW;
    if (in_flow) {
        const float2 f = *reinterpret_cast<const float2 *>(flow + ((size_t)fy * fW + fx) * 2);
        fu = f.x; fv = f.y;
    }
    const float px = map_coord(x, fu, sign), py = map_coord(y, fv, sign);
    const Tap tp = (quant == OFL_QUANT_OPENCV) ? make_tap<OFL_QUANT_OPENCV>(px, py) : make_tap<OFL_QUANT_EXACT>(px, py);

    int wi[4];
    if (quant == OFL_QUANT_OPENCV) {
        wi[0] = (32 - tp.ay) * (32 - tp.ax) * 32; wi[1] = (32 - tp.ay) * tp.ax * 32;
        wi[2] = tp.ay * (32 - tp.ax) * 32;        wi[3] = tp.ay * tp.ax * 32;
    } else {
        wi[0] = __float2int_rn(tp.w0 * 32768.0f); wi[1] = __float2int_rn(tp.w1 * 32768.0f);
        wi[2] = __float2int_rn(tp.w2 * 32768.0f); wi[3] = __float2int_rn(tp.w3 * 32768.0f);
    }

    bool   in[4];
    size_t off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = tp.iy + (k >> 1), xx = tp.ix + (k & 1);
        in[k]  = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
        off[k] = in[k] ? (size_t)yy * W + xx : 0;
    }
    const size_t o = (size_t)y * W + x;
    const bool fixed_u8 = (sizeof(T) == 1) && arith == OFL_ARITH_NATIVE;

    auto do_channel = [&](int cc) {
        typedef typename Acc<T>::type A;
        A v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = in[k] ? (A)src[off[k] * C + cc] : (A)0;
        if constexpr (sizeof(T) == 8) {
            dst[o * C + cc] = (T)blend4d(v[0], v[1], v[2], v[3], tp);
        } else {
            if (fixed_u8) {
                const int acc = (int)v[0] * wi[0] + (int)v[1] * wi[1] + (int)v[2] * wi[2] + (int)v[3] * wi[3];
                dst[o * C + cc] = (T)min(max((acc + (1 << 14)) >> 15, 0), 255);
            } else {
                dst[o * C + cc] = finish<T>(blend4((float)v[0], (float)v[1], (float)v[2], (float)v[3], tp), arith);
            }
        }
    };
    if constexpr (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) do_channel(c);
    } else {
        for (int c = 0; c < C; ++c) do_channel(c);
    }

    if (valid) {
        int m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = in[k] ? (smask ? (smask[off[k]] != 0) : 1) : 0;
        bool ok;
        if (rule == OFL_RULE_GE_HALF) {
            const int acc = m[0] * wi[0] + m[1] * wi[1] + m[2] * wi[2] + m[3] * wi[3];
            ok = ((acc + (1 << 14)) >> 15) == 1;
        } else {
            const float s = blend4((float)m[0], (float)m[1], (float)m[2], (float)m[3], tp);
            ok = (rule == OFL_RULE_EQ1) ? (s == 1.0f) : (cv_round(s) == 1);
        }
        if (fmask) ok = ok & in_flow & (in_flow ? fmask[(size_t)fy * fW + fx] != 0 : false);
        valid[o] = ok ? 1 : 0;
    }
}

template <typename T>
int launch_gather_t(const void *src, int C, int H, int W, const float *flow, int fH, int fW, int pad_top,
                    int pad_left, int sign, const uint8_t *smask, const uint8_t *fmask, void *dst,
                    uint8_t *valid, int quant, int arith, int rule, hipStream_t s)
{
    const int tiles_x = (W + 31) / 32, tiles_y = (H + 7) / 8;
    const int nblocks = tiles_x * tiles_y;
#define OFL_GATHER_LAUNCH(CT)                                                                          \
    hipLaunchKernelGGL((gather_kernel<T, CT>), dim3(nblocks), dim3(256), 0, s, (const T *)src, C, H, W, \
                       flow, fH, fW, pad_top, pad_left, sign, smask, fmask, (T *)dst, valid, quant,     \
                       arith, rule, tiles_x, nblocks)
    switch (C) {
    case 1: OFL_GATHER_LAUNCH(1); break;
    case 2: OFL_GATHER_LAUNCH(2); break;
    case 3: OFL_GATHER_LAUNCH(3); break;
    case 4: OFL_GATHER_LAUNCH(4); break;
    default: OFL_GATHER_LAUNCH(0); break;
    }
#undef OFL_GATHER_LAUNCH
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

size_t dtype_size(int dtype)
{
    switch (dtype) {
    case OFL_U8: return 1;
    case OFL_I16: case OFL_U16: return 2;
    case OFL_F32: return 4;
    case OFL_F64: return 8;
    default: return 0;
    }
}

int check_dims(const char *who, int H, int W)
{
    // cv2.remap asserts src/dst dims < SHRT_MAX (coordinates are int16 inside OpenCV)
    if (H <= 0 || W <= 0 || H > 32766 || W > 32766)
        return fail(OFL_E_INVALID, "%s: H, W must be in [1, 32766] (got %d x %d)", who, H, W);
    return OFL_OK;
}

// scoped device buffer for the host-pointer entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes)
    {
        hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
        return e == hipSuccess ? OFL_OK : hip_fail(e, "hipMalloc");
    }
};

}  // namespace

extern "C" {

int ofl_compose3_dev(const float *fa, const uint8_t *ma, const float *fb, const uint8_t *mb,
                     int sign, int H, int W, int batch, float *out, uint8_t *mout,
                     uint32_t *stats, int quant, void *stream)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_compose3", H, W));
    if (!fa || !ma || !fb || !mb || !out || !mout) return fail(OFL_E_INVALID, "ofl_compose3: NULL pointer");
    if (batch <= 0) return fail(OFL_E_INVALID, "ofl_compose3: batch must be >= 1");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_compose3: sign must be +1 or -1");
    if (quant != OFL_QUANT_OPENCV && quant != OFL_QUANT_EXACT) return fail(OFL_E_INVALID, "ofl_compose3: bad quant");
    hipStream_t s = stream_of(stream);
    const float th = 1e-3f;   // DEFAULT_THRESHOLD, utils.py:22 (compared in float32)

    if (W % kPx == 0 && W >= 2) {
        const int tiles_x = (W + kTileW - 1) / kTileW, tiles_y = (H + kTileH - 1) / kTileH;
        const long long nb = (long long)tiles_x * tiles_y * batch;
        if (nb > 0x7fffffffLL) return fail(OFL_E_INVALID, "ofl_compose3: too many tiles");
        const int nblocks = (int)nb, tpf = tiles_x * tiles_y;
#define OFL_C3(Q, S)                                                                                    \
        hipLaunchKernelGGL((compose3_kernel<Q, S>), dim3(nblocks), dim3(256), 0, s, fa, ma, fb, mb, sign, \
                           H, W, tiles_x, tpf, nblocks, out, mout, stats, th)
        if (quant == OFL_QUANT_OPENCV) { if (stats) OFL_C3(OFL_QUANT_OPENCV, true); else OFL_C3(OFL_QUANT_OPENCV, false); }
        else                           { if (stats) OFL_C3(OFL_QUANT_EXACT, true);  else OFL_C3(OFL_QUANT_EXACT, false); }
#undef OFL_C3
    } else {
        const size_t n = (size_t)batch * H * W;
        const int nblocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        if (quant == OFL_QUANT_OPENCV)
            hipLaunchKernelGGL((compose3_generic_kernel<OFL_QUANT_OPENCV>), dim3(nblocks), dim3(256), 0, s,
                               fa, ma, fb, mb, sign, H, W, n, out, mout, stats, th);
        else
            hipLaunchKernelGGL((compose3_generic_kernel<OFL_QUANT_EXACT>), dim3(nblocks), dim3(256), 0, s,
                               fa, ma, fb, mb, sign, H, W, n, out, mout, stats, th);
    }
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_compose3(const float *fa, const uint8_t *ma, const float *fb, const uint8_t *mb,
                 int sign, int H, int W, int batch, float *out, uint8_t *mout,
                 uint32_t *stats_host, int quant)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_compose3", H, W));
    if (batch <= 0) return fail(OFL_E_INVALID, "ofl_compose3: batch must be >= 1");
    if (!fa || !ma || !fb || !mb || !out || !mout) return fail(OFL_E_INVALID, "ofl_compose3: NULL pointer");
    const size_t n = (size_t)batch * H * W;
    hipStream_t s = rt().stream;
    DevBuf dfa, dma, dfb, dmb, dout, dmout, dst;
    OFL_TRY(dfa.alloc(n * 8)); OFL_TRY(dma.alloc(n)); OFL_TRY(dfb.alloc(n * 8)); OFL_TRY(dmb.alloc(n));
    OFL_TRY(dout.alloc(n * 8)); OFL_TRY(dmout.alloc(n)); OFL_TRY(dst.alloc((size_t)batch * 8 * 4));
    OFL_HIP(hipMemcpyAsync(dfa.p, fa, n * 8, hipMemcpyHostToDevice, s));
    OFL_HIP(hipMemcpyAsync(dma.p, ma, n, hipMemcpyHostToDevice, s));
    OFL_HIP(hipMemcpyAsync(dfb.p, fb, n * 8, hipMemcpyHostToDevice, s));
    OFL_HIP(hipMemcpyAsync(dmb.p, mb, n, hipMemcpyHostToDevice, s));
    OFL_HIP(hipMemsetAsync(dst.p, 0, (size_t)batch * 8 * 4, s));
    OFL_TRY(ofl_compose3_dev((const float *)dfa.p, (const uint8_t *)dma.p, (const float *)dfb.p,
                             (const uint8_t *)dmb.p, sign, H, W, batch, (float *)dout.p, (uint8_t *)dmout.p,
                             stats_host ? (uint32_t *)dst.p : nullptr, quant, s));
    OFL_HIP(hipMemcpyAsync(out, dout.p, n * 8, hipMemcpyDeviceToHost, s));
    OFL_HIP(hipMemcpyAsync(mout, dmout.p, n, hipMemcpyDeviceToHost, s));
    uint32_t words[8];
    for (int b = 0; stats_host && b < batch; ++b) {
        OFL_HIP(hipMemcpyAsync(words, (uint32_t *)dst.p + (size_t)b * 8, sizeof(words), hipMemcpyDeviceToHost, s));
        OFL_HIP(hipStreamSynchronize(s));
        uint32_t a = 0, bb = 0;
        for (int k = 0; k < 4; ++k) { a |= words[k] ? (1u << k) : 0u; bb |= words[4 + k] ? (1u << k) : 0u; }
        stats_host[2 * b] = a; stats_host[2 * b + 1] = bb;
    }
    OFL_HIP(hipStreamSynchronize(s));
    return OFL_OK;
}

int ofl_gather_bilinear_dev(const void *src, int dtype, int C, int H, int W,
                            const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                            const uint8_t *smask, const uint8_t *fmask,
                            void *dst, uint8_t *valid,
                            int quant, int arith, int rule, void *stream)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_gather_bilinear", H, W));
    if (!flow) return fail(OFL_E_INVALID, "ofl_gather_bilinear: NULL flow");
    if (C == 0) {   // validity-only launch: no image channels are read or written
        if (src || dst || !valid) return fail(OFL_E_INVALID, "ofl_gather_bilinear: C == 0 needs src = dst = NULL and valid != NULL");
        dtype = OFL_U8;
    } else if (!src || !dst) {
        return fail(OFL_E_INVALID, "ofl_gather_bilinear: NULL pointer");
    }
    if (C < 0) return fail(OFL_E_INVALID, "ofl_gather_bilinear: C must be >= 0");
    if (dtype_size(dtype) == 0) return fail(OFL_E_INVALID, "ofl_gather_bilinear: unsupported dtype %d", dtype);
    if (fH <= 0 || fW <= 0 || pad_top < 0 || pad_left < 0 || pad_top + fH > H || pad_left + fW > W)
        return fail(OFL_E_INVALID, "ofl_gather_bilinear: flow %dx%d at (%d,%d) does not fit target %dx%d",
                    fH, fW, pad_top, pad_left, H, W);
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_gather_bilinear: sign must be +1 or -1");
    if (quant != OFL_QUANT_OPENCV && quant != OFL_QUANT_EXACT) return fail(OFL_E_INVALID, "ofl_gather_bilinear: bad quant");
    if (rule < OFL_RULE_EQ1 || rule > OFL_RULE_GT_HALF) return fail(OFL_E_INVALID, "ofl_gather_bilinear: bad rule");
    if (arith != OFL_ARITH_NATIVE && arith != OFL_ARITH_FLOAT_RNE) return fail(OFL_E_INVALID, "ofl_gather_bilinear: bad arith");
    hipStream_t s = stream_of(stream);
    switch (dtype) {
    case OFL_U8:  return launch_gather_t<uint8_t>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, s);
    case OFL_I16: return launch_gather_t<int16_t>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, s);
    case OFL_U16: return launch_gather_t<uint16_t>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, s);
    case OFL_F32: return launch_gather_t<float>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, s);
    default:      return launch_gather_t<double>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, s);
    }
}

int ofl_gather_bilinear(const void *src, int dtype, int C, int H, int W,
                        const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                        const uint8_t *smask, const uint8_t *fmask,
                        void *dst, uint8_t *valid,
                        int quant, int arith, int rule)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_gather_bilinear", H, W));
    const size_t es = dtype_size(dtype);
    if (es == 0 || C <= 0) return fail(OFL_E_INVALID, "ofl_gather_bilinear: unsupported dtype/C");
    if (!src || !flow || !dst) return fail(OFL_E_INVALID, "ofl_gather_bilinear: NULL pointer");
    if (fH <= 0 || fW <= 0) return fail(OFL_E_INVALID, "ofl_gather_bilinear: bad flow shape");
    const size_t n = (size_t)H * W, nf = (size_t)fH * fW;
    hipStream_t s = rt().stream;
    DevBuf dsrc, dflow, dsm, dfm, ddst, dval;
    OFL_TRY(dsrc.alloc(n * C * es)); OFL_TRY(dflow.alloc(nf * 8)); OFL_TRY(ddst.alloc(n * C * es));
    OFL_HIP(hipMemcpyAsync(dsrc.p, src, n * C * es, hipMemcpyHostToDevice, s));
    OFL_HIP(hipMemcpyAsync(dflow.p, flow, nf * 8, hipMemcpyHostToDevice, s));
    if (smask) { OFL_TRY(dsm.alloc(n)); OFL_HIP(hipMemcpyAsync(dsm.p, smask, n, hipMemcpyHostToDevice, s)); }
    if (fmask) { OFL_TRY(dfm.alloc(nf)); OFL_HIP(hipMemcpyAsync(dfm.p, fmask, nf, hipMemcpyHostToDevice, s)); }
    if (valid) OFL_TRY(dval.alloc(n));
    OFL_TRY(ofl_gather_bilinear_dev(dsrc.p, dtype, C, H, W, (const float *)dflow.p, fH, fW, pad_top, pad_left, sign,
                                    (const uint8_t *)dsm.p, (const uint8_t *)dfm.p, ddst.p, (uint8_t *)dval.p,
                                    quant, arith, rule, s));
    OFL_HIP(hipMemcpyAsync(dst, ddst.p, n * C * es, hipMemcpyDeviceToHost, s));
    if (valid) OFL_HIP(hipMemcpyAsync(valid, dval.p, n, hipMemcpyDeviceToHost, s));
    OFL_HIP(hipStreamSynchronize(s));
    return OFL_OK;
}

}  // extern "C"
