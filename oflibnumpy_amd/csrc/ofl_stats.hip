// ofl_stats.hip -- K4 zero-flow / finite statistics and K5 element-wise epilogues (gfx950).
// Pure streaming kernels: 16-byte loads per lane, grid-stride over at most 2048 workgroups.
#include "ofl_common.h"
#include <stdlib.h>
#include <algorithm>

#pragma clang fp contract(off)

using namespace ofl;

namespace {

__device__ __forceinline__ bool finite2(float a, float b)
{
    return (fabsf(a) <= 3.402823466e38f) && (fabsf(b) <= 3.402823466e38f);   // false for NaN / Inf
}

// stats[0] receives the OR of OFL_STAT_* bits.  One atomicOr per WORKGROUP, and only for bits the
// word does not hold yet, keeps same-address traffic to a handful of operations per launch.
constexpr int kStatChunks = 8;
__global__ __launch_bounds__(256)
void flow_stats_kernel(const float *__restrict__ flow, const uint8_t *__restrict__ mask, size_t n_px,
                       float th, uint32_t *__restrict__ stats)
{
    __shared__ uint32_t block_bits;
    if (threadIdx.x == 0) block_bits = 0;
    __syncthreads();

    uint32_t bits = 0;
    const size_t n2 = n_px / 2;                    // pixel pairs (float4)
    // kStatChunks pairs per thread, a workgroup apart, all loads in flight before the first is looked at (one pair per thread and
    // workgroup -- 16 200 workgroups at 4K, each with its ballots, LDS atomic and barrier -- read a field at 1.9 TB/s)
    const size_t base = (size_t)blockIdx.x * (kStatChunks * 256) + threadIdx.x;
    float4   f4[kStatChunks];
    uint32_t m2[kStatChunks];
#pragma unroll
    for (int k = 0; k < kStatChunks; ++k) {
        const size_t i = base + (size_t)k * 256;
        f4[k] = i < n2 ? reinterpret_cast<const float4 *>(flow)[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        m2[k] = (i < n2 && mask) ? (uint32_t)reinterpret_cast<const uint16_t *>(mask)[i] : 0x0101u;
    }
#pragma unroll
    for (int k = 0; k < kStatChunks; ++k) {
        if (base + (size_t)k * 256 >= n2) continue;
        const float4 f = f4[k];
        const bool m0 = (m2[k] & 0xffu) != 0, m1 = (m2[k] & 0xff00u) != 0;
        bits |= stat_bits(f.x, f.y, m0, th) | stat_bits(f.z, f.w, m1, th);
        if (!(m0 && m1)) bits |= OFL_STAT_MASK_HAS_ZERO;
        if (!finite2(f.x, f.y) || !finite2(f.z, f.w)) bits |= OFL_STAT_NONFINITE;
    }
    if ((n_px & 1) && blockIdx.x == 0 && threadIdx.x == 0) {   // odd tail pixel
        const size_t i = n_px - 1;
        const float u = flow[2 * i], v = flow[2 * i + 1];
        bits |= stat_bits(u, v, mask ? mask[i] != 0 : true, th);
        if (mask && !mask[i]) bits |= OFL_STAT_MASK_HAS_ZERO;
        if (!finite2(u, v)) bits |= OFL_STAT_NONFINITE;
    }
    // wave OR via ballots, one LDS atomic per wave, one global atomic per workgroup
    uint32_t wbits = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) wbits |= (__ballot((bits >> k) & 1u) != 0ull) ? (1u << k) : 0u;
    if ((threadIdx.x & 63) == 0 && wbits) atomicOr(&block_bits, wbits);
    __syncthreads();
    if (threadIdx.x == 0 && block_bits) {
        const uint32_t cur = __hip_atomic_load(stats, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((cur | block_bits) != cur) atomicOr(stats, block_bits);
    }
}

// Extent of the sampling positions of the masked vectors (Flow.get_padding, flow_class.py:1214-1226):
// thresholded vectors (|component| < th -> 0, utils.py:310-315), position = float32(grid + sign * v) exactly as
// NumPy's in-place float32 -= int64 rounds it; min / max over the masked pixels through order-preserving
// uint32 keys and one atomicMin / atomicMax per wave and quantity.  ext = { min y, max y, min x, max x }.
__device__ __forceinline__ uint32_t order_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

__device__ __forceinline__ float order_value(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void flow_extent_init_kernel(uint32_t *ext)
{
    if (threadIdx.x < 4) ext[threadIdx.x] = (threadIdx.x & 1) ? 0u : 0xffffffffu;     // max slots / min slots
}

__global__ void flow_extent_finish_kernel(uint32_t *ext)
{
    if (threadIdx.x < 4) {
        const uint32_t k = ext[threadIdx.x];
        float v = order_value(k);
        if (!(threadIdx.x & 1) && k == 0xffffffffu) v = __uint_as_float(0x7f800000u);      // nothing masked: +inf
        if ((threadIdx.x & 1) && k == 0u) v = __uint_as_float(0xff800000u);                // nothing masked: -inf
        reinterpret_cast<float *>(ext)[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256)
void flow_extent_kernel(const float *__restrict__ flow, const uint8_t *__restrict__ mask, int H, int W, int sign,
                        float th, uint32_t *__restrict__ ext)
{
    const size_t n = (size_t)H * W, stride = (size_t)gridDim.x * blockDim.x;
    uint32_t kmin_y = 0xffffffffu, kmax_y = 0u, kmin_x = 0xffffffffu, kmax_x = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (mask && !mask[i]) continue;
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        float2 f = reinterpret_cast<const float2 *>(flow)[i];
        if (f.x < th && f.x > -th) f.x = 0.0f;
        if (f.y < th && f.y > -th) f.y = 0.0f;
        const uint32_t kx = order_key(map_coord(x, f.x, sign)), ky = order_key(map_coord(y, f.y, sign));
        kmin_x = min(kmin_x, kx); kmax_x = max(kmax_x, kx);
        kmin_y = min(kmin_y, ky); kmax_y = max(kmax_y, ky);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        kmin_y = min(kmin_y, (uint32_t)__shfl_xor((int)kmin_y, off)); kmax_y = max(kmax_y, (uint32_t)__shfl_xor((int)kmax_y, off));
        kmin_x = min(kmin_x, (uint32_t)__shfl_xor((int)kmin_x, off)); kmax_x = max(kmax_x, (uint32_t)__shfl_xor((int)kmax_x, off));
    }
    if ((threadIdx.x & 63) == 0) {
        if (kmin_y != 0xffffffffu) atomicMin(&ext[0], kmin_y);
        if (kmax_y != 0u) atomicMax(&ext[1], kmax_y);
        if (kmin_x != 0xffffffffu) atomicMin(&ext[2], kmin_x);
        if (kmax_x != 0u) atomicMax(&ext[3], kmax_x);
    }
}

// out = a + alpha * b, mout = ma & mb  (b/mb optional: out = alpha * a, mout = ma)
// Each thread moves two float4 (2 x 2 pixels, one per 1-KiB wave chunk) and one 4-byte mask word: every
// wave instruction covers contiguous memory, and the 1-byte mask streams need half the instructions
// the vector streams need instead of the same number.
__global__ __launch_bounds__(256)
void axpy_kernel(const float *__restrict__ a, const uint8_t *__restrict__ ma,
                 const float *__restrict__ b, const uint8_t *__restrict__ mb, float alpha, size_t n_px,
                 float *__restrict__ out, uint8_t *__restrict__ mout)
{
#ifndef OFL_AXPY_CHUNKS
#define OFL_AXPY_CHUNKS 2
#endif
    // work unit = 128 * OFL_AXPY_CHUNKS pixels per wave (float4 chunks of 128 px, 32 mask words each)
    constexpr int kCh = OFL_AXPY_CHUNKS, kUnit = 128 * kCh;
    const size_t n_units = n_px / kUnit;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t u = wave; u < n_units; u += n_waves) {
        const size_t p0 = u * kUnit;
#pragma unroll
        for (int h = 0; h < kCh; ++h) {
            const size_t i = (p0 + h * 128) / 2 + lane;          // float4 index (2 px)
            const float4 va = reinterpret_cast<const float4 *>(a)[i];
            float4 r;
            if (b) {
                const float4 vb = reinterpret_cast<const float4 *>(b)[i];
                r = make_float4(__fadd_rn(va.x, __fmul_rn(alpha, vb.x)), __fadd_rn(va.y, __fmul_rn(alpha, vb.y)),
                                __fadd_rn(va.z, __fmul_rn(alpha, vb.z)), __fadd_rn(va.w, __fmul_rn(alpha, vb.w)));
            } else {
                r = make_float4(__fmul_rn(alpha, va.x), __fmul_rn(alpha, va.y), __fmul_rn(alpha, va.z), __fmul_rn(alpha, va.w));
            }
            reinterpret_cast<float4 *>(out)[i] = r;
        }
        if (mout) {
#pragma unroll
            for (int h = 0; h < kCh / 2; ++h) {
                const size_t i = (p0 + h * 256) / 4 + lane;       // mask word index (4 px)
                uint32_t m = reinterpret_cast<const uint32_t *>(ma)[i];
                if (mb) m &= reinterpret_cast<const uint32_t *>(mb)[i];
                reinterpret_cast<uint32_t *>(mout)[i] = m;
            }
        }
    }
    // tail (< 256 px): one pixel per thread of the first workgroup
    const size_t tail0 = n_units * kUnit;
    if (blockIdx.x == 0) {
        for (size_t i = tail0 + threadIdx.x; i < n_px; i += blockDim.x) {
            for (int c = 0; c < 2; ++c)
                out[2 * i + c] = b ? __fadd_rn(a[2 * i + c], __fmul_rn(alpha, b[2 * i + c])) : __fmul_rn(alpha, a[2 * i + c]);
            if (mout) mout[i] = (uint8_t)(ma[i] & (mb ? mb[i] : 1));
        }
    }
}

__global__ __launch_bounds__(256)
void mask_and_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint8_t *__restrict__ out, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
        reinterpret_cast<uint32_t *>(out)[i] = reinterpret_cast<const uint32_t *>(a)[i] & reinterpret_cast<const uint32_t *>(b)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = n4 * 4 + threadIdx.x;
        out[i] = a[i] & b[i];
    }
}

// element-wise dtype conversion (the casts NumPy performs around griddata, utils.py:253 / :258)
template <typename S, typename D>
__global__ __launch_bounds__(256)
void convert_kernel(const S *__restrict__ src, D *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (D)src[i];
}

// out = float32(grid +- vecs): NumPy adds the int64 grid to the float32 array in float64 and rounds once
__global__ __launch_bounds__(256)
void grid_offset_kernel(const float *__restrict__ vecs, int sign, int H, int W, float *__restrict__ out)
{
    const size_t n = (size_t)H * W, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int y = (int)(i / W), x = (int)(i - (size_t)y * W);
        const float2 f = reinterpret_cast<const float2 *>(vecs)[i];
        reinterpret_cast<float2 *>(out)[i] = make_float2(map_coord(x, f.x, sign), map_coord(y, f.y, sign));
    }
}

// Sparse bilinear sampling of a flow field at float64 points (row, col), restating the reference's
// bilinear_interpolation(flow[..., ::-1], pts) (utils.py:161-196) in float64, INCLUDING its pairing of the
// (ver1, hor0) sample with the (ver0, hor1) weight and vice versa.  out[i] = (v, u) interpolated.
__global__ __launch_bounds__(256)
void sample_points_kernel(const float *__restrict__ flow, int H, int W, const double *__restrict__ pts, size_t n,
                          double *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double ver = pts[2 * i], hor = pts[2 * i + 1];
        const int v0 = (int)floor(ver), h0 = (int)floor(hor);
        const int v0c = min(max(v0, 0), H - 1), h0c = min(max(h0, 0), W - 1);
        const int v1c = min(max(v0 + 1, 0), H - 1), h1c = min(max(h0 + 1, 0), W - 1);
        const double w_a = ((double)v1c - ver) * ((double)h1c - hor), w_b = ((double)v1c - ver) * (hor - (double)h0c);
        const double w_c = (ver - (double)v0c) * ((double)h1c - hor), w_d = (ver - (double)v0c) * (hor - (double)h0c);
        const float2 da = reinterpret_cast<const float2 *>(flow)[(size_t)v0c * W + h0c];
        const float2 db = reinterpret_cast<const float2 *>(flow)[(size_t)v1c * W + h0c];
        const float2 dc = reinterpret_cast<const float2 *>(flow)[(size_t)v0c * W + h1c];
        const float2 dd = reinterpret_cast<const float2 *>(flow)[(size_t)v1c * W + h1c];
        out[2 * i]     = ((w_a * (double)da.y + w_b * (double)db.y) + w_c * (double)dc.y) + w_d * (double)dd.y;
        out[2 * i + 1] = ((w_a * (double)da.x + w_b * (double)db.x) + w_c * (double)dc.x) + w_d * (double)dd.x;
    }
}

#ifndef OFL_AXPY_CHUNKS
#define OFL_AXPY_CHUNKS 2
#endif

int stream_grid(size_t n_items)
{
    size_t nb = (n_items + 255) / 256;
    if (nb < 1) nb = 1;
    // One workgroup per 256 items, no persistent cap: on MI355X a copy that is dispatched as one workgroup per
    // chunk streams at 6.0-6.3 TB/s, the same loop on a resident-sized grid at 5.1-5.3 (tools/copy_sweep.hip).
    const size_t per_cu = (size_t)OFL_KNOB_INT("OFL_STREAM_GRID_CAP", 0);      // workgroups per CU, 0 = uncapped (experiments build only)
    const size_t cap = per_cu ? (size_t)rt().n_cu * per_cu : (size_t)0x7fffffff;
    return (int)(nb < cap ? nb : cap);
}

template <typename S, typename D>
void launch_convert(const void *src, void *dst, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL((convert_kernel<S, D>), dim3(stream_grid(n)), dim3(256), 0, s, (const S *)src, (D *)dst, n);
}

}  // namespace

extern "C" {

int ofl_flow_stats_dev(const float *flow, const uint8_t *mask, size_t n_px, float threshold,
                       uint32_t *stats, void *stream)
{
    OFL_TRY(need_device());
    if (!flow || !stats) return fail(OFL_E_INVALID, "ofl_flow_stats: NULL pointer");
    hipStream_t s = stream_of(stream);
    OFL_HIP(hipMemsetAsync(stats, 0, sizeof(uint32_t), s));
    if (n_px == 0) return OFL_OK;
    hipLaunchKernelGGL(flow_stats_kernel, dim3((unsigned)((n_px / 2 + kStatChunks * 256 - 1) / (kStatChunks * 256) + (n_px < 2 ? 1 : 0))), dim3(256), 0, s,
                       flow, mask, n_px, threshold, stats);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_flow_stats(const float *flow, const uint8_t *mask, size_t n_px, float threshold, uint32_t *stats_host)
{
    OFL_TRY(need_device());
    if (!flow || !stats_host) return fail(OFL_E_INVALID, "ofl_flow_stats: NULL pointer");
    hipStream_t s = rt().stream;
    void *df = nullptr, *dm = nullptr, *ds = nullptr;
    int rc = OFL_OK;
    hipError_t e;
    if ((e = hipMalloc(&df, n_px * 8 + 16)) != hipSuccess) return hip_fail(e, "hipMalloc");
    if ((e = hipMalloc(&ds, 16)) != hipSuccess) { (void)hipFree(df); return hip_fail(e, "hipMalloc"); }
    if (mask && (e = hipMalloc(&dm, n_px + 16)) != hipSuccess) { (void)hipFree(df); (void)hipFree(ds); return hip_fail(e, "hipMalloc"); }
    do {
        if ((e = hipMemcpyAsync(df, flow, n_px * 8, hipMemcpyHostToDevice, s)) != hipSuccess) { rc = hip_fail(e, "H2D"); break; }
        if (mask && (e = hipMemcpyAsync(dm, mask, n_px, hipMemcpyHostToDevice, s)) != hipSuccess) { rc = hip_fail(e, "H2D"); break; }
        if ((rc = ofl_flow_stats_dev((const float *)df, (const uint8_t *)dm, n_px, threshold, (uint32_t *)ds, s)) != OFL_OK) break;
        if ((e = hipMemcpyAsync(stats_host, ds, 4, hipMemcpyDeviceToHost, s)) != hipSuccess) { rc = hip_fail(e, "D2H"); break; }
        if ((e = hipStreamSynchronize(s)) != hipSuccess) { rc = hip_fail(e, "sync"); break; }
    } while (0);
    (void)hipFree(df); (void)hipFree(ds); if (dm) (void)hipFree(dm);
    return rc;
}

int ofl_axpy_dev(const float *a, const uint8_t *ma, const float *b, const uint8_t *mb, float alpha,
                 size_t n_px, float *out, uint8_t *mout, void *stream)
{
    OFL_TRY(need_device());
    if (!a || !out) return fail(OFL_E_INVALID, "ofl_axpy: NULL pointer");
    if (mout && !ma) return fail(OFL_E_INVALID, "ofl_axpy: mout requires ma");
    if (n_px == 0) return OFL_OK;
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(n_px / (2 * OFL_AXPY_CHUNKS))), dim3(256), 0, stream_of(stream),
                       a, ma, b, mb, alpha, n_px, out, mout);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_sample_points_dev(const float *flow, int H, int W, const double *pts_rc, size_t n, double *out_rc, void *stream)
{
    OFL_TRY(need_device());
    if (!flow || !pts_rc || !out_rc || H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_sample_points: bad arguments");
    if (n == 0) return OFL_OK;
    hipLaunchKernelGGL(sample_points_kernel, dim3(stream_grid(n)), dim3(256), 0, stream_of(stream), flow, H, W, pts_rc, n, out_rc);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_mask_and_dev(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, void *stream)
{
    OFL_TRY(need_device());
    if (!a || !b || !out) return fail(OFL_E_INVALID, "ofl_mask_and: NULL pointer");
    if (n == 0) return OFL_OK;
    hipLaunchKernelGGL(mask_and_kernel, dim3(stream_grid(n / 4)), dim3(256), 0, stream_of(stream), a, b, out, n);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_flow_extent_dev(const float *vecs, const uint8_t *mask, int H, int W, int sign, float threshold,
                        float *extent, void *stream)
{
    OFL_TRY(need_device());
    if (!vecs || !extent || H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_flow_extent: bad arguments");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_flow_extent: sign must be +1 or -1");
    hipStream_t s = stream_of(stream);
    uint32_t *ext = reinterpret_cast<uint32_t *>(extent);
    hipLaunchKernelGGL(flow_extent_init_kernel, dim3(1), dim3(64), 0, s, ext);
    const size_t n = (size_t)H * W;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);          // <= 4096 wave atomics per slot
    hipLaunchKernelGGL(flow_extent_kernel, dim3(grid), dim3(256), 0, s, vecs, mask, H, W, sign, threshold, ext);
    hipLaunchKernelGGL(flow_extent_finish_kernel, dim3(1), dim3(64), 0, s, ext);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_convert_dev(const void *src, int src_dtype, void *dst, int dst_dtype, size_t n, void *stream)
{
    OFL_TRY(need_device());
    if (!src || !dst) return fail(OFL_E_INVALID, "ofl_convert: NULL pointer");
    if (n == 0) return OFL_OK;
    hipStream_t s = stream_of(stream);
    if (dst_dtype == OFL_F32) {
        switch (src_dtype) {
        case OFL_U8:  launch_convert<uint8_t, float>(src, dst, n, s); break;
        case OFL_I16: launch_convert<int16_t, float>(src, dst, n, s); break;
        case OFL_U16: launch_convert<uint16_t, float>(src, dst, n, s); break;
        case OFL_F32: launch_convert<float, float>(src, dst, n, s); break;
        case OFL_F64: launch_convert<double, float>(src, dst, n, s); break;
        default: return fail(OFL_E_INVALID, "ofl_convert: unsupported source dtype %d", src_dtype);
        }
    } else if (src_dtype == OFL_F32) {
        switch (dst_dtype) {
        case OFL_U8:  launch_convert<float, uint8_t>(src, dst, n, s); break;
        case OFL_I16: launch_convert<float, int16_t>(src, dst, n, s); break;
        case OFL_U16: launch_convert<float, uint16_t>(src, dst, n, s); break;
        case OFL_F64: launch_convert<float, double>(src, dst, n, s); break;
        default: return fail(OFL_E_INVALID, "ofl_convert: unsupported destination dtype %d", dst_dtype);
        }
    } else {
        return fail(OFL_E_INVALID, "ofl_convert: one side must be OFL_F32");
    }
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_grid_offset_dev(const float *vecs, int sign, int H, int W, float *out, void *stream)
{
    OFL_TRY(need_device());
    if (!vecs || !out || H <= 0 || W <= 0) return fail(OFL_E_INVALID, "ofl_grid_offset: bad arguments");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_grid_offset: sign must be +1 or -1");
    hipLaunchKernelGGL(grid_offset_kernel, dim3(stream_grid((size_t)H * W)), dim3(256), 0, stream_of(stream), vecs, sign, H, W, out);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

}  // extern "C"
