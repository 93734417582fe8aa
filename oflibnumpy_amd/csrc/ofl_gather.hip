// ofl_gather.hip -- K1 general bilinear gather and K2 fused mode-3 composition for gfx950.
//
// Both kernels are HBM-bound byte movers (no MFMA): every output pixel streams its flow vector
// (coalesced, 16 B per lane), derives the cv2.remap sample position, and gathers the 2x2 source
// neighbourhood.  The two horizontally adjacent taps of one source row are fetched with ONE
// 8-byte-aligned 16-byte load; reuse between neighbouring pixels is served by the per-CU L1 and the
// per-XCD L2, which is why workgroups own 2-D tiles and consecutive tiles of one XCD are neighbours
// (blocks b and b+8 share an XCD on MI355X).
//
// Numerics follow oracle/ofl_oracle.c operation for operation (contraction disabled) so that the
// float results and the validity masks are bit-identical to the CPU restatement.
#include "ofl_common.h"
#include <type_traits>

#pragma clang fp contract(off)

using namespace ofl;

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(8))) Pair2 { float lo_u, lo_v, hi_u, hi_v; };   // taps (ix, ix+1) of a float2 field
struct __attribute__((packed, aligned(1))) U32u { uint32_t v; };               // a 4-byte load from any address

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

// Packed mask planes (device-resident chains, ofl_compose3_bits_dev): one BIT per pixel, bit x & 31 of the 32-bit word x >> 5,
// rows padded to whole words.  The three 1-byte mask streams of the fused compose are 3 of its 27 B/px by the contract but up
// to five times that in partially used 128-byte lines on a rotated sampling grid; as bit planes they are 0.4 B/px.
// taps (x, x + 1) of a plane row as bits 0 and 1 -- one 4-byte load from the byte that holds bit x (x & 7 <= 7: both bits lie
// within the 32; planes are allocated with 16 bytes of slack, ofl_mask_bits_bytes)
__device__ __forceinline__ uint32_t c3_bits_at(const uint8_t *plane_row, int x)
{
    return reinterpret_cast<const U32u *>(plane_row + (x >> 3))->v >> (x & 7);
}
__device__ __forceinline__ uint32_t c3_bit(const uint8_t *plane_row, int x) { return ((uint32_t)plane_row[x >> 3] >> (x & 7)) & 1u; }
// 16 bits -> the even bit positions of 32
__device__ __forceinline__ uint32_t c3_spread16(uint32_t x)
{
    x &= 0xFFFFu;
    x = (x | (x << 8)) & 0x00FF00FFu; x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u; x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// ------------------------------------------------------------------------------------ K2
// Workgroup tile: 128 px x 8 rows, 256 threads.  Lane (lx, ly) owns the two pixel PAIRS at
// x = 2*lx and x = 64 + 2*lx of its tile row, so that every stream instruction of a wave covers
// contiguous memory (32 lanes x 16 B = 512 B of vectors, 32 lanes x 2 B = 64 B of mask per row):
// no half-filled 64-byte requests on either the read or the write side.
#ifndef OFL_C3_LANES_X
#define OFL_C3_LANES_X 32            // lanes along x per tile row (16 / 32 / 64): tile = 4*LX px x 256/LX rows
#endif
#ifndef OFL_C3_NT
#define OFL_C3_NT 3                  // experiment knob: 1 = non-temporal stores, 2 = non-temporal stream loads
#endif
constexpr int kC3LanesX = OFL_C3_LANES_X;
constexpr int kC3TileW = 4 * kC3LanesX, kC3TileH = 256 / kC3LanesX, kC3Px = 4;

struct C3Pos {            // what must stay live while the gather is in flight
    int   ix, iy;         // top-left tap
    float fx, fy;         // fractional position (multiples of 1/32 with cv2's snapping)
};

struct C3Tap {
    int   ax, ay;         // "fraction is non-zero" indicators for the validity test
    float w0, w1, w2, w3;
};

// Sample position: utils.py:231-235 evaluates float32(float64(grid) -/+ float64(flow)).  For an integer
// grid coordinate below 2^15 and any float32 flow value this equals the single float32 operation
// (the float64 sum is exact unless |flow| < 2^-14, and then it cannot land on a float32 rounding
// midpoint, whose distance from the integer is at least 2^-24 * 2^floor(log2 grid)), so one v_add_f32
// / v_sub_f32 reproduces NumPy bit for bit; tests compare against the float64 oracle.
template <int QUANT>
__device__ __forceinline__ C3Pos c3_pos(int gx, int gy, float fu, float fv, int sign)
{
    const float px = sign >= 0 ? __fadd_rn((float)gx, fu) : __fsub_rn((float)gx, fu);
    const float py = sign >= 0 ? __fadd_rn((float)gy, fv) : __fsub_rn((float)gy, fv);
    C3Pos p;
    if (QUANT == OFL_QUANT_OPENCV) {
        const int sx = cv_round(__fmul_rn(px, 32.0f)), sy = cv_round(__fmul_rn(py, 32.0f));
        p.ix = sx >> 5; p.iy = sy >> 5;      // unsaturated: beyond +-32767 every tap is outside anyway
        p.fx = (float)(sx & 31) * (1.0f / 32.0f);
        p.fy = (float)(sy & 31) * (1.0f / 32.0f);
    } else {
        float flx = floorf(px), fly = floorf(py);
        p.fx = __fsub_rn(px, flx); p.fy = __fsub_rn(py, fly);
        flx = fminf(fmaxf(flx, -32768.0f), 32767.0f);
        fly = fminf(fmaxf(fly, -32768.0f), 32767.0f);
        p.ix = (int)flx; p.iy = (int)fly;
    }
    return p;
}

// Bilinear weights from the fractional position; evaluated AFTER the gather has been issued so that
// they do not occupy registers while the loads are in flight.
template <int QUANT>
__device__ __forceinline__ C3Tap c3_weights(const C3Pos &p)
{
    C3Tap t;
    t.ax = (QUANT == OFL_QUANT_OPENCV) ? (p.fx != 0.0f) : 1;
    t.ay = (QUANT == OFL_QUANT_OPENCV) ? (p.fy != 0.0f) : 1;
    const float x0 = __fsub_rn(1.0f, p.fx), y0 = __fsub_rn(1.0f, p.fy);
    t.w0 = __fmul_rn(y0, x0); t.w1 = __fmul_rn(y0, p.fx);
    t.w2 = __fmul_rn(p.fy, x0); t.w3 = __fmul_rn(p.fy, p.fx);
    return t;
}

__device__ __forceinline__ float c3_blend(float v00, float v01, float v10, float v11, const C3Tap &t)
{
    float s = __fmul_rn(v00, t.w0);
    s = __fadd_rn(s, __fmul_rn(v01, t.w1));
    s = __fadd_rn(s, __fmul_rn(v10, t.w2));
    s = __fadd_rn(s, __fmul_rn(v11, t.w3));
    return s;
}

// "interpolated mask == 1" (flow_class.py:668).  With cv2's 1/32-px weights (multiples of 1/1024
// that sum to exactly 1) the float sum equals 1 iff every tap with a non-zero weight carries mask 1,
// which is pure integer logic; the un-snapped extension mode keeps the float comparison.
template <int QUANT>
__device__ __forceinline__ bool c3_valid(bool m00, bool m01, bool m10, bool m11, const C3Tap &t)
{
    if (QUANT == OFL_QUANT_OPENCV) {
        const bool zx = t.ax == 0, zy = t.ay == 0;
        return m00 & (m01 | zx) & (m10 | zy) & (m11 | zx | zy);
    }
    return c3_blend(m00 ? 1.0f : 0.0f, m01 ? 1.0f : 0.0f, m10 ? 1.0f : 0.0f, m11 ? 1.0f : 0.0f, t) == 1.0f;
}

struct C3Args {
    const float *fa; const uint8_t *ma; const float *fb; const uint8_t *mb;
    float *out; uint8_t *mout; uint32_t *stats;
    int sign, H, W, tiles_x, tiles_per_field, ntiles;
    float th;
    int mwpr;              // BITS kernels: 32-bit words per row of the packed mask planes ma / mb / mout (bit x & 31 of word x >> 5); else 0
    int swz_group;         // default kernel: tile rows one XCD owns per group (<= 1: natural order); one-shot variant: workgroups per XCD-swizzle group
    int xpose_rows;        // transposed-gather kernel: source rows a streamed 128-px segment may cross on the direct path
#ifdef OFL_EXPERIMENTS
    int ablate;            // OFL_C3_ABLATE (experiments build only): 1 = skip the gather, 2 = skip the stores, ...
#endif
};
#ifdef OFL_EXPERIMENTS
#define OFL_ABLATE(a, bit) (((a).ablate & (bit)) != 0)
#else
#define OFL_ABLATE(a, bit) false
#endif

struct C3Stream {          // one lane's share of a tile row of the streamed field fb/mb
    float4   v[2];
    uint32_t m[2];
};

struct C3Stat {            // running maxima for the zero-flow predicates of one field pair
    float amax_m, bmax, bmax_m;
};

// LX = lanes along x per tile row: tile = 4*LX px wide, 256/LX rows high; lane (lx, ly) owns the pixel
// pairs at x = 2*lx and x = 2*LX + 2*lx.
template <int LX>
__device__ __forceinline__ void c3_tile_coords(const C3Args &a, int tile, int &b, int &y, int (&xg)[2])
{
    b = tile / a.tiles_per_field;
    const int t  = tile - b * a.tiles_per_field;
    const int ty = t / a.tiles_x, tx = t - ty * a.tiles_x;
    y = ty * (256 / LX) + (threadIdx.x / LX);
    xg[0] = tx * (4 * LX) + 2 * (threadIdx.x % LX);
    xg[1] = xg[0] + 2 * LX;
}

template <int LX, bool BITS = false>
__device__ __forceinline__ C3Stream c3_load_stream(const C3Args &a, int tile)
{
    int b, y, xg[2];
    c3_tile_coords<LX>(a, tile, b, y, xg);
    const size_t base = (size_t)b * a.H * a.W + (size_t)y * a.W;
    C3Stream s;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        s.v[g] = make_float4(0.f, 0.f, 0.f, 0.f);
        s.m[g] = 0;
        if (BITS) {
            if (y < a.H && xg[g] < a.W) {
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a.fb + 2 * (base + xg[g])));
                s.v[g] = make_float4(t.x, t.y, t.z, t.w);
                // the pair's two bits (xg is even: both in one word; the 16 lanes of a word read the same address)
                const uint32_t w = reinterpret_cast<const uint32_t *>(a.mb)[((size_t)b * a.H + y) * a.mwpr + (xg[g] >> 5)] >> (xg[g] & 31);
                s.m[g] = (w & 1u) | ((w & 2u) << 7);
            }
            continue;
        }
        if (y < a.H && xg[g] < a.W) {
            if (OFL_C3_NT & 2) {
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(a.fb + 2 * (base + xg[g])));
                s.v[g] = make_float4(t.x, t.y, t.z, t.w);
                s.m[g] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(a.mb + base + xg[g]));
            } else {
                s.v[g] = *reinterpret_cast<const float4 *>(a.fb + 2 * (base + xg[g]));
                s.m[g] = *reinterpret_cast<const uint16_t *>(a.mb + base + xg[g]);
            }
        }
    }
    return s;
}

__device__ __forceinline__ void c3_flush_stats(const C3Args &a, int b, const C3Stat &st)
{
    // Flag words (set-only, plain idempotent stores; thousands of waves issuing atomics on one
    // address would serialise at the memory side).  fb: all four predicates, exact.  fa: words 0/1
    // are set when a GATHERED masked vector is non-zero / above threshold -- a certificate that fa
    // is not zero; a clear word means "not observed" and is confirmed by ofl_flow_stats_dev.
    const bool c[6] = { st.amax_m > 0.0f, st.amax_m >= a.th, st.bmax_m > 0.0f, st.bmax_m >= a.th,
                        st.bmax > 0.0f, st.bmax >= a.th };
    const int  slot[6] = { 0, 1, 4, 5, 6, 7 };
    uint32_t *s = a.stats + (size_t)b * 8;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const bool any = __ballot(c[k]) != 0ull;
        if (any && (threadIdx.x & 63) == 0 &&
            __hip_atomic_load(s + slot[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
            s[slot[k]] = 1u;
    }
}

// stream out: out = fb + B(fa), mout = mb & valid   (flow_class.py:332-334, 668, 680)
template <bool STATS, bool BITS = false>
__device__ __forceinline__ void c3_finish(const C3Args &a, size_t row, const int (&xg)[2], const bool (&act)[2],
                                          const float (&bu)[kC3Px], const float (&bv)[kC3Px], const bool (&bm)[kC3Px],
                                          const float (&su)[kC3Px], const float (&sv)[kC3Px], const bool (&ok)[kC3Px],
                                          C3Stat &st)
{
    if constexpr (BITS) {
        // vectors as in the byte form; the mask leaves as whole words: a wave is two tile rows (lanes 0 .. 31 / 32 .. 63), the
        // ballots of the pairs' even and odd pixels are interleaved by the lanes that store (lane 0 / 16 of a row: the words of
        // pixels 0 .. 31 / 32 .. 63 of the 64-px stretch)
        const int lx = threadIdx.x & 31;
        const size_t wrow = (row / (size_t)a.W) * (size_t)a.mwpr;       // row = (b * H + y) * W
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int j = 2 * g;
            const bool be = act[g] && ok[j] && bm[j], bo = act[g] && ok[j + 1] && bm[j + 1];
            const unsigned long long me = __ballot(be), mo = __ballot(bo);
            if (act[g]) {
                const float4 o4 = make_float4(__fadd_rn(bu[j], su[j]), __fadd_rn(bv[j], sv[j]),
                                              __fadd_rn(bu[j + 1], su[j + 1]), __fadd_rn(bv[j + 1], sv[j + 1]));
                const v4f t = { o4.x, o4.y, o4.z, o4.w };
                __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(a.out + 2 * (row + xg[g])));
                if (STATS) {
                    const float a0 = fmaxf(fabsf(bu[j]), fabsf(bv[j])), a1 = fmaxf(fabsf(bu[j + 1]), fabsf(bv[j + 1]));
                    st.bmax   = fmaxf(st.bmax, fmaxf(a0, a1));
                    st.bmax_m = fmaxf(st.bmax_m, fmaxf(bm[j] ? a0 : 0.0f, bm[j + 1] ? a1 : 0.0f));
                }
            }
            // (lane 0 / 16 of a row: its pair starts the word, so its own `act` says whether the word exists -- x < W, y < H)
            if ((lx & 15) == 0 && act[g]) {
                const uint32_t e32 = (uint32_t)(me >> (threadIdx.x & 32)), o32 = (uint32_t)(mo >> (threadIdx.x & 32));
                const uint32_t word = c3_spread16(e32 >> lx) | (c3_spread16(o32 >> lx) << 1);
                __builtin_nontemporal_store(word, reinterpret_cast<uint32_t *>(a.mout) + wrow + (xg[g] >> 5));
            }
        }
        return;
    }
    // Mask bytes leave as DWORDS when the row length allows aligned ones: the lane pairs (2k, 2k + 1) own four consecutive
    // pixels, the even lane stores both lanes' bytes (sub-dword stores cost as much per instruction as 16-byte ones)
    const bool quad = (a.W & 3) == 0 && !OFL_ABLATE(a, 64);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int jj = 2 * g;
        const uint32_t mine = ((ok[jj] && bm[jj]) ? 1u : 0u) | ((ok[jj + 1] && bm[jj + 1]) ? 0x100u : 0u);
        const uint32_t other = (uint32_t)__shfl_xor((int)mine, 1);
        if (act[g] && !(OFL_ABLATE(a, 2) && su[2 * g] != 12345.0f)) {
            const int j = 2 * g;
            const float4 o4 = make_float4(__fadd_rn(bu[j], su[j]), __fadd_rn(bv[j], sv[j]),
                                          __fadd_rn(bu[j + 1], su[j + 1]), __fadd_rn(bv[j + 1], sv[j + 1]));
            const uint16_t mo = (uint16_t)mine;
            if (OFL_C3_NT & 1) {
                const v4f t = { o4.x, o4.y, o4.z, o4.w };
                __builtin_nontemporal_store(t, reinterpret_cast<v4f *>(a.out + 2 * (row + xg[g])));
                if (quad) {
                    if ((threadIdx.x & 1) == 0) __builtin_nontemporal_store(mine | (other << 16), reinterpret_cast<uint32_t *>(a.mout + row + xg[g]));
                } else {
                    __builtin_nontemporal_store(mo, reinterpret_cast<uint16_t *>(a.mout + row + xg[g]));
                }
            } else {
                *reinterpret_cast<float4 *>(a.out + 2 * (row + xg[g])) = o4;
                *reinterpret_cast<uint16_t *>(a.mout + row + xg[g]) = mo;
            }
            if (STATS) {
                const float a0 = fmaxf(fabsf(bu[j]), fabsf(bv[j])), a1 = fmaxf(fabsf(bu[j + 1]), fabsf(bv[j + 1]));
                st.bmax   = fmaxf(st.bmax, fmaxf(a0, a1));
                st.bmax_m = fmaxf(st.bmax_m, fmaxf(bm[j] ? a0 : 0.0f, bm[j + 1] ? a1 : 0.0f));
            }
        }
    }
}

// One tile: taps from the already loaded stream data, gather, blend, store.
template <int QUANT, bool STATS, bool BITS = false>
__device__ __forceinline__ void c3_tile(const C3Args &a, int tile, const C3Stream &in, C3Stat &st)
{
    int b, y, xg[2];
    c3_tile_coords<kC3LanesX>(a, tile, b, y, xg);
    const int H = a.H, W = a.W, sign = a.sign;
    const size_t field = (size_t)b * H * W;
    const float   *fa = a.fa + field * 2;
    const uint8_t *ma = BITS ? a.ma + (size_t)b * H * a.mwpr * 4 : a.ma + field;      // BITS: the field's bit plane, rows of mwpr words
    const int      mrow = a.mwpr * 4;                                                  // ... and its row pitch in bytes
    const bool act[2] = { y < H && xg[0] < W, y < H && xg[1] < W };
    const size_t row = field + (size_t)y * W;

    const float bu[kC3Px] = { in.v[0].x, in.v[0].z, in.v[1].x, in.v[1].z };
    const float bv[kC3Px] = { in.v[0].y, in.v[0].w, in.v[1].y, in.v[1].w };
    const bool  bm[kC3Px] = { (in.m[0] & 0xffu) != 0, (in.m[0] & 0xff00u) != 0,
                              (in.m[1] & 0xffu) != 0, (in.m[1] & 0xff00u) != 0 };

    C3Pos tp[kC3Px];
    bool inside = true, outside = true;
#pragma unroll
    for (int j = 0; j < kC3Px; ++j) {
        tp[j] = c3_pos<QUANT>(xg[j >> 1] + (j & 1), y, bu[j], bv[j], sign);
        const bool in_j  = (unsigned)tp[j].ix <= (unsigned)(W - 2) && (unsigned)tp[j].iy <= (unsigned)(H - 2);
        const bool out_j = tp[j].ix < -1 || tp[j].ix >= W || tp[j].iy < -1 || tp[j].iy >= H;
        inside  = inside && (in_j || !act[j >> 1]);
        outside = outside && (out_j || !act[j >> 1]);
    }
    if (H < 2) inside = false;
    if (OFL_ABLATE(a, 1)) outside = true;

    float su[kC3Px], sv[kC3Px];
    bool  ok[kC3Px];
#pragma unroll
    for (int j = 0; j < kC3Px; ++j) { su[j] = 0.0f; sv[j] = 0.0f; ok[j] = false; }

    if (__all(outside)) {
        // every tap of every pixel of this wave lies outside the source: B(.) = 0, nothing to fetch
    } else if (__all(inside)) {
        // interior fast path: all four taps in bounds, no clamping and no selects
        Pair2    p0[kC3Px], p1[kC3Px];
        uint32_t m0[kC3Px], m1[kC3Px];
        // The two pixels of a pair usually sample the same two source rows at columns at most two apart (axis-aligned
        // sampling: scalings, translations): their eight mask taps are then four bytes of each row -- ONE unaligned
        // 4-byte load per row and pair instead of one 2-byte load per row and pixel (4 instead of 8 mask gathers per lane).
        bool share = true;
#pragma unroll
        for (int g = 0; g < 2; ++g)
            share = share && (!act[g] || (tp[2 * g].iy == tp[2 * g + 1].iy && (unsigned)(tp[2 * g + 1].ix - tp[2 * g].ix) <= 2u &&
                                          tp[2 * g].ix + 3 < W));
        share = __all(share) && !OFL_ABLATE(a, 32);
#pragma unroll
        for (int j = 0; j < kC3Px; ++j) {
            const size_t s0 = act[j >> 1] ? (size_t)tp[j].iy * W + tp[j].ix : 0;
            p0[j] = *reinterpret_cast<const Pair2 *>(fa + 2 * s0);
            p1[j] = *reinterpret_cast<const Pair2 *>(fa + 2 * (s0 + W));
            if (OFL_ABLATE(a, 16)) { m0[j] = 0x0101u; m1[j] = 0x0101u; continue; }       // TA-cost probe: no mask gathers
            if (BITS) {
                // bits 0 / 1 = taps ix / ix + 1 (bits 2, 3 serve the pair's other pixel when it shares the load)
                if (share && (j & 1)) {
                    const int sh = act[j >> 1] ? tp[j].ix - tp[j - 1].ix : 0;
                    m0[j] = m0[j - 1] >> sh;
                    m1[j] = m1[j - 1] >> sh;
                } else {
                    const uint8_t *r0 = ma + (size_t)(act[j >> 1] ? tp[j].iy : 0) * mrow;
                    const int ix = act[j >> 1] ? tp[j].ix : 0;
                    m0[j] = c3_bits_at(r0, ix);
                    m1[j] = c3_bits_at(r0 + mrow, ix);
                }
                continue;
            }
            if (share) {
                if ((j & 1) == 0) {
                    m0[j] = reinterpret_cast<const U32u *>(ma + s0)->v;
                    m1[j] = reinterpret_cast<const U32u *>(ma + s0 + W)->v;
                } else {
                    const int sh = act[j >> 1] ? 8 * (tp[j].ix - tp[j - 1].ix) : 0;
                    m0[j] = m0[j - 1] >> sh;
                    m1[j] = m1[j - 1] >> sh;
                }
            } else {
                m0[j] = (uint32_t)ma[s0] | ((uint32_t)ma[s0 + 1] << 8);
                m1[j] = (uint32_t)ma[s0 + W] | ((uint32_t)ma[s0 + W + 1] << 8);
            }
        }
#pragma unroll
        for (int j = 0; j < kC3Px; ++j) {
            const C3Tap w = c3_weights<QUANT>(tp[j]);
            su[j] = c3_blend(p0[j].lo_u, p0[j].hi_u, p1[j].lo_u, p1[j].hi_u, w);
            sv[j] = c3_blend(p0[j].lo_v, p0[j].hi_v, p1[j].lo_v, p1[j].hi_v, w);
            const bool m00 = (m0[j] & (BITS ? 1u : 0xffu)) != 0, m01 = (m0[j] & (BITS ? 2u : 0xff00u)) != 0;
            const bool m10 = (m1[j] & (BITS ? 1u : 0xffu)) != 0, m11 = (m1[j] & (BITS ? 2u : 0xff00u)) != 0;
            ok[j] = c3_valid<QUANT>(m00, m01, m10, m11, w);
            if (STATS) st.amax_m = fmaxf(st.amax_m, (m00 && act[j >> 1]) ? fmaxf(fabsf(p0[j].lo_u), fabsf(p0[j].lo_v)) : 0.0f);
        }
    } else {
        // border path: clamp the addresses, zero the taps that fall outside (cv2 BORDER_CONSTANT 0)
#pragma unroll
        for (int j = 0; j < kC3Px; ++j) {
            const int  ixc = min(max(tp[j].ix, 0), max(W - 2, 0));
            const int  d   = tp[j].ix - ixc;
            const bool r0  = (unsigned)tp[j].iy < (unsigned)H;
            const bool r1  = (unsigned)(tp[j].iy + 1) < (unsigned)H;
            const int  y0c = min(max(tp[j].iy, 0), H - 1);
            const int  y1c = min(max(tp[j].iy + 1, 0), H - 1);
            const size_t s0 = (size_t)y0c * W + ixc, s1 = (size_t)y1c * W + ixc;
            const Pair2 p0 = *reinterpret_cast<const Pair2 *>(fa + 2 * s0);
            const Pair2 p1 = *reinterpret_cast<const Pair2 *>(fa + 2 * s1);
            uint32_t q00, q01, q10, q11;
            if (BITS) {
                const uint8_t *r0p = ma + (size_t)y0c * mrow, *r1p = ma + (size_t)y1c * mrow;
                q00 = c3_bit(r0p, ixc); q01 = c3_bit(r0p, min(ixc + 1, W - 1)); q10 = c3_bit(r1p, ixc); q11 = c3_bit(r1p, min(ixc + 1, W - 1));
            } else { q00 = ma[s0]; q01 = ma[s0 + 1]; q10 = ma[s1]; q11 = ma[s1 + 1]; }
            float u00, v00, u01, v01, u10, v10, u11, v11, a00, a01, a10, a11;
            select_pair(p0, d, r0, u00, v00, u01, v01);
            select_pair(p1, d, r1, u10, v10, u11, v11);
            select_mask(q00, q01, d, r0, a00, a01);
            select_mask(q10, q11, d, r1, a10, a11);
            const C3Tap w = c3_weights<QUANT>(tp[j]);
            su[j] = c3_blend(u00, u01, u10, u11, w);
            sv[j] = c3_blend(v00, v01, v10, v11, w);
            ok[j] = c3_valid<QUANT>(a00 != 0.0f, a01 != 0.0f, a10 != 0.0f, a11 != 0.0f, w);
            if (STATS) st.amax_m = fmaxf(st.amax_m, (a00 != 0.0f && act[j >> 1]) ? fmaxf(fabsf(u00), fabsf(v00)) : 0.0f);
        }
    }

    c3_finish<STATS, BITS>(a, row, xg, act, bu, bv, bm, su, sv, ok, st);
}

#ifndef OFL_C3_ONESHOT_WAVES
#define OFL_C3_ONESHOT_WAVES 6       // waves per SIMD the register allocator is asked to leave room for
#endif

#ifdef OFL_EXPERIMENTS   // ---- non-default compose variants (OFL_C3_VARIANT = 0 / 2): measured, kept for A/B runs only
// Persistent form: gridDim.x workgroups (a multiple of 8, sized to the chip's residency) walk the tile
// list with stride gridDim.x.  The stream loads of the NEXT tile are issued before the gathers of the
// current one are consumed, which takes the fb round trip out of every tile's dependency chain
// (load fb -> addresses -> gather -> store), and the launch pays one ramp-up / tail instead of one per
// resident-set of workgroups.  At every step the workgroups of one XCD (blockIdx % 8) own a contiguous
// run of tiles, so neighbouring tiles share gather halos in that XCD's L2.
#ifndef OFL_C3_PREFETCH
#define OFL_C3_PREFETCH 1            // tiles of stream data kept in flight ahead of the one being gathered (1 or 2)
#endif

template <int QUANT, bool STATS>
__global__ __launch_bounds__(256)
void compose3_kernel(const C3Args a)
{
    const int nb  = gridDim.x;
    const int per = nb >> 3;
    const int lane_tile = ((nb & 7) == 0 && a.swz_group != 0) ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
    int tile = lane_tile;
    if (tile >= a.ntiles) return;
    C3Stat st = { 0.0f, 0.0f, 0.0f };
    int cur_b = tile / a.tiles_per_field;
    C3Stream in = c3_load_stream<kC3LanesX>(a, tile);
#if OFL_C3_PREFETCH == 2
    C3Stream in2;
    if (tile + nb < a.ntiles) in2 = c3_load_stream<kC3LanesX>(a, tile + nb);
#endif
    while (true) {
        const int next = tile + nb;
        const bool more = next < a.ntiles;
        C3Stream nxt;
#if OFL_C3_PREFETCH == 2
        if (next + nb < a.ntiles) nxt = c3_load_stream<kC3LanesX>(a, next + nb);   // two tiles ahead
#else
        if (more) nxt = c3_load_stream<kC3LanesX>(a, next);          // prefetch: in flight during this tile's gather
#endif
        c3_tile<QUANT, STATS>(a, tile, in, st);
        if (!more) break;
        if (STATS) {
            const int nb_ = next / a.tiles_per_field;
            if (nb_ != cur_b) {
                c3_flush_stats(a, cur_b, st);
                st.amax_m = st.bmax = st.bmax_m = 0.0f;
                cur_b = nb_;
            }
        }
#if OFL_C3_PREFETCH == 2
        in = in2;
        in2 = nxt;
#else
        in = nxt;
#endif
        tile = next;
    }
    if (STATS) c3_flush_stats(a, cur_b, st);
}

// One workgroup per tile (no persistence, no prefetch): the hardware dispatcher keeps every wave slot
// filled and the active tiles form a window that sweeps memory in order.
template <int QUANT, bool STATS>
__global__ __launch_bounds__(256, OFL_C3_ONESHOT_WAVES)
void compose3_oneshot_kernel(const C3Args a)
{
    // XCD-aware tile order inside groups of a.swz_group workgroups: the 8 XCDs sweep ONE window of memory
    // together, each owning a contiguous eighth of it (neighbouring tiles share gather halos in one L2)
    int tile = blockIdx.x;
    if (a.swz_group > 0) {
        const int g = blockIdx.x / a.swz_group, r = blockIdx.x - g * a.swz_group;
        const int size = min(a.swz_group, (int)gridDim.x - g * a.swz_group);
        tile = g * a.swz_group + xcd_swizzle(r, size);
    }
    if (tile >= a.ntiles) return;
    C3Stat st = { 0.0f, 0.0f, 0.0f };
    const C3Stream in = c3_load_stream<kC3LanesX>(a, tile);
    c3_tile<QUANT, STATS>(a, tile, in, st);
    if (STATS) c3_flush_stats(a, tile / a.tiles_per_field, st);
}
#endif  // OFL_EXPERIMENTS

#ifdef OFL_EXPERIMENTS   // ---- LDS-staged compose variant (OFL_C3_VARIANT = 1)
// ------------------------------------------------------------------------------------ K2, LDS-staged form
// The direct form above fetches 36 B per output pixel through the texture path (two unaligned 16-byte
// gathers + two 2-byte mask gathers) and its row-strip tiles lose L1 locality when the sampling grid is
// rotated.  Here a workgroup owns a compact 32 x 32 output tile, finds the bounding box of its sample
// positions (wave shuffles + 4 LDS atomics per wave), stages that source rectangle ONCE with coalesced,
// aligned 16-byte row loads into LDS as {u, v, mask, -} quads, and reads the taps with ds_read_b128.
// Texture-path bytes per pixel drop to 9 (stream) + ~9 x (source px per output px), independent of the
// rotation of the sampling grid.  Tiles whose footprint does not fit the LDS budget use the direct path.
#ifndef OFL_LDS_LX
#define OFL_LDS_LX 8
#endif
#ifndef OFL_LDS_CAP
#define OFL_LDS_CAP 2400
#endif
constexpr int kLdsLX  = OFL_LDS_LX;   // 8 lanes x 4 px = 32 px wide, 32 rows (16 -> 64 x 16)
constexpr int kLdsCap = OFL_LDS_CAP;  // staged source pixels per workgroup (16 B each; 2400 = 37.5 KB -> 4 workgroups / CU)

// Wave-wide min / max on the VALU with DPP row shifts and row broadcasts (gfx9 reduction idiom): four
// row_shr steps leave each 16-lane row's result in its last lane, row_bcast:15 / row_bcast:31 carry it
// across rows; lane 63 ends up with the wave's result.  No LDS traffic, no dependent ds_bpermute chain.
template <bool IS_MIN>
__device__ __forceinline__ int wave_minmax(int v)
{
#define OFL_DPP_STEP(CTRL, ROWMASK)                                                        \
    {                                                                                      \
        const int t = __builtin_amdgcn_update_dpp(v, v, CTRL, ROWMASK, 0xf, false);        \
        v = IS_MIN ? min(v, t) : max(v, t);                                                \
    }
    OFL_DPP_STEP(0x111, 0xf)    // row_shr:1
    OFL_DPP_STEP(0x112, 0xf)    // row_shr:2
    OFL_DPP_STEP(0x114, 0xf)    // row_shr:4
    OFL_DPP_STEP(0x118, 0xf)    // row_shr:8
    OFL_DPP_STEP(0x142, 0xa)    // row_bcast:15 into rows 1 and 3
    OFL_DPP_STEP(0x143, 0xc)    // row_bcast:31 into rows 2 and 3
#undef OFL_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}
#endif  // OFL_EXPERIMENTS

// ---------------------------------------------------------------------------------------------------------
// K2, transposed-gather form (variant 3).  The streaming layout (a wave = 2 rows x 128 px) is what the stream loads
// and stores want, but when the sampling grid is ROTATED each gather instruction of such a wave touches 60+ cache
// lines (one per source row its 128-px segment crosses): the texture cache's tag look-ups, not HBM, bound the
// kernel (rocprofv3: 2.75 x the cache accesses of the axis-aligned case).  Workgroups that see such a field hand
// their sampling vectors through LDS to a BLOCK layout (a wave = 32 px x 8 rows, every gather instruction one 8 x 8
// block of it), gather and blend there (~15 lines per instruction), and hand the results back for the streaming stores.  Same arithmetic
// per pixel, bit-identical results; axis-aligned fields keep the direct path.
constexpr int kXpRowF2 = 128 + 4;             // LDS row stride in float2 (padding spreads the 8 rows over the banks)
constexpr int kXposeRows = 6;                 // source rows one streamed 128-px segment may cross before we transpose
constexpr int kC3XcdRows = 1;                 // vertically adjacent tiles one XCD owns per dispatch group (1 = natural tile order)

template <int QUANT, bool STATS, bool BITS = false>
__device__ __forceinline__ void c3_sample_block(const C3Args &a, const float *__restrict__ fa, const uint8_t *__restrict__ ma,
                                                const int (&gx)[4], int gy, const float (&fu)[4], const float (&fv)[4],
                                                float (&su)[4], float (&sv)[4], bool (&ok)[4], C3Stat &st)
{
    const int H = a.H, W = a.W;
    const int mrow = a.mwpr * 4;              // BITS: `ma` is the field's bit plane, rows of mwpr words
    C3Pos tp[4];
    bool act[4], inside = true, outside = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        act[j] = gy < H && gx[j] < W;
        tp[j] = c3_pos<QUANT>(gx[j], gy, fu[j], fv[j], a.sign);
        const bool in_j  = (unsigned)tp[j].ix <= (unsigned)(W - 2) && (unsigned)tp[j].iy <= (unsigned)(H - 2);
        const bool out_j = tp[j].ix < -1 || tp[j].ix >= W || tp[j].iy < -1 || tp[j].iy >= H;
        inside  = inside && (in_j || !act[j]);
        outside = outside && (out_j || !act[j]);
        su[j] = 0.0f; sv[j] = 0.0f; ok[j] = false;
    }
    if (H < 2) inside = false;
    if (__all(outside)) return;
    if (__all(inside)) {
        Pair2    p0[4], p1[4];
        uint32_t m0[4], m1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t s0 = act[j] ? (size_t)tp[j].iy * W + tp[j].ix : 0;
            p0[j] = *reinterpret_cast<const Pair2 *>(fa + 2 * s0);
            p1[j] = *reinterpret_cast<const Pair2 *>(fa + 2 * (s0 + W));
            if (BITS) {
                const uint8_t *r0 = ma + (size_t)(act[j] ? tp[j].iy : 0) * mrow;
                const int ix = act[j] ? tp[j].ix : 0;
                m0[j] = c3_bits_at(r0, ix);
                m1[j] = c3_bits_at(r0 + mrow, ix);
            } else {
                m0[j] = (uint32_t)ma[s0] | ((uint32_t)ma[s0 + 1] << 8);
                m1[j] = (uint32_t)ma[s0 + W] | ((uint32_t)ma[s0 + W + 1] << 8);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const C3Tap w = c3_weights<QUANT>(tp[j]);
            su[j] = c3_blend(p0[j].lo_u, p0[j].hi_u, p1[j].lo_u, p1[j].hi_u, w);
            sv[j] = c3_blend(p0[j].lo_v, p0[j].hi_v, p1[j].lo_v, p1[j].hi_v, w);
            const bool m00 = (m0[j] & (BITS ? 1u : 0xffu)) != 0, m01 = (m0[j] & (BITS ? 2u : 0xff00u)) != 0;
            const bool m10 = (m1[j] & (BITS ? 1u : 0xffu)) != 0, m11 = (m1[j] & (BITS ? 2u : 0xff00u)) != 0;
            ok[j] = c3_valid<QUANT>(m00, m01, m10, m11, w);
            if (STATS) st.amax_m = fmaxf(st.amax_m, (m00 && act[j]) ? fmaxf(fabsf(p0[j].lo_u), fabsf(p0[j].lo_v)) : 0.0f);
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {       // border path, as in c3_tile
        const int  ixc = min(max(tp[j].ix, 0), max(W - 2, 0));
        const int  d   = tp[j].ix - ixc;
        const bool r0  = (unsigned)tp[j].iy < (unsigned)H;
        const bool r1  = (unsigned)(tp[j].iy + 1) < (unsigned)H;
        const int  y0c = min(max(tp[j].iy, 0), H - 1);
        const int  y1c = min(max(tp[j].iy + 1, 0), H - 1);
        const size_t s0 = (size_t)y0c * W + ixc, s1 = (size_t)y1c * W + ixc;
        const Pair2 p0 = *reinterpret_cast<const Pair2 *>(fa + 2 * s0);
        const Pair2 p1 = *reinterpret_cast<const Pair2 *>(fa + 2 * s1);
        uint32_t q00, q01, q10, q11;
        if (BITS) {
            const uint8_t *r0p = ma + (size_t)y0c * mrow, *r1p = ma + (size_t)y1c * mrow;
            q00 = c3_bit(r0p, ixc); q01 = c3_bit(r0p, min(ixc + 1, W - 1)); q10 = c3_bit(r1p, ixc); q11 = c3_bit(r1p, min(ixc + 1, W - 1));
        } else { q00 = ma[s0]; q01 = ma[s0 + 1]; q10 = ma[s1]; q11 = ma[s1 + 1]; }
        float u00, v00, u01, v01, u10, v10, u11, v11, a00, a01, a10, a11;
        select_pair(p0, d, r0, u00, v00, u01, v01);
        select_pair(p1, d, r1, u10, v10, u11, v11);
        select_mask(q00, q01, d, r0, a00, a01);
        select_mask(q10, q11, d, r1, a10, a11);
        const C3Tap w = c3_weights<QUANT>(tp[j]);
        su[j] = c3_blend(u00, u01, u10, u11, w);
        sv[j] = c3_blend(v00, v01, v10, v11, w);
        ok[j] = c3_valid<QUANT>(a00 != 0.0f, a01 != 0.0f, a10 != 0.0f, a11 != 0.0f, w);
        if (STATS) st.amax_m = fmaxf(st.amax_m, (a00 != 0.0f && act[j]) ? fmaxf(fabsf(u00), fabsf(v00)) : 0.0f);
    }
}

#ifndef OFL_C3_BITS_WAVES
#define OFL_C3_BITS_WAVES 5          // the packed-mask form needs a few registers more than the 6-wave budget leaves (11 spilled there)
#endif
template <int QUANT, bool STATS, bool BITS = false>      // BITS: the three masks are packed bit planes (ofl_compose3_bits_dev)
__global__ __launch_bounds__(256, BITS ? OFL_C3_BITS_WAVES : OFL_C3_ONESHOT_WAVES)
void compose3_xpose_kernel(const C3Args a)
{
    static_assert(kC3LanesX == 32, "the transposed form assumes 128 x 8 tiles");
    __shared__ __attribute__((aligned(16))) float2 xp_v[8 * kXpRowF2];
    __shared__ __attribute__((aligned(16))) uint8_t xp_ok[8 * 128];
    // Block -> tile.  Workgroups are dealt round-robin over the 8 XCDs (block b runs on XCD b % 8), each with its own L2.
    // In natural tile order the tile BELOW a tile is tiles_x blocks later and on another XCD, so the source rows the two
    // share (bilinear halo; for a rotated sampling grid most of their cache lines) are fetched once per XCD.  With
    // swz_group = R > 1 a group of 8 R consecutive blocks covers 8 tile columns x R tile rows and XCD x owns column x of
    // it: R vertically adjacent tiles, dispatched 8 blocks apart, meet in ONE L2 -- while all XCDs still sweep the same
    // window of memory (whole-field contiguous runs per XCD were measured slower).
    int tile = blockIdx.x;
    if (a.swz_group > 1) {
        const int R = a.swz_group, per = 8 * R;
        const int g = blockIdx.x / per, i = blockIdx.x - g * per;
        const int tiles_y = a.tiles_per_field / a.tiles_x, n_gx = (a.tiles_x + 7) >> 3, n_gy = (tiles_y + R - 1) / R;
        const int fld = g / (n_gx * n_gy), gg = g - fld * (n_gx * n_gy), gy = gg / n_gx, gx = gg - gy * n_gx;
        const int ty = gy * R + (i >> 3), tx = gx * 8 + (i & 7);
        if (tx >= a.tiles_x || ty >= tiles_y) return;
        tile = fld * a.tiles_per_field + ty * a.tiles_x + tx;
    }
    if (tile >= a.ntiles) return;
    C3Stat st = { 0.0f, 0.0f, 0.0f };
    const C3Stream in = c3_load_stream<kC3LanesX, BITS>(a, tile);
    int b, y, xg[2];
    c3_tile_coords<kC3LanesX>(a, tile, b, y, xg);
    const int H = a.H, W = a.W;
    const bool act[2] = { y < H && xg[0] < W, y < H && xg[1] < W };
    // Does a streamed row segment of this tile cross many source rows (sample row = y -/+ v)?  Every wave answers from
    // the SAME two values -- the vertical flow at the two ends of the tile's first row (uniform loads, L2 hits) -- so
    // the workgroup agrees without a barrier and the direct path costs nothing extra.
    const int t   = tile - b * a.tiles_per_field;
    const int ty  = t / a.tiles_x, tx = t - ty * a.tiles_x;
    bool rotated;
    {
        const int x0 = tx * 128, x1 = min(x0 + 127, W - 1), yy = min(ty * 8, H - 1);
        const float *frow = a.fb + 2 * ((size_t)b * H * W + (size_t)yy * W);
        rotated = fabsf(frow[2 * x1 + 1] - frow[2 * x0 + 1]) * 128.0f > (float)(a.xpose_rows * (x1 - x0 + 1));
    }
    if (!rotated) {
        c3_tile<QUANT, STATS, BITS>(a, tile, in, st);
        if (STATS) c3_flush_stats(a, tile / a.tiles_per_field, st);
        return;
    }
    const int ly = threadIdx.x >> 5, lx = threadIdx.x & 31;
    // streaming layout -> LDS: the sampling vectors of this lane's two pixel pairs
    *reinterpret_cast<float4 *>(&xp_v[ly * kXpRowF2 + 2 * lx])      = in.v[0];
    *reinterpret_cast<float4 *>(&xp_v[ly * kXpRowF2 + 64 + 2 * lx]) = in.v[1];
    __syncthreads();
    // block layout: wave w owns columns [32 w, 32 w + 32) of all 8 rows, a lane four adjacent pixels
    // every gather INSTRUCTION (fixed j) then covers a compact 8 x 8 block: pixel j of lane (r, c) is column 8 j + c
    const int w4  = threadIdx.x >> 6, l = threadIdx.x & 63, r = l >> 3, c0 = 32 * w4 + (l & 7);
    {
        float fu[4], fv[4];
        int   gx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float2 f = xp_v[r * kXpRowF2 + c0 + 8 * j];
            fu[j] = f.x; fv[j] = f.y;
            gx[j] = tx * 128 + c0 + 8 * j;
        }
        float su[4], sv[4];
        bool  ok[4];
        const size_t field = (size_t)b * H * W;
        c3_sample_block<QUANT, STATS, BITS>(a, a.fa + field * 2, BITS ? a.ma + (size_t)b * H * a.mwpr * 4 : a.ma + field, gx, ty * 8 + r, fu, fv, su, sv, ok, st);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xp_v[r * kXpRowF2 + c0 + 8 * j] = make_float2(su[j], sv[j]);
            xp_ok[r * 128 + c0 + 8 * j] = ok[j] ? 1 : 0;
        }
    }
    __syncthreads();
    // back in the streaming layout: add, combine the masks, store
    {
        const float4 s01 = *reinterpret_cast<const float4 *>(&xp_v[ly * kXpRowF2 + 2 * lx]);
        const float4 s23 = *reinterpret_cast<const float4 *>(&xp_v[ly * kXpRowF2 + 64 + 2 * lx]);
        const uint32_t k01 = *reinterpret_cast<const uint16_t *>(&xp_ok[ly * 128 + 2 * lx]);
        const uint32_t k23 = *reinterpret_cast<const uint16_t *>(&xp_ok[ly * 128 + 64 + 2 * lx]);
        const float su[kC3Px] = { s01.x, s01.z, s23.x, s23.z }, sv[kC3Px] = { s01.y, s01.w, s23.y, s23.w };
        const bool  ok[kC3Px] = { (k01 & 0xffu) != 0, (k01 & 0xff00u) != 0, (k23 & 0xffu) != 0, (k23 & 0xff00u) != 0 };
        const float bu[kC3Px] = { in.v[0].x, in.v[0].z, in.v[1].x, in.v[1].z };
        const float bv[kC3Px] = { in.v[0].y, in.v[0].w, in.v[1].y, in.v[1].w };
        const bool  bm[kC3Px] = { (in.m[0] & 0xffu) != 0, (in.m[0] & 0xff00u) != 0, (in.m[1] & 0xffu) != 0, (in.m[1] & 0xff00u) != 0 };
        const size_t row = (size_t)b * H * W + (size_t)y * W;
        c3_finish<STATS, BITS>(a, row, xg, act, bu, bv, bm, su, sv, ok, st);
    }
    if (STATS) c3_flush_stats(a, tile / a.tiles_per_field, st);
}

#ifdef OFL_EXPERIMENTS
template <int QUANT, bool STATS>
__device__ __forceinline__ void c3_tile_lds(const C3Args &a, int tile, const C3Stream &in, C3Stat &st,
                                            float4 *lds, int *box_now, int *box_next)
{
    int b, y, xg[2];
    c3_tile_coords<kLdsLX>(a, tile, b, y, xg);
    const int H = a.H, W = a.W, sign = a.sign;
    const size_t field = (size_t)b * H * W;
    const float   *fa = a.fa + field * 2;
    const uint8_t *ma = a.ma + field;
    const bool act[2] = { y < H && xg[0] < W, y < H && xg[1] < W };
    const size_t row = field + (size_t)y * W;

    const float bu[kC3Px] = { in.v[0].x, in.v[0].z, in.v[1].x, in.v[1].z };
    const float bv[kC3Px] = { in.v[0].y, in.v[0].w, in.v[1].y, in.v[1].w };
    const bool  bm[kC3Px] = { (in.m[0] & 0xffu) != 0, (in.m[0] & 0xff00u) != 0,
                              (in.m[1] & 0xffu) != 0, (in.m[1] & 0xff00u) != 0 };

    // ---- sample positions and this lane's share of the footprint
    C3Pos tp[kC3Px];
    bool  use[kC3Px];
    int bx0 = 0x7fffffff, by0 = 0x7fffffff, bx1 = -0x7fffffff, by1 = -0x7fffffff;
#pragma unroll
    for (int j = 0; j < kC3Px; ++j) {
        tp[j] = c3_pos<QUANT>(xg[j >> 1] + (j & 1), y, bu[j], bv[j], sign);
        const bool out_j = tp[j].ix < -1 || tp[j].ix >= W || tp[j].iy < -1 || tp[j].iy >= H;
        use[j] = act[j >> 1] && !out_j && !OFL_ABLATE(a, 1);
        if (use[j]) {
            bx0 = min(bx0, max(tp[j].ix, 0));     bx1 = max(bx1, min(tp[j].ix + 1, W - 1));
            by0 = min(by0, max(tp[j].iy, 0));     by1 = max(by1, min(tp[j].iy + 1, H - 1));
        }
    }
    bx0 = wave_minmax<true>(bx0);  by0 = wave_minmax<true>(by0);
    bx1 = wave_minmax<false>(bx1); by1 = wave_minmax<false>(by1);
    if ((threadIdx.x & 63) == 0 && bx1 >= bx0) {
        atomicMin(&box_now[0], bx0); atomicMin(&box_now[1], by0);
        atomicMax(&box_now[2], bx1); atomicMax(&box_now[3], by1);
    }
    __syncthreads();                                                   // A: footprint complete
    const int X0 = box_now[0] & ~1, Y0 = box_now[1], X1 = box_now[2], Y1 = box_now[3];
    if (threadIdx.x == 0) {                                            // re-arm the other box for the next tile
        box_next[0] = 0x7fffffff; box_next[1] = 0x7fffffff; box_next[2] = -0x7fffffff; box_next[3] = -0x7fffffff;
    }
    const bool any    = X1 >= X0 && Y1 >= Y0;
    const int  bw     = any ? ((X1 - X0 + 2) & ~1) : 0;                // even number of columns covering X0..X1
    const int  bh     = any ? (Y1 - Y0 + 1) : 0;
    const bool staged = any && bw * bh <= kLdsCap && !OFL_ABLATE(a, 4);

    if (staged) {
        const int pairs = bw >> 1, total = pairs * bh;
        for (int idx = threadIdx.x; idx < total; idx += 256) {
            const int r = idx / pairs, cp = idx - r * pairs;
            const size_t g = (size_t)(Y0 + r) * W + (X0 + 2 * cp);
            const float4   v = *reinterpret_cast<const float4 *>(fa + 2 * g);
            const uint32_t m = *reinterpret_cast<const uint16_t *>(ma + g);
            lds[r * bw + 2 * cp]     = make_float4(v.x, v.y, (m & 0xffu) ? 1.0f : 0.0f, 0.0f);
            lds[r * bw + 2 * cp + 1] = make_float4(v.z, v.w, (m & 0xff00u) ? 1.0f : 0.0f, 0.0f);
        }
    }
    __syncthreads();                                                   // B: tile staged (and box_next re-armed)

    float su[kC3Px], sv[kC3Px];
    bool  ok[kC3Px];
#pragma unroll
    for (int j = 0; j < kC3Px; ++j) { su[j] = 0.0f; sv[j] = 0.0f; ok[j] = false; }

    if (staged && !OFL_ABLATE(a, 8)) {
#pragma unroll
        for (int j = 0; j < kC3Px; ++j) {
            if (!use[j]) continue;
            const int ix = tp[j].ix, iy = tp[j].iy;
            const bool cx0 = ix >= 0, cx1 = ix + 1 <= W - 1, cy0 = iy >= 0, cy1 = iy + 1 <= H - 1;
            const int lx0 = max(ix, 0) - X0, lx1 = min(ix + 1, W - 1) - X0;
            const int ly0 = max(iy, 0) - Y0, ly1 = min(iy + 1, H - 1) - Y0;
            float4 t00 = lds[ly0 * bw + lx0], t01 = lds[ly0 * bw + lx1];
            float4 t10 = lds[ly1 * bw + lx0], t11 = lds[ly1 * bw + lx1];
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!(cx0 && cy0)) t00 = z;                                // taps outside the image: BORDER_CONSTANT 0
            if (!(cx1 && cy0)) t01 = z;
            if (!(cx0 && cy1)) t10 = z;
            if (!(cx1 && cy1)) t11 = z;
            const C3Tap w = c3_weights<QUANT>(tp[j]);
            su[j] = c3_blend(t00.x, t01.x, t10.x, t11.x, w);
            sv[j] = c3_blend(t00.y, t01.y, t10.y, t11.y, w);
            ok[j] = c3_valid<QUANT>(t00.z != 0.0f, t01.z != 0.0f, t10.z != 0.0f, t11.z != 0.0f, w);
            if (STATS) st.amax_m = fmaxf(st.amax_m, t00.z != 0.0f ? fmaxf(fabsf(t00.x), fabsf(t00.y)) : 0.0f);
        }
    } else if (any) {
        // footprint too large for the LDS budget: per-pixel gathers straight from global memory
#pragma unroll
        for (int j = 0; j < kC3Px; ++j) {
            if (!use[j]) continue;
            const int  ixc = min(max(tp[j].ix, 0), max(W - 2, 0));
            const int  d   = tp[j].ix - ixc;
            const bool r0  = (unsigned)tp[j].iy < (unsigned)H;
            const bool r1  = (unsigned)(tp[j].iy + 1) < (unsigned)H;
            const int  y0c = min(max(tp[j].iy, 0), H - 1);
            const int  y1c = min(max(tp[j].iy + 1, 0), H - 1);
            const size_t s0 = (size_t)y0c * W + ixc, s1 = (size_t)y1c * W + ixc;
            const Pair2 p0 = *reinterpret_cast<const Pair2 *>(fa + 2 * s0);
            const Pair2 p1 = *reinterpret_cast<const Pair2 *>(fa + 2 * s1);
            const uint32_t q00 = ma[s0], q01 = ma[s0 + 1], q10 = ma[s1], q11 = ma[s1 + 1];
            float u00, v00, u01, v01, u10, v10, u11, v11, a00, a01, a10, a11;
            select_pair(p0, d, r0, u00, v00, u01, v01);
            select_pair(p1, d, r1, u10, v10, u11, v11);
            select_mask(q00, q01, d, r0, a00, a01);
            select_mask(q10, q11, d, r1, a10, a11);
            const C3Tap w = c3_weights<QUANT>(tp[j]);
            su[j] = c3_blend(u00, u01, u10, u11, w);
            sv[j] = c3_blend(v00, v01, v10, v11, w);
            ok[j] = c3_valid<QUANT>(a00 != 0.0f, a01 != 0.0f, a10 != 0.0f, a11 != 0.0f, w);
            if (STATS) st.amax_m = fmaxf(st.amax_m, a00 != 0.0f ? fmaxf(fabsf(u00), fabsf(v00)) : 0.0f);
        }
    }
    c3_finish<STATS>(a, row, xg, act, bu, bv, bm, su, sv, ok, st);
}

template <int QUANT, bool STATS>
__global__ __launch_bounds__(256)
void compose3_lds_kernel(const C3Args a)
{
    __shared__ float4 lds[kLdsCap];
    __shared__ int    box[2][4];
    if (threadIdx.x < 8) box[threadIdx.x >> 2][threadIdx.x & 3] = (threadIdx.x & 2) ? -0x7fffffff : 0x7fffffff;
    __syncthreads();
    const int nb  = gridDim.x;
    const int per = nb >> 3;
    int tile = (nb & 7) == 0 ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= a.ntiles) return;
    C3Stat st = { 0.0f, 0.0f, 0.0f };
    int cur_b = tile / a.tiles_per_field;
    int parity = 0;
    C3Stream in = c3_load_stream<kLdsLX>(a, tile);
    while (true) {
        const int next = tile + nb;
        const bool more = next < a.ntiles;
        C3Stream nxt;
        if (more) nxt = c3_load_stream<kLdsLX>(a, next);     // prefetch across this tile's two barriers
        c3_tile_lds<QUANT, STATS>(a, tile, in, st, lds, box[parity], box[parity ^ 1]);
        if (!more) break;
        if (STATS) {
            const int nb_ = next / a.tiles_per_field;
            if (nb_ != cur_b) {
                c3_flush_stats(a, cur_b, st);
                st.amax_m = st.bmax = st.bmax_m = 0.0f;
                cur_b = nb_;
            }
        }
        in = nxt;
        tile = next;
        parity ^= 1;
    }
    if (STATS) c3_flush_stats(a, cur_b, st);
}
#endif  // OFL_EXPERIMENTS

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

// Batched launches of K1 (ofl_gather_bilinear_batch_dev): blockIdx.y = the field of the batch; element strides from one field's
// arrays to the next (0: the array is shared by the whole batch -- one source image warped by B flows; or absent).
struct GBatch { size_t src, smask, flow, fmask, dst, valid; };

template <typename T, int CT>
__global__ __launch_bounds__(256)
void gather_kernel(const T *__restrict__ src0, int Crt, int H, int W,
                   const float *__restrict__ flow0, int fH, int fW, int pad_top, int pad_left, int sign,
                   const uint8_t *__restrict__ smask0, const uint8_t *__restrict__ fmask0,
                   T *__restrict__ dst0, uint8_t *__restrict__ valid0,
                   int quant, int arith, int rule, int tiles_x, int nblocks, int row0, int rows, GBatch bs)
{
    const size_t bi = blockIdx.y;
    const T *__restrict__ src = src0 + bi * bs.src;
    const float *__restrict__ flow = flow0 + bi * bs.flow;
    const uint8_t *__restrict__ smask = smask0 ? smask0 + bi * bs.smask : nullptr;
    const uint8_t *__restrict__ fmask = fmask0 ? fmask0 + bi * bs.fmask : nullptr;
    T *__restrict__ dst = dst0 + bi * bs.dst;
    uint8_t *__restrict__ valid = valid0 ? valid0 + bi * bs.valid : nullptr;
    const int C    = CT > 0 ? CT : Crt;
    const int tile = xcd_swizzle(blockIdx.x, nblocks);
    const int ty   = tile / tiles_x, tx = tile - ty * tiles_x;
    const int x    = tx * 32 + (threadIdx.x & 31);
    const int yl   = ty * 8 + (threadIdx.x >> 5);           // row inside the output band [row0, row0 + rows)
    const int y    = row0 + yl;
    if (x >= W || yl >= rows) return;

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
    const size_t o = (size_t)yl * W + x;
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

// K1, paired form (W even, C = 1..4): same lane mapping as K2 -- 128 x 8 tiles, each lane owns the pixel
// pairs at x = 2*lx and x = 64 + 2*lx, so flow loads (16 B), image stores (2*C elements) and validity stores
// (2 B) of a wave are contiguous -- and the same three wave-uniform paths (all samples outside / all
// inside / border).  Numerics are those of gather_kernel (and of the oracle), element for element.
// BYTES (2 .. 16) consecutive bytes from a possibly unaligned address into 32-bit words
template <int BYTES>
__device__ __forceinline__ void load_run(const uint8_t *p, uint32_t (&w)[4])
{
    if constexpr (BYTES == 2) {
        w[0] = *reinterpret_cast<const uint16_t *>(p);
    } else if constexpr (BYTES == 4) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
    } else if constexpr (BYTES == 6) {
        w[0] = *reinterpret_cast<const uint32_t *>(p);
        w[1] = *reinterpret_cast<const uint16_t *>(p + 4);
    } else if constexpr (BYTES == 8) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        w[0] = t.x; w[1] = t.y;
    } else if constexpr (BYTES == 12) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        w[0] = t.x; w[1] = t.y;
        w[2] = *reinterpret_cast<const uint32_t *>(p + 8);
    } else {
        static_assert(BYTES == 16, "load_run: unsupported run length");
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    }
}

template <typename T>
__device__ __forceinline__ T run_elem(const uint32_t (&w)[4], int i)      // i is a compile-time constant after unrolling
{
    if constexpr (sizeof(T) == 1)      return (T)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
    else if constexpr (sizeof(T) == 2) return (T)((w[i >> 1] >> (16 * (i & 1))) & 0xffffu);
    else                               return (T)__uint_as_float(w[i]);
}

template <typename T, int CT, bool INSIDE>
__device__ __forceinline__ void gather_px(const T *__restrict__ src, const uint8_t *__restrict__ smask, int H, int W,
                                          const Tap &tp, const int (&wi)[4], bool fixed_u8, bool sep, int arith, int rule,
                                          bool want_valid, T (&res)[CT], bool &ok)
{
    bool     in[4];
    uint32_t off[4];          // 32-bit offsets (the paired kernel takes images below 4 GiB): scalar base + 32-bit lane offset, no 64-bit address math
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int yy = tp.iy + (k >> 1), xx = tp.ix + (k & 1);
        in[k]  = INSIDE ? true : ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W);
        off[k] = in[k] ? __umul24((uint32_t)yy, (uint32_t)W) + (uint32_t)xx : 0u;
    }
    // the mask taps are requested BEFORE the image taps (their latency hides behind the image loads); inside the
    // image the two taps of a row are one 2-byte load (unaligned addresses are fine for global loads on gfx950)
    int m[4] = { 0, 0, 0, 0 };
    if (want_valid) {
        if (!smask) {
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = in[k] ? 1 : 0;
        } else if (INSIDE) {
            const uint32_t r0 = *reinterpret_cast<const uint16_t *>(smask + off[0]), r1 = *reinterpret_cast<const uint16_t *>(smask + off[2]);
            m[0] = (r0 & 0xffu) != 0; m[1] = (r0 & 0xff00u) != 0; m[2] = (r1 & 0xffu) != 0; m[3] = (r1 & 0xff00u) != 0;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) m[k] = in[k] ? (smask[off[k]] != 0) : 0;
        }
    }
    typedef typename Acc<T>::type A;
    // 8- and 16-bit images inside the source: the two taps of a row are 2 * CT adjacent elements = 2 .. 16 bytes,
    // fetched as one or two wide loads per row instead of 2 * CT element loads (unaligned global loads are fine)
    constexpr bool kRun = INSIDE && (sizeof(T) <= 2 || (sizeof(T) == 4 && CT <= 2));     // float: 1 or 2 channels = 8 / 16 bytes per row
    uint32_t run[2][4] = { { 0u, 0u, 0u, 0u }, { 0u, 0u, 0u, 0u } };
    if constexpr (kRun) {
        constexpr int kRB = 2 * CT * (int)sizeof(T);
        // 6- and 12-byte runs are read as ONE 8- / 16-byte load (two or four bytes too many) wherever that stays inside the
        // image -- everywhere but the last pixels of the last row
        const uint32_t total = (uint32_t)H * (uint32_t)W * (uint32_t)(CT * sizeof(T));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t at = off[2 * r] * (uint32_t)(CT * sizeof(T));
            const uint8_t *pr = reinterpret_cast<const uint8_t *>(src) + at;
            if constexpr (kRB == 6 || kRB == 12) {
                if (at + kRB + kRB / 3 <= total) load_run<kRB + kRB / 3>(pr, run[r]);
                else                             load_run<kRB>(pr, run[r]);
            } else load_run<kRB>(pr, run[r]);
        }
    }
    // uint8 images with cv2's 15-bit fixed-point weights: w_k = 32 (32 - ay | ay) (32 - ax | ax), so
    //   (sum v_k w_k + 2^14) >> 15  ==  ((32 - ay) h0 + ay h1 + 512) >> 10   with   h_r = (32 - ax) v_r0 + ax v_r1
    // exactly (32 S + 16384 = 32 (S + 512)); the weights sum to 1024, so the result needs no clamp.  Per channel: one
    // v_perm_b32 per row puts the two taps side by side, one v_dot4_u32_u8 blends them horizontally, one multiply-add
    // vertically -- about half the integer work of the four-tap form (which stays for the un-snapped weights).
    bool done_sep = false;
    if constexpr (kRun && sizeof(T) == 1) {
        if (fixed_u8 && sep) {
            const uint32_t wx = (uint32_t)(32 - tp.ax) | ((uint32_t)tp.ax << 8);
            const int wy0 = 32 - tp.ay, wy1 = tp.ay;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                constexpr uint32_t kZero = 0x0C0C0000u;                       // selector bytes 2, 3: constant 0
                const uint32_t sel = kZero | (uint32_t)c | ((uint32_t)(CT + c) << 8);
                const uint32_t t0 = __builtin_amdgcn_perm(run[0][1], run[0][0], sel), t1 = __builtin_amdgcn_perm(run[1][1], run[1][0], sel);
                const uint32_t h0 = __builtin_amdgcn_udot4(t0, wx, 0u, false), h1 = __builtin_amdgcn_udot4(t1, wx, 0u, false);
                res[c] = (T)((__umul24(h0, (uint32_t)wy0) + __umul24(h1, (uint32_t)wy1) + 512u) >> 10);
            }
            done_sep = true;
        }
    }
    if (!done_sep) {
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        A v[4];
        if constexpr (kRun) {
            v[0] = (A)run_elem<T>(run[0], c); v[1] = (A)run_elem<T>(run[0], CT + c);
            v[2] = (A)run_elem<T>(run[1], c); v[3] = (A)run_elem<T>(run[1], CT + c);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = in[k] ? (A)(src + off[k] * (uint32_t)CT)[c] : (A)0;      // (one zero-extended base per tap: the channel loads still merge)
        }
        if constexpr (sizeof(T) == 8) {
            res[c] = (T)blend4d(v[0], v[1], v[2], v[3], tp);
        } else {
            if (fixed_u8) {
                const int acc = __mul24((int)v[0], wi[0]) + __mul24((int)v[1], wi[1]) + __mul24((int)v[2], wi[2]) + __mul24((int)v[3], wi[3]);
                res[c] = (T)min(max((acc + (1 << 14)) >> 15, 0), 255);
            } else {
                res[c] = finish<T>(blend4((float)v[0], (float)v[1], (float)v[2], (float)v[3], tp), arith);
            }
        }
    }
    }
    ok = false;
    if (want_valid) {
        if (rule == OFL_RULE_GE_HALF) {
            const int acc = __mul24(m[0], wi[0]) + __mul24(m[1], wi[1]) + __mul24(m[2], wi[2]) + __mul24(m[3], wi[3]);
            ok = ((acc + (1 << 14)) >> 15) == 1;
        } else {
            const float sm = blend4((float)m[0], (float)m[1], (float)m[2], (float)m[3], tp);
            ok = (rule == OFL_RULE_EQ1) ? (sm == 1.0f) : (cv_round(sm) == 1);
        }
    }
}

// ---- the all-inside path in two phases: every pixel's loads are issued before the first pixel is blended ----
// gather_px asks for a pixel's taps and blends them at once; four pixels of a lane one after the other are four dependent
// rounds of loads behind the round of the flow itself (PMC: a wave of the 8-bit kernel lives 11.9 us for 2 000 cycles of
// instructions, 70 % of it waiting; six waves per SIMD cannot hide five rounds of 2.3 us).  For a wave whose taps are all inside
// the source -- nearly every wave -- the loads need no predication: px_load asks for all of them, px_blend is gather_px's
// arithmetic on what they return.  Same values, same operations, same order per pixel: the same bits.
template <typename T, int CT> struct PxRun { static constexpr bool value = sizeof(T) <= 2 || (sizeof(T) == 4 && CT <= 2); };

template <typename T, int CT>
struct PxLoad {
    uint32_t run[2][4];                              // PxRun: the two taps of a row as one run of 2 * CT elements
    typename Acc<T>::type v[PxRun<T, CT>::value ? 1 : 4][PxRun<T, CT>::value ? 1 : CT];      // otherwise: tap by tap
    uint32_t m01, m23;                               // the source mask's two bytes per row
};

template <typename T, int CT, bool WIDE>        // WIDE: 6- and 12-byte runs may be read as 8 / 16 bytes (the caller checked the image's end)
__device__ __forceinline__ void px_load(const T *__restrict__ src, const uint8_t *__restrict__ smask, int W, int ix, int iy,
                                        bool want_valid, PxLoad<T, CT> &L)
{
    const uint32_t off0 = __umul24((uint32_t)iy, (uint32_t)W) + (uint32_t)ix, off2 = off0 + (uint32_t)W;
    L.m01 = L.m23 = 0x0101u;
    if (want_valid && smask) {
        L.m01 = *reinterpret_cast<const uint16_t *>(smask + off0);
        L.m23 = *reinterpret_cast<const uint16_t *>(smask + off2);
    }
    if constexpr (PxRun<T, CT>::value) {
        constexpr int kRB = 2 * CT * (int)sizeof(T);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            L.run[r][0] = L.run[r][1] = L.run[r][2] = L.run[r][3] = 0u;
            const uint8_t *pr = reinterpret_cast<const uint8_t *>(src) + (r ? off2 : off0) * (uint32_t)(CT * sizeof(T));
            if constexpr ((kRB == 6 || kRB == 12) && WIDE) load_run<kRB + kRB / 3>(pr, L.run[r]);
            else load_run<kRB>(pr, L.run[r]);
        }
    } else {
        typedef typename Acc<T>::type A;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const T *t = src + ((k >> 1) ? off2 : off0) * (uint32_t)CT + (uint32_t)((k & 1) * CT);
#pragma unroll
            for (int c = 0; c < CT; ++c) L.v[k][c] = (A)t[c];
        }
    }
}

template <typename T, int CT>
__device__ __forceinline__ void px_blend(const PxLoad<T, CT> &L, const Tap &tp, const int (&wi)[4], bool fixed_u8, bool sep, int arith, int rule,
                                         bool want_valid, T (&res)[CT], bool &ok)
{
    typedef typename Acc<T>::type A;
    const int m[4] = { (L.m01 & 0xffu) != 0, (L.m01 & 0xff00u) != 0, (L.m23 & 0xffu) != 0, (L.m23 & 0xff00u) != 0 };
    bool done_sep = false;
    if constexpr (PxRun<T, CT>::value && sizeof(T) == 1) {
        if (fixed_u8 && sep) {                       // (gather_px: the separable fixed-point form)
            const uint32_t wx = (uint32_t)(32 - tp.ax) | ((uint32_t)tp.ax << 8);
            const int wy0 = 32 - tp.ay, wy1 = tp.ay;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                constexpr uint32_t kZero = 0x0C0C0000u;
                const uint32_t sel = kZero | (uint32_t)c | ((uint32_t)(CT + c) << 8);
                const uint32_t t0 = __builtin_amdgcn_perm(L.run[0][1], L.run[0][0], sel), t1 = __builtin_amdgcn_perm(L.run[1][1], L.run[1][0], sel);
                const uint32_t h0 = __builtin_amdgcn_udot4(t0, wx, 0u, false), h1 = __builtin_amdgcn_udot4(t1, wx, 0u, false);
                res[c] = (T)((__umul24(h0, (uint32_t)wy0) + __umul24(h1, (uint32_t)wy1) + 512u) >> 10);
            }
            done_sep = true;
        }
    }
    if (!done_sep) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            A v[4];
            if constexpr (PxRun<T, CT>::value) {
                v[0] = (A)run_elem<T>(L.run[0], c); v[1] = (A)run_elem<T>(L.run[0], CT + c);
                v[2] = (A)run_elem<T>(L.run[1], c); v[3] = (A)run_elem<T>(L.run[1], CT + c);
            } else {
                v[0] = L.v[0][c]; v[1] = L.v[1][c]; v[2] = L.v[2][c]; v[3] = L.v[3][c];
            }
            if constexpr (sizeof(T) == 8) {
                res[c] = (T)blend4d(v[0], v[1], v[2], v[3], tp);
            } else {
                if (fixed_u8) {
                    const int acc = __mul24((int)v[0], wi[0]) + __mul24((int)v[1], wi[1]) + __mul24((int)v[2], wi[2]) + __mul24((int)v[3], wi[3]);
                    res[c] = (T)min(max((acc + (1 << 14)) >> 15, 0), 255);
                } else {
                    res[c] = finish<T>(blend4((float)v[0], (float)v[1], (float)v[2], (float)v[3], tp), arith);
                }
            }
        }
    }
    ok = false;
    if (want_valid) {
        if (rule == OFL_RULE_GE_HALF) {
            const int acc = __mul24(m[0], wi[0]) + __mul24(m[1], wi[1]) + __mul24(m[2], wi[2]) + __mul24(m[3], wi[3]);
            ok = ((acc + (1 << 14)) >> 15) == 1;
        } else {
            const float sm = blend4((float)m[0], (float)m[1], (float)m[2], (float)m[3], tp);
            ok = (rule == OFL_RULE_EQ1) ? (sm == 1.0f) : (cv_round(sm) == 1);
        }
    }
}

// taps, gather and blend of four pixels (gx[j], y) of one lane; wave-uniform all-outside / all-inside / border paths
template <typename T, int CT>
__device__ __forceinline__ void gather2_core(const T *__restrict__ src, const uint8_t *__restrict__ smask, int H, int W,
                                             const int (&gx)[4], int y, const bool (&act4)[4],
                                             const float (&fu)[4], const float (&fv)[4], int sign,
                                             int quant, int arith, int rule, bool want_valid,
                                             T (&res)[4][CT], bool (&ok)[4])
{
    Tap  tp[4];
    int  wi[4][4];
    bool inside = true, outside = true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // float32(float64(grid) +- float64(flow)) of utils.py:231-235 as ONE float32 operation: exact for integer grid
        // coordinates below 2^15 (proof at c3_pos; the float64 form costs a dozen half-rate instructions per pixel)
        const float px = sign >= 0 ? __fadd_rn((float)gx[j], fu[j]) : __fsub_rn((float)gx[j], fu[j]);
        const float py = sign >= 0 ? __fadd_rn((float)y, fv[j]) : __fsub_rn((float)y, fv[j]);
        tp[j] = (quant == OFL_QUANT_OPENCV) ? make_tap<OFL_QUANT_OPENCV>(px, py) : make_tap<OFL_QUANT_EXACT>(px, py);
        if (quant == OFL_QUANT_OPENCV) {
            wi[j][0] = __mul24(32 - tp[j].ay, 32 - tp[j].ax) * 32; wi[j][1] = __mul24(32 - tp[j].ay, tp[j].ax) * 32;
            wi[j][2] = __mul24(tp[j].ay, 32 - tp[j].ax) * 32;      wi[j][3] = __mul24(tp[j].ay, tp[j].ax) * 32;
        } else {
            wi[j][0] = __float2int_rn(tp[j].w0 * 32768.0f); wi[j][1] = __float2int_rn(tp[j].w1 * 32768.0f);
            wi[j][2] = __float2int_rn(tp[j].w2 * 32768.0f); wi[j][3] = __float2int_rn(tp[j].w3 * 32768.0f);
        }
        const bool in_j  = (unsigned)tp[j].ix <= (unsigned)(W - 2) && (unsigned)tp[j].iy <= (unsigned)(H - 2);
        const bool out_j = tp[j].ix < -1 || tp[j].ix >= W || tp[j].iy < -1 || tp[j].iy >= H;
        inside  = inside && (in_j || !act4[j]);
        outside = outside && (out_j || !act4[j]);
    }
    if (H < 2) inside = false;
    const bool fixed_u8 = (sizeof(T) == 1) && arith == OFL_ARITH_NATIVE;

#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ok[j] = false;
#pragma unroll
        for (int c = 0; c < CT; ++c) res[j][c] = (T)0;
    }
    const bool all_outside = __all(outside), all_inside = !all_outside && __all(inside);
    if (all_outside) {
        // nothing to fetch: every tap of every pixel of this wave is outside the source
    } else if (all_inside && sizeof(T) < 8) {
        // (float64 images keep the one-pixel-at-a-time form: their taps alone are 4 * CT * 2 registers per pixel)
        constexpr int kRB = 2 * CT * (int)sizeof(T);
        constexpr bool kPadded = PxRun<T, CT>::value && (kRB == 6 || kRB == 12);
        bool wide = true;                            // may every 6- / 12-byte run of this lane be read as 8 / 16 bytes?
        if constexpr (kPadded) {
            const uint32_t total = (uint32_t)H * (uint32_t)W * (uint32_t)(CT * sizeof(T));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (act4[j]) wide = wide && (__umul24((uint32_t)(tp[j].iy + 1), (uint32_t)W) + (uint32_t)tp[j].ix) * (uint32_t)(CT * sizeof(T)) + kRB + kRB / 3 <= total;
        }
        // pixels in flight together: all four when a pixel's taps are runs (<= 8 registers), two otherwise (float images of 3 / 4
        // channels: 12 / 16 registers per pixel)
        constexpr int kGroup = PxRun<T, CT>::value ? 4 : 2;
        const bool all_wide = __all(wide);
        int lix[4], liy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { lix[j] = act4[j] ? tp[j].ix : 0; liy[j] = act4[j] ? tp[j].iy : 0; }      // (a pixel off the frame reads the image's first taps)
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += kGroup) {
            PxLoad<T, CT> L[kGroup];
            if (all_wide) {                          // (the branch around the whole group, not inside it: each load phase is straight-line code)
#pragma unroll
                for (int j = 0; j < kGroup; ++j) px_load<T, CT, true>(src, smask, W, lix[j0 + j], liy[j0 + j], want_valid, L[j]);
            } else {
#pragma unroll
                for (int j = 0; j < kGroup; ++j) px_load<T, CT, false>(src, smask, W, lix[j0 + j], liy[j0 + j], want_valid, L[j]);
            }
#pragma unroll
            for (int j = 0; j < kGroup; ++j)
                if (act4[j0 + j]) px_blend<T, CT>(L[j], tp[j0 + j], wi[j0 + j], fixed_u8, quant == OFL_QUANT_OPENCV, arith, rule, want_valid, res[j0 + j], ok[j0 + j]);
        }
    } else if (all_inside) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (act4[j]) gather_px<T, CT, true>(src, smask, H, W, tp[j], wi[j], fixed_u8, quant == OFL_QUANT_OPENCV, arith, rule, want_valid, res[j], ok[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (act4[j]) gather_px<T, CT, false>(src, smask, H, W, tp[j], wi[j], fixed_u8, quant == OFL_QUANT_OPENCV, arith, rule, want_valid, res[j], ok[j]);
    }

}

#ifndef OFL_G2_NT_FLOW
#define OFL_G2_NT_FLOW 0       // measured: non-temporal flow loads / image stores do not help this kernel (they do help K2)
#endif
#ifndef OFL_G2_NT_DST
#define OFL_G2_NT_DST 0
#endif
// SPEC: the (quant, arith, rule) triple as compile-time constants for the combinations the Flow algebra actually asks for --
// 1 = cv2's 1/32-px snap + uint8 fixed point + `>= 1/2` (a uint8 image with a boolean mask, flow_class.py:644), 2 = the snap +
// float accumulate + `== 1` (float images and Flow targets), 3 = the snap + float sum rounded half to even + `> 1/2` (a uint8 image
// with the default mask: an int16 concat, flow_class.py:615); 0 = whatever the arguments say.  Same code, the branches folded.
// (4 = 2 held to five waves per SIMD: the folded float kernel needs 74 VGPRs instead of 90, and the sixth wave that buys is worth 5 % on
// config 2 and COSTS 5 % on config 5 't', whose scattered taps of a 400 MB image thrash the caches -- the launcher picks by the source's size)
template <typename T, int CT, int SPEC = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, SPEC == 4 ? 5 : 8)))
void gather2_kernel(const T *__restrict__ src0, int H, int W,
                    const float *__restrict__ flow0, int fH, int fW, int pad_top, int pad_left, int sign,
                    const uint8_t *__restrict__ smask0, const uint8_t *__restrict__ fmask0,
                    T *__restrict__ dst0, uint8_t *__restrict__ valid0,
                    int quant, int arith, int rule, int tiles_x, int nblocks, int row0, int rows, int xpose_rows, GBatch bs)
{
    if (SPEC == 1) { quant = OFL_QUANT_OPENCV; arith = OFL_ARITH_NATIVE; rule = OFL_RULE_GE_HALF; }
    if (SPEC == 2 || SPEC == 4) { quant = OFL_QUANT_OPENCV; arith = OFL_ARITH_NATIVE; rule = OFL_RULE_EQ1; }
    if (SPEC == 3) { quant = OFL_QUANT_OPENCV; arith = OFL_ARITH_FLOAT_RNE; rule = OFL_RULE_GT_HALF; }
    // (batched launches: one field per blockIdx.y; the offsets below stay 32-bit relative to the FIELD's base pointers)
    const size_t bi = blockIdx.y;
    const T *__restrict__ src = src0 + bi * bs.src;
    const float *__restrict__ flow = flow0 + bi * bs.flow;
    const uint8_t *__restrict__ smask = smask0 ? smask0 + bi * bs.smask : nullptr;
    const uint8_t *__restrict__ fmask = fmask0 ? fmask0 + bi * bs.fmask : nullptr;
    T *__restrict__ dst = dst0 + bi * bs.dst;
    uint8_t *__restrict__ valid = valid0 ? valid0 + bi * bs.valid : nullptr;
    const int tile = nblocks > 0 ? xcd_swizzle(blockIdx.x, nblocks) : (int)blockIdx.x;     // nblocks <= 0: natural order
    const int ty   = tile / tiles_x, tx = tile - ty * tiles_x;
    const int lx   = threadIdx.x & 31, yl = ty * 8 + (threadIdx.x >> 5), y = row0 + yl;     // output band [row0, row0 + rows)
    const int xg[2] = { tx * 128 + 2 * lx, tx * 128 + 64 + 2 * lx };
    const bool act[2] = { yl < rows && xg[0] < W, yl < rows && xg[1] < W };
    const int  fy = y - pad_top;
    const bool row_in_flow = (unsigned)fy < (unsigned)fH;
    const bool aligned = ((pad_left | fW) & 1) == 0;          // 16-byte aligned flow pairs

    float fu[4], fv[4];
    bool  inf[4];
    uint32_t fmw[2] = { 0u, 0u };          // flow-mask bytes of the pixel pairs, fetched with the flow (not at the store)
    // The common wave -- both pixel pairs of every lane inside the flow area, pairs 16-byte aligned -- asks for its two vector
    // pairs and its two mask words in one go; the general form below predicates every load and waits for each (its loads sit in
    // divergent branches: four dependent rounds of loads before the first tap is asked for).
    const bool whole = aligned && act[0] && act[1] && row_in_flow && (unsigned)(xg[0] - pad_left) < (unsigned)fW && (unsigned)(xg[0] - pad_left + 1) < (unsigned)fW &&
                       (unsigned)(xg[1] - pad_left) < (unsigned)fW && (unsigned)(xg[1] - pad_left + 1) < (unsigned)fW;
    if (__all(whole)) {
        const uint32_t o0 = __umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)(xg[0] - pad_left), o1 = __umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)(xg[1] - pad_left);
        const float4 f0 = *reinterpret_cast<const float4 *>(flow + o0 * 2), f1 = *reinterpret_cast<const float4 *>(flow + o1 * 2);
        if (fmask && valid) {
            fmw[0] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(fmask + o0));
            fmw[1] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(fmask + o1));
        }
        fu[0] = f0.x; fv[0] = f0.y; fu[1] = f0.z; fv[1] = f0.w; fu[2] = f1.x; fv[2] = f1.y; fu[3] = f1.z; fv[3] = f1.w;
        inf[0] = inf[1] = inf[2] = inf[3] = true;
    } else
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int fx = xg[g] - pad_left;
        inf[2 * g]     = act[g] && row_in_flow && (unsigned)fx < (unsigned)fW;
        inf[2 * g + 1] = act[g] && row_in_flow && (unsigned)(fx + 1) < (unsigned)fW;
        fu[2 * g] = fv[2 * g] = fu[2 * g + 1] = fv[2 * g + 1] = 0.0f;
        if (fmask && valid) {
            if (aligned && inf[2 * g] && inf[2 * g + 1]) {
                fmw[g] = __builtin_nontemporal_load(reinterpret_cast<const uint16_t *>(fmask + __umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)fx));
            } else {
                if (inf[2 * g]) fmw[g] |= fmask[__umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)fx];
                if (inf[2 * g + 1]) fmw[g] |= (uint32_t)fmask[__umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)fx + 1] << 8;
            }
        }
        if (aligned && inf[2 * g] && inf[2 * g + 1]) {
#if OFL_G2_NT_FLOW
            const v4f f = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(flow + (__umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)fx) * 2));
#else
            const float4 f = *reinterpret_cast<const float4 *>(flow + (__umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)fx) * 2);
#endif
            fu[2 * g] = f.x; fv[2 * g] = f.y; fu[2 * g + 1] = f.z; fv[2 * g + 1] = f.w;
        } else {
#pragma unroll
            for (int e = 0; e < 2; ++e)
                if (inf[2 * g + e]) {
                    const float2 f = *reinterpret_cast<const float2 *>(flow + (__umul24((uint32_t)fy, (uint32_t)fW) + (uint32_t)fx + e) * 2);
                    fu[2 * g + e] = f.x; fv[2 * g + e] = f.y;
                }
        }
    }

    const bool want_valid = valid != nullptr;
    T    res[4][CT];
    bool ok[4];
    // transposed gather for rotated sampling grids (see K2, variant 3) -- float images with 3 or 4 channels only:
    // measured +10 % on a 30-degree grid.  For 1 / 2 channels and for 8- / 16-bit images the extra registers cost a wave
    // of occupancy (and the byte-wise LDS hand-over is slow), which loses more than the fewer tag look-ups save.
    constexpr bool kXp = sizeof(T) == 4 && CT >= 3;
    __shared__ __attribute__((aligned(16))) float2  xp_f[kXp ? 8 * kXpRowF2 : 1];
    __shared__ __attribute__((aligned(16))) T       xp_r[kXp ? 8 * 128 * CT : 1];
    __shared__ __attribute__((aligned(16))) uint8_t xp_k[kXp ? 8 * 128 : 1];
    bool rotated = false;
    if constexpr (kXp) {
        // uniform decision from the vertical flow at the two ends of the tile's first row (zero outside the flow area)
        const int x0 = tx * 128, x1 = min(x0 + 127, W - 1), y0 = row0 + min(ty * 8, rows - 1);
        const int fy0 = y0 - pad_top, fxa = x0 - pad_left, fxb = x1 - pad_left;
        const bool rin = (unsigned)fy0 < (unsigned)fH;
        const float va = (rin && (unsigned)fxa < (unsigned)fW) ? flow[((size_t)fy0 * fW + fxa) * 2 + 1] : 0.0f;
        const float vb = (rin && (unsigned)fxb < (unsigned)fW) ? flow[((size_t)fy0 * fW + fxb) * 2 + 1] : 0.0f;
        rotated = fabsf(vb - va) * 128.0f > (float)(xpose_rows * (x1 - x0 + 1));
    }
    if (!rotated) {
        const int  gx[4]   = { xg[0], xg[0] + 1, xg[1], xg[1] + 1 };
        const bool act4[4] = { act[0], act[0], act[1], act[1] };
        gather2_core<T, CT>(src, smask, H, W, gx, y, act4, fu, fv, sign, quant, arith, rule, want_valid, res, ok);
    } else if constexpr (kXp) {
        const int ly = threadIdx.x >> 5;
        *reinterpret_cast<float4 *>(&xp_f[ly * kXpRowF2 + 2 * lx])      = make_float4(fu[0], fv[0], fu[1], fv[1]);
        *reinterpret_cast<float4 *>(&xp_f[ly * kXpRowF2 + 64 + 2 * lx]) = make_float4(fu[2], fv[2], fu[3], fv[3]);
        __syncthreads();
        // block layout: wave w owns columns [32 w, 32 w + 32) of the 8 rows; gather instruction j covers the 8 x 8 block
        // of columns 8 j .. 8 j + 7
        const int w4 = threadIdx.x >> 6, l = threadIdx.x & 63, r = l >> 3, c0 = 32 * w4 + (l & 7);
        {
            const int ylb = ty * 8 + r;
            int   gxb[4];
            bool  actb[4];
            float fub[4], fvb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float2 f = xp_f[r * kXpRowF2 + c0 + 8 * j];
                fub[j] = f.x; fvb[j] = f.y;
                gxb[j] = tx * 128 + c0 + 8 * j;
                actb[j] = ylb < rows && gxb[j] < W;
            }
            T    resb[4][CT];
            bool okb[4];
            gather2_core<T, CT>(src, smask, H, W, gxb, row0 + ylb, actb, fub, fvb, sign, quant, arith, rule, want_valid, resb, okb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int c = 0; c < CT; ++c) xp_r[(r * 128 + c0 + 8 * j) * CT + c] = resb[j][c];
                xp_k[r * 128 + c0 + 8 * j] = okb[j] ? 1 : 0;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = (j >> 1) * 64 + 2 * lx + (j & 1);
#pragma unroll
            for (int c = 0; c < CT; ++c) res[j][c] = xp_r[(ly * 128 + col) * CT + c];
            ok[j] = xp_k[ly * 128 + col] != 0;
        }
    }

    uint32_t vword[2] = { 0u, 0u };
    // 8- and 16-bit results leave as DWORDS as well: the two pixels of a lane are 2 .. 16 bytes; 2- and 6-byte pairs are
    // joined with the neighbouring lane's (four consecutive pixels = 4 / 12 bytes, dword-aligned when W is a multiple of 4).
    // Element-wise byte stores made this kernel instruction-bound: RGB uint8 cost 12 stores per lane, now 3 per lane PAIR.
    constexpr int kPB = 2 * CT * (int)sizeof(T);          // bytes of a pixel pair
    constexpr bool kPack = sizeof(T) <= 2;
    const bool quad4 = (W & 3) == 0;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const uint32_t o = __umul24((uint32_t)yl, (uint32_t)W) + (uint32_t)xg[g];
        if constexpr (kPack) {
            uint32_t pw[4] = { 0u, 0u, 0u, 0u };
#pragma unroll
            for (int i = 0; i < 2 * CT; ++i) {
                const uint32_t v = (uint32_t)(typename std::make_unsigned<T>::type)res[2 * g + i / CT][i % CT];
                if constexpr (sizeof(T) == 1) pw[i >> 2] |= v << (8 * (i & 3));
                else                          pw[i >> 1] |= v << (16 * (i & 1));
            }
            uint8_t *d8 = reinterpret_cast<uint8_t *>(dst) + o * (uint32_t)(CT * sizeof(T));
            if constexpr (kPB == 2 || kPB == 6) {
                const uint32_t q0 = (uint32_t)__shfl_xor((int)pw[0], 1), q1 = (uint32_t)__shfl_xor((int)pw[1], 1);
                if (quad4) {
                    if (act[g] && (lx & 1) == 0) {
                        if constexpr (kPB == 2) {
                            *reinterpret_cast<uint32_t *>(d8) = pw[0] | (q0 << 16);
                        } else {
                            uint32_t *d32 = reinterpret_cast<uint32_t *>(d8);
                            d32[0] = pw[0]; d32[1] = pw[1] | (q0 << 16); d32[2] = (q0 >> 16) | (q1 << 16);
                        }
                    }
                } else if (act[g]) {
                    if constexpr (kPB == 2) *reinterpret_cast<uint16_t *>(d8) = (uint16_t)pw[0];
                    else { *reinterpret_cast<uint16_t *>(d8) = (uint16_t)pw[0]; *reinterpret_cast<uint16_t *>(d8 + 2) = (uint16_t)(pw[0] >> 16);
                           *reinterpret_cast<uint16_t *>(d8 + 4) = (uint16_t)pw[1]; }
                }
            } else if (act[g]) {
                if constexpr (kPB == 4)       *reinterpret_cast<uint32_t *>(d8) = pw[0];
                else if constexpr (kPB == 8)  *reinterpret_cast<uint2 *>(d8) = make_uint2(pw[0], pw[1]);
                else if constexpr (kPB == 12) { uint32_t *d32 = reinterpret_cast<uint32_t *>(d8); d32[0] = pw[0]; d32[1] = pw[1]; d32[2] = pw[2]; }
                else                          *reinterpret_cast<uint4 *>(d8) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            }
        } else if (act[g]) {
            T *d = dst + o * (uint32_t)CT;
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
#if OFL_G2_NT_DST
                for (int c = 0; c < CT; ++c) __builtin_nontemporal_store(res[2 * g + e][c], &d[e * CT + c]);
#else
                for (int c = 0; c < CT; ++c) d[e * CT + c] = res[2 * g + e][c];
#endif
        }
        if (act[g] && want_valid) {
            uint32_t m = (ok[2 * g] ? 1u : 0u) | (ok[2 * g + 1] ? 0x100u : 0u);
            if (fmask) m &= ((fmw[g] & 0xffu) ? 1u : 0u) | ((fmw[g] & 0xff00u) ? 0x100u : 0u);
            vword[g] = m;
        }
    }
    if (want_valid) {
        // Validity bytes leave as DWORDS: the lane pairs (2k, 2k + 1) own four consecutive pixels, the even lane
        // stores both lanes' bytes -- sub-dword stores cost as much per instruction as 16-byte ones.
        const bool quad = (W & 3) == 0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const uint32_t other = (uint32_t)__shfl_xor((int)vword[g], 1);
            if (!act[g]) continue;
            const uint32_t o = __umul24((uint32_t)yl, (uint32_t)W) + (uint32_t)xg[g];
            if (quad) {
                if ((lx & 1) == 0) __builtin_nontemporal_store(vword[g] | (other << 16), reinterpret_cast<uint32_t *>(valid + o));
            } else {
                __builtin_nontemporal_store((uint16_t)vword[g], reinterpret_cast<uint16_t *>(valid + o));
            }
        }
    }
}

template <typename T>
int launch_gather_t(const void *src, int C, int H, int W, const float *flow, int fH, int fW, int pad_top,
                    int pad_left, int sign, const uint8_t *smask, const uint8_t *fmask, void *dst,
                    uint8_t *valid, int quant, int arith, int rule, int row0, int rows, hipStream_t s,
                    int batch = 1, bool shared_src = false, bool shared_smask = false)
{
    GBatch bs;
    bs.src = shared_src ? 0 : (size_t)H * W * C; bs.smask = shared_smask ? 0 : (size_t)H * W;
    bs.flow = (size_t)fH * fW * 2; bs.fmask = (size_t)fH * fW; bs.dst = (size_t)rows * W * C; bs.valid = (size_t)rows * W;
    // (the paired kernel addresses with 32-bit offsets: images, flows and results below 4 GiB)
    if (W % 2 == 0 && C >= 1 && C <= 4 && (unsigned long long)H * W * C * sizeof(T) < (1ull << 32) && (unsigned long long)fH * fW * 8 < (1ull << 32)) {
        const int tiles_x = (W + 127) / 128, tiles_y = (rows + 7) / 8;
        const int nblocks = tiles_x * tiles_y;
        static const int swz = OFL_KNOB_INT("OFL_G2_SWZ", 0);                      // 1 = XCD swizzle (experiments build only)
        static const int xpose_rows = OFL_KNOB_INT("OFL_G2_XPOSE_ROWS", kXposeRows);   // (experiments build only)
#define OFL_GATHER2_LAUNCH_S(CT, SPEC)                                                                   \
        hipLaunchKernelGGL((gather2_kernel<T, CT, SPEC>), dim3(nblocks, batch), dim3(256), 0, s, (const T *)src, H, W, \
                           flow, fH, fW, pad_top, pad_left, sign, smask, fmask, (T *)dst, valid, quant,   \
                           arith, rule, tiles_x, swz ? nblocks : 0, row0, rows, xpose_rows, bs)
#define OFL_GATHER2_LAUNCH(CT) OFL_GATHER2_LAUNCH_S(CT, 0)
        static const int spec_on = OFL_KNOB_INT("OFL_G2_SPEC", 1);                 // (experiments build only) 0: the general kernel for everything
        // the combinations the Flow algebra asks for run as specialised instantiations (same code, branches folded: 7 - 10 % faster)
        int spec = 0;
        if (spec_on && quant == OFL_QUANT_OPENCV) {
            if (std::is_same<T, uint8_t>::value && arith == OFL_ARITH_NATIVE && rule == OFL_RULE_GE_HALF) spec = 1;
            else if (std::is_same<T, float>::value && arith == OFL_ARITH_NATIVE && rule == OFL_RULE_EQ1)
                spec = (unsigned long long)H * W * C * sizeof(T) > (128ull << 20) ? 4 : 2;      // a source beyond half the Infinity Cache: five waves
            else if (std::is_same<T, uint8_t>::value && arith == OFL_ARITH_FLOAT_RNE && rule == OFL_RULE_GT_HALF) spec = 3;
        }
        bool launched = false;
        if constexpr (std::is_same<T, uint8_t>::value || std::is_same<T, float>::value) {
            constexpr int kA = std::is_same<T, float>::value ? 2 : 1, kB = std::is_same<T, float>::value ? 4 : 3;      // the specialisations of this type
            if (spec != 0) {
                launched = true;
                const int cc = C < 4 ? C : 4;
                if (spec == kA)      { if (cc == 1) OFL_GATHER2_LAUNCH_S(1, kA); else if (cc == 2) OFL_GATHER2_LAUNCH_S(2, kA); else if (cc == 3) OFL_GATHER2_LAUNCH_S(3, kA); else OFL_GATHER2_LAUNCH_S(4, kA); }
                else                 { if (cc == 1) OFL_GATHER2_LAUNCH_S(1, kB); else if (cc == 2) OFL_GATHER2_LAUNCH_S(2, kB); else if (cc == 3) OFL_GATHER2_LAUNCH_S(3, kB); else OFL_GATHER2_LAUNCH_S(4, kB); }
            }
        }
        if (!launched) switch (C) {
        case 1: OFL_GATHER2_LAUNCH(1); break;
        case 2: OFL_GATHER2_LAUNCH(2); break;
        case 3: OFL_GATHER2_LAUNCH(3); break;
        default: OFL_GATHER2_LAUNCH(4); break;
        }
#undef OFL_GATHER2_LAUNCH
#undef OFL_GATHER2_LAUNCH_S
        OFL_HIP(hipGetLastError());
        return OFL_OK;
    }
    const int tiles_x = (W + 31) / 32, tiles_y = (rows + 7) / 8;
    const int nblocks = tiles_x * tiles_y;
#define OFL_GATHER_LAUNCH(CT)                                                                          \
    hipLaunchKernelGGL((gather_kernel<T, CT>), dim3(nblocks, batch), dim3(256), 0, s, (const T *)src, C, H, W, \
                       flow, fH, fW, pad_top, pad_left, sign, smask, fmask, (T *)dst, valid, quant,     \
                       arith, rule, tiles_x, nblocks, row0, rows, bs)
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

#ifdef OFL_EXPERIMENTS
// resident workgroups per CU of each compose3 instantiation (queried once per process)
template <typename K>
int c3_query_blocks(K kernel)
{
    int n = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, 256, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 4; }
    return n;
}

int c3_blocks_per_cu(int quant, bool with_stats, bool lds)
{
    static int cache[2][2][2] = { { { 0, 0 }, { 0, 0 } }, { { 0, 0 }, { 0, 0 } } };
    const int q = quant == OFL_QUANT_OPENCV ? 0 : 1;
    int &v = cache[q][with_stats ? 1 : 0][lds ? 1 : 0];
    if (v == 0) {
        int n;
        if (lds) {
            if (q == 0) n = with_stats ? c3_query_blocks(compose3_lds_kernel<OFL_QUANT_OPENCV, true>) : c3_query_blocks(compose3_lds_kernel<OFL_QUANT_OPENCV, false>);
            else        n = with_stats ? c3_query_blocks(compose3_lds_kernel<OFL_QUANT_EXACT, true>) : c3_query_blocks(compose3_lds_kernel<OFL_QUANT_EXACT, false>);
        } else {
            if (q == 0) n = with_stats ? c3_query_blocks(compose3_kernel<OFL_QUANT_OPENCV, true>) : c3_query_blocks(compose3_kernel<OFL_QUANT_OPENCV, false>);
            else        n = with_stats ? c3_query_blocks(compose3_kernel<OFL_QUANT_EXACT, true>) : c3_query_blocks(compose3_kernel<OFL_QUANT_EXACT, false>);
        }
        const char *env = getenv("OFL_C3_BLOCKS_PER_CU");      // tuning knob
        if (env && atoi(env) > 0) n = atoi(env);
        v = n < 1 ? 1 : (n > 8 ? 8 : n);
    }
    return v;
}
#endif  // OFL_EXPERIMENTS

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

// uint8 masks <-> packed bit planes (rows padded to whole words): one thread per word
__global__ __launch_bounds__(256)
void mask_pack_kernel(const uint8_t *__restrict__ mask, size_t rows, int W, int wpr, uint32_t *__restrict__ bits)
{
    const size_t words = rows * (size_t)wpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        const size_t r = i / (size_t)wpr;
        const int x0 = (int)(i - r * (size_t)wpr) * 32;
        const uint8_t *m = mask + r * (size_t)W + x0;
        uint32_t w = 0;
        if (x0 + 32 <= W && (((size_t)m) & 15) == 0) {
            const uint4 a = reinterpret_cast<const uint4 *>(m)[0], b = reinterpret_cast<const uint4 *>(m)[1];
            const uint32_t d[8] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w };
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) w |= ((d[k] >> (8 * e)) & 0xffu) ? 1u << (4 * k + e) : 0u;
        } else {
            for (int k = 0; k < 32 && x0 + k < W; ++k) w |= m[k] ? 1u << k : 0u;
        }
        bits[i] = w;
    }
}

__global__ __launch_bounds__(256)
void mask_unpack_kernel(const uint32_t *__restrict__ bits, size_t rows, int W, int wpr, uint8_t *__restrict__ mask)
{
    const size_t words = rows * (size_t)wpr;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
        const size_t r = i / (size_t)wpr;
        const int x0 = (int)(i - r * (size_t)wpr) * 32;
        const uint32_t w = bits[i];
        uint8_t *m = mask + r * (size_t)W + x0;
        for (int k = 0; k < 32 && x0 + k < W; ++k) m[k] = (w >> k) & 1u;
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

    if (W % 2 == 0) {
        // One workgroup per 128 x 8 tile in natural order -- the dispatcher keeps every wave slot filled and all XCDs sweep
        // one window of memory (measured +2..7 % over a persistent grid at 1..8 4K pairs per launch) -- with the gather
        // transposed through LDS in workgroups whose sampling grid is rotated.
        int tiles_x = (W + kC3TileW - 1) / kC3TileW, tiles_y = (H + kC3TileH - 1) / kC3TileH;
        long long nt = (long long)tiles_x * tiles_y * batch;
        if (nt > 0x7fffffffLL) return fail(OFL_E_INVALID, "ofl_compose3: too many tiles");
        static const int xpose_rows = OFL_KNOB_INT("OFL_C3_XPOSE_ROWS", kXposeRows);
        static const int xcd_rows = OFL_KNOB_INT("OFL_C3_XCD_ROWS", kC3XcdRows);       // tile rows one XCD owns per group (see the kernel)
        const long long n_groups = (long long)batch * ((tiles_x + 7) / 8) * ((tiles_y + xcd_rows - 1) / xcd_rows);
        const long long nblk = xcd_rows > 1 ? n_groups * 8 * xcd_rows : nt;
        if (nblk > 0x7fffffffLL) return fail(OFL_E_INVALID, "ofl_compose3: too many tiles");
#ifdef OFL_EXPERIMENTS
        // A/B variants: OFL_C3_VARIANT = 2 the same without the transposition, 0 persistent grid with stream prefetch,
        // 1 source tile staged in LDS; OFL_C3_ABLATE switches parts of the kernel off, OFL_C3_SWZ tries XCD-aware tile orders
        static const int ablate = OFL_KNOB_INT("OFL_C3_ABLATE", 0);
        static const int variant = OFL_KNOB_INT("OFL_C3_VARIANT", 3);
        static const int swz = OFL_KNOB_INT("OFL_C3_SWZ", 0);
        const bool use_lds = variant == 1;
        if (use_lds) {
            tiles_x = (W + 4 * kLdsLX - 1) / (4 * kLdsLX); tiles_y = (H + 256 / kLdsLX - 1) / (256 / kLdsLX);
            nt = (long long)tiles_x * tiles_y * batch;
        }
        C3Args a = { fa, ma, fb, mb, out, mout, stats, sign, H, W, tiles_x, tiles_x * tiles_y, (int)nt, th, 0, variant == 3 ? xcd_rows : swz, xpose_rows, ablate };
        int grid = rt().n_cu * c3_blocks_per_cu(quant, stats != nullptr, use_lds);      // persistent variants: what the chip keeps resident
        if (grid > (int)nt) grid = (int)nt;
        if (grid >= 8) grid &= ~7;
#define OFL_C3(Q, S) do { if (use_lds) hipLaunchKernelGGL((compose3_lds_kernel<Q, S>), dim3(grid), dim3(256), 0, s, a); \
                          else if (variant == 2) hipLaunchKernelGGL((compose3_oneshot_kernel<Q, S>), dim3((int)nt), dim3(256), 0, s, a); \
                          else if (variant == 0) hipLaunchKernelGGL((compose3_kernel<Q, S>), dim3(grid), dim3(256), 0, s, a); \
                          else hipLaunchKernelGGL((compose3_xpose_kernel<Q, S>), dim3((int)nblk), dim3(256), 0, s, a); } while (0)
#else
        C3Args a = { fa, ma, fb, mb, out, mout, stats, sign, H, W, tiles_x, tiles_x * tiles_y, (int)nt, th, 0, xcd_rows, xpose_rows };
#define OFL_C3(Q, S) hipLaunchKernelGGL((compose3_xpose_kernel<Q, S>), dim3((int)nblk), dim3(256), 0, s, a)
#endif
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

int ofl_mask_bits_bytes(int H, int W, int batch, size_t *bytes)
{
    if (!bytes || H <= 0 || W <= 0 || batch <= 0) return fail(OFL_E_INVALID, "ofl_mask_bits_bytes: bad arguments");
    *bytes = (size_t)batch * H * ((W + 31) / 32) * 4 + 16;      // 16 bytes of slack: the gather reads 4 bytes from the byte of a tap's bit
    return OFL_OK;
}

int ofl_mask_pack_dev(const uint8_t *mask, int H, int W, int batch, uint32_t *bits, void *stream)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_mask_pack", H, W));
    if (!mask || !bits || batch <= 0) return fail(OFL_E_INVALID, "ofl_mask_pack: NULL pointer / batch < 1");
    const size_t words = (size_t)batch * H * ((W + 31) / 32);
    hipLaunchKernelGGL(mask_pack_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 65535)), dim3(256), 0, stream_of(stream),
                       mask, (size_t)batch * H, W, (W + 31) / 32, bits);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_mask_unpack_dev(const uint32_t *bits, int H, int W, int batch, uint8_t *mask, void *stream)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_mask_unpack", H, W));
    if (!mask || !bits || batch <= 0) return fail(OFL_E_INVALID, "ofl_mask_unpack: NULL pointer / batch < 1");
    const size_t words = (size_t)batch * H * ((W + 31) / 32);
    hipLaunchKernelGGL(mask_unpack_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 65535)), dim3(256), 0, stream_of(stream),
                       bits, (size_t)batch * H, W, (W + 31) / 32, mask);
    OFL_HIP(hipGetLastError());
    return OFL_OK;
}

int ofl_compose3_bits_dev(const float *fa, const uint32_t *ma_bits, const float *fb, const uint32_t *mb_bits,
                          int sign, int H, int W, int batch, float *out, uint32_t *mout_bits,
                          uint32_t *stats, void *stream)
{
    OFL_TRY(need_device());
    OFL_TRY(check_dims("ofl_compose3_bits", H, W));
    if (!fa || !ma_bits || !fb || !mb_bits || !out || !mout_bits) return fail(OFL_E_INVALID, "ofl_compose3_bits: NULL pointer");
    if (batch <= 0) return fail(OFL_E_INVALID, "ofl_compose3_bits: batch must be >= 1");
    if (sign != 1 && sign != -1) return fail(OFL_E_INVALID, "ofl_compose3_bits: sign must be +1 or -1");
    if (W % 2) return fail(OFL_E_INVALID, "ofl_compose3_bits: the packed-mask kernel takes even widths (odd ones: ofl_compose3_dev)");
    hipStream_t s = stream_of(stream);
    const int tiles_x = (W + kC3TileW - 1) / kC3TileW, tiles_y = (H + kC3TileH - 1) / kC3TileH;
    const long long nt = (long long)tiles_x * tiles_y * batch;
    if (nt > 0x7fffffffLL) return fail(OFL_E_INVALID, "ofl_compose3_bits: too many tiles");
    C3Args a = { fa, reinterpret_cast<const uint8_t *>(ma_bits), fb, reinterpret_cast<const uint8_t *>(mb_bits), out,
                 reinterpret_cast<uint8_t *>(mout_bits), stats, sign, H, W, tiles_x, tiles_x * tiles_y, (int)nt, 1e-3f, (W + 31) / 32, 1, kXposeRows
#ifdef OFL_EXPERIMENTS
                 , 0
#endif
    };
    if (stats) hipLaunchKernelGGL((compose3_xpose_kernel<OFL_QUANT_OPENCV, true, true>), dim3((int)nt), dim3(256), 0, s, a);
    else       hipLaunchKernelGGL((compose3_xpose_kernel<OFL_QUANT_OPENCV, false, true>), dim3((int)nt), dim3(256), 0, s, a);
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
                             nullptr, quant, s));
    OFL_HIP(hipMemcpyAsync(out, dout.p, n * 8, hipMemcpyDeviceToHost, s));
    OFL_HIP(hipMemcpyAsync(mout, dmout.p, n, hipMemcpyDeviceToHost, s));
    // exact predicates for the host caller: one pass of the statistics kernel per field (the fused
    // launch above only certifies "fa is not zero" from the vectors it happened to gather)
    for (int b = 0; stats_host && b < batch; ++b) {
        const size_t hw = (size_t)H * W;
        uint32_t bits[2] = { 0, 0 };
        for (int k = 0; k < 2; ++k) {
            const float *f = (const float *)(k == 0 ? dfa.p : dfb.p) + (size_t)b * hw * 2;
            const uint8_t *m = (const uint8_t *)(k == 0 ? dma.p : dmb.p) + (size_t)b * hw;
            OFL_TRY(ofl_flow_stats_dev(f, m, hw, 1e-3f, (uint32_t *)dst.p, s));
            OFL_HIP(hipMemcpyAsync(&bits[k], dst.p, 4, hipMemcpyDeviceToHost, s));
            OFL_HIP(hipStreamSynchronize(s));
        }
        stats_host[2 * b] = bits[0] & 15u; stats_host[2 * b + 1] = bits[1] & 15u;
    }
    OFL_HIP(hipStreamSynchronize(s));
    return OFL_OK;
}

static int gather_rows_impl(const void *src, int dtype, int C, int H, int W,
                            const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                            const uint8_t *smask, const uint8_t *fmask,
                            void *dst, uint8_t *valid,
                            int quant, int arith, int rule, int row0, int rows, void *stream,
                            int batch = 1, bool shared_src = false, bool shared_smask = false)
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
    if (row0 < 0 || rows <= 0 || row0 + rows > H)
        return fail(OFL_E_INVALID, "ofl_gather_bilinear: rows [%d, %d) outside the %d-row result", row0, row0 + rows, H);
    hipStream_t s = stream_of(stream);
    switch (dtype) {
    case OFL_U8:  return launch_gather_t<uint8_t>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, row0, rows, s, batch, shared_src, shared_smask);
    case OFL_I16: return launch_gather_t<int16_t>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, row0, rows, s, batch, shared_src, shared_smask);
    case OFL_U16: return launch_gather_t<uint16_t>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, row0, rows, s, batch, shared_src, shared_smask);
    case OFL_F32: return launch_gather_t<float>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, row0, rows, s, batch, shared_src, shared_smask);
    default:      return launch_gather_t<double>(src, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid, quant, arith, rule, row0, rows, s, batch, shared_src, shared_smask);
    }
}

int ofl_gather_bilinear_dev(const void *src, int dtype, int C, int H, int W,
                            const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                            const uint8_t *smask, const uint8_t *fmask,
                            void *dst, uint8_t *valid,
                            int quant, int arith, int rule, void *stream)
{
    return gather_rows_impl(src, dtype, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid,
                            quant, arith, rule, 0, H, stream);
}

int ofl_gather_bilinear_batch_dev(const void *src, int src_shared, int dtype, int C, int H, int W, int batch,
                                  const float *flow, int fH, int fW, int pad_top, int pad_left, int sign,
                                  const uint8_t *smask, int smask_shared, const uint8_t *fmask,
                                  void *dst, uint8_t *valid, int quant, int arith, int rule, void *stream)
{
    if (batch < 1 || batch > 65535) return fail(OFL_E_INVALID, "ofl_gather_bilinear_batch: batch must be in [1, 65535]");
    return gather_rows_impl(src, dtype, C, H, W, flow, fH, fW, pad_top, pad_left, sign, smask, fmask, dst, valid,
                            quant, arith, rule, 0, H, stream, batch, src_shared != 0, smask_shared != 0);
}

int ofl_gather_rows_dev(const void *src, int dtype, int C, int H, int W, int row0, int rows,
                        const float *flow_rows, int sign, const uint8_t *smask, const uint8_t *fmask_rows,
                        void *dst_rows, uint8_t *valid_rows, int quant, int arith, int rule, void *stream)
{
    if (rows <= 0 || row0 < 0) return fail(OFL_E_INVALID, "ofl_gather_rows: bad row band");
    return gather_rows_impl(src, dtype, C, H, W, flow_rows, rows, W, row0, 0, sign, smask, fmask_rows, dst_rows, valid_rows,
                            quant, arith, rule, row0, rows, stream);
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
