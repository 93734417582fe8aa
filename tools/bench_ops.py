#!/usr/bin/env python
"""The BASELINE.json configs besides the headline (2, 3, 4, 5) and the Delaunay-path cases, device-resident and
HIP-event timed.  One JSON object per operation with the algorithmic bytes of SURVEY.md 8(d).

Every operation ROTATES over distinct input AND output buffers whose footprints sum to at least 3 x 256 MiB (the
Infinity Cache), exactly as bench.py does for the headline kernel: round 2 timed one resident working set in a loop,
and the rates of everything that fits the cache were cache rates.

    python tools/bench_ops.py [--iters 20] [--only substring[,substring]]        # substrings of the keys below

bench.py imports `collect()` and appends the same entries to the driver-run line as `"configs": {key: {...}}`
(one rocprofv3 row per key: tools/prof_ops.sh).
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev

nat = of.native
MALL = 256 << 20


def n_sets(footprint_bytes):
    """how many distinct working sets a rotation needs so that no launch finds its bytes in the Infinity Cache"""
    return max(2, int(math.ceil(3.0 * MALL / max(1, footprint_bytes))))


def timed(fns, iters, warm=3):
    """fns: one closure per working set (or a single closure).  Results are kept alive in a ring as long as the rotation,
    so that buffers from the recycling pool are distinct per set, too."""
    lib = nat.load()
    fns = fns if isinstance(fns, (list, tuple)) else [fns]
    ring = [None] * len(fns)
    for i in range(max(warm, len(fns))):
        ring[i % len(fns)] = fns[i % len(fns)]()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    nat.check(lib.ofl_event_create(ctypes.byref(e0)))
    nat.check(lib.ofl_event_create(ctypes.byref(e1)))
    nat.check(lib.ofl_device_sync())
    t0 = time.perf_counter()
    nat.check(lib.ofl_event_record(e0, None))
    for i in range(iters):
        ring[i % len(fns)] = fns[i % len(fns)]()
    nat.check(lib.ofl_event_record(e1, None))
    nat.check(lib.ofl_device_sync())
    wall = (time.perf_counter() - t0) / iters
    ms = ctypes.c_float()
    nat.check(lib.ofl_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    nat.check(lib.ofl_event_destroy(e0))
    nat.check(lib.ofl_event_destroy(e1))
    return ms.value / iters * 1e-3, wall, len(fns)


def entry(key, name, shape, bytes_per_px, dev_s, wall_s, sets=1, note="", units=1, kernel=""):
    n = shape[0] * shape[1] * units
    return {"key": key, "op": name, "shape": list(shape), "algorithmic_bytes": int(round(bytes_per_px * n)),
            "device_ms": round(dev_s * 1e3, 4), "wall_ms": round(wall_s * 1e3, 4),
            "GBps_algorithmic": round(bytes_per_px * n / dev_s / 1e9, 1),
            "frac_of_8TBps": round(bytes_per_px * n / dev_s / 8e12, 4), "rotating_sets": sets, "kernel": kernel, "note": note}


def report(name, shape, bytes_per_px, dev_s, wall_s, sets=1, note="", units=1, key="", kernel=""):
    print(json.dumps(entry(key, name, shape, bytes_per_px, dev_s, wall_s, sets, note, units, kernel)), flush=True)


def copies(arr, n, wrap):
    """n device copies of one host array (the rotation is about where the bytes live, not about their values)"""
    return [wrap(arr) for _ in range(n)]


def affine_flow(transforms, h, w, ref):
    """Synthetic affine field for benchmark inputs: the same function of (x, y) as Flow.from_transforms, evaluated by
    broadcasting instead of 2 M tiny matmuls (1.2 s per 1080p field in the reference's formulation, which the host API keeps
    for bit-identical inputs in the tests).  Values agree to float32 rounding; the benchmarks only need the shape of the field."""
    from oflibnumpy_amd.utils import matrix_from_transforms
    m = matrix_from_transforms(transforms)
    if ref == 't':
        m = np.linalg.pinv(m)
    x, y = np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64)
    v = np.empty((h, w, 2), np.float32)
    if m[2, 0] == 0.0 and m[2, 1] == 0.0:                   # affine: separable sums, one float64 temporary per channel
        sg = (-1.0 if ref == 't' else 1.0) / m[2, 2]
        v[..., 0] = np.add.outer(sg * (m[0, 1] * y + m[0, 2]), sg * (m[0, 0] - m[2, 2]) * x)
        v[..., 1] = np.add.outer(sg * ((m[1, 1] - m[2, 2]) * y + m[1, 2]), sg * m[1, 0] * x)
        return v
    x, y = x[None, :], y[:, None]
    d = m[2, 0] * x + m[2, 1] * y + m[2, 2]
    v[..., 0] = (m[0, 0] * x + m[0, 1] * y + m[0, 2]) / d - x
    v[..., 1] = (m[1, 0] * x + m[1, 1] * y + m[1, 2]) / d - y
    return -v if ref == 't' else v


def flow_sets(vecs, ref, n_px, mask=None, bpp=18):
    ds = [dev.DeviceFlow.from_host(vecs, ref, mask) for _ in range(n_sets(bpp * n_px))]
    for d in ds:
        d.stats()
    return ds


# ------------------------------------------------------------------------------------------------ sections
def sec_config2(it):
    """config 2: 1080 x 1920 't' flow applied to an RGB image with valid area (K1)"""
    lib = nat.load()
    h, w = 1080, 1920
    n = h * w
    f1 = of.Flow(affine_flow([['rotation', 960, 540, -30]], h, w, 't'), 't')
    img = np.random.default_rng(1).random((h, w, 3), dtype=np.float32)
    k = n_sets(34 * n)
    flows = [f1.to_device() for _ in range(k)]
    imgs = copies(img, k, dev.DeviceImage.from_host)
    t = timed([(lambda d=d, im=im: dev.gather_bilinear(im, d.vecs, (h, w), -1, fmask=d.mask, want_valid=True))
               for d, im in zip(flows, imgs)], max(it, 3 * k))
    yield entry("c2_apply_t_1080p", "apply 't' RGB f32 + valid (K1)", (h, w), 34, *t, kernel="gather2_kernel",
                note="BASELINE config 2; rotated sampling pattern; one image per launch")
    if hasattr(dev, "gather_bilinear_batch"):
        # the same work as ONE launch over 16 flows x 16 images (the batch entry of K1): what a 1080p-sized job needs to get
        # past the single generation of waves a lone launch is
        B = 16
        sets = []
        for s in range(2):                                   # 2 x 16 x 70.5 MB = 2.3 GB: every launch streams from HBM
            fb, mb, ib = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 12)
            for i in range(B):
                nat.check(lib.ofl_copy_dev(fb.ptr + i * n * 8, flows[i % k].vecs.ptr, n * 8, None))
                nat.check(lib.ofl_copy_dev(mb.ptr + i * n, flows[i % k].mask.ptr, n, None))
                nat.check(lib.ofl_copy_dev(ib.ptr + i * n * 12, imgs[i % k].buf.ptr, n * 12, None))
            sets.append((fb, mb, ib, dev.DeviceBuffer(B * n * 12), dev.DeviceBuffer(B * n)))
        nat.check(lib.ofl_device_sync())
        t = timed([(lambda s=s: dev.gather_bilinear_batch(s[2], np.float32, 3, h, w, B, s[0], -1, fmask=s[1], dst=s[3], valid=s[4]))
                   for s in sets], max(it // 2, 6))
        yield entry("c2_apply_t_16x1080p_batch", "apply 't' RGB f32 + valid, 16 fields x 16 images in one launch (K1 batch)", (h, w), 34, *t,
                    units=B, kernel="gather2_kernel", note="BASELINE config 2 batched (ofl_gather_bilinear_batch_dev)")
        del sets
    k = n_sets(17 * n)
    flows = flows[:k] if len(flows) >= k else [f1.to_device() for _ in range(k)]
    u8img = (img * 255).astype(np.uint8)
    u8 = copies(u8img, k, dev.DeviceImage.from_host)
    tm = (np.random.default_rng(3).random((h, w)) > 0.1).astype(np.uint8)
    tms = copies(tm, k, dev.DeviceBuffer.from_host)
    t = timed([(lambda d=d, im=im, tm=tm: dev.gather_bilinear(im, d.vecs, (h, w), -1, smask=tm, fmask=d.mask, want_valid=True,
                                                               arith=nat.ARITH_NATIVE, rule=nat.RULE_GE_HALF))
               for d, im, tm in zip(flows, u8, tms)], max(it, 3 * k))
    yield entry("c2_apply_t_1080p_u8", "apply 't' RGB uint8 + target mask + valid (K1)", (h, w), 17, *t, kernel="gather2_kernel",
                note="8 + 1 flow, 3 + 3 image, 1 + 1 masks")
    if hasattr(dev, "gather_bilinear_batch"):
        B = 32
        sets = []
        for s in range(2):
            fb, mb, ib, sb = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 3), dev.DeviceBuffer(B * n)
            for i in range(B):
                nat.check(lib.ofl_copy_dev(fb.ptr + i * n * 8, flows[i % k].vecs.ptr, n * 8, None))
                nat.check(lib.ofl_copy_dev(mb.ptr + i * n, flows[i % k].mask.ptr, n, None))
                nat.check(lib.ofl_copy_dev(ib.ptr + i * n * 3, u8[i % k].buf.ptr, n * 3, None))
                nat.check(lib.ofl_copy_dev(sb.ptr + i * n, tms[i % k].ptr, n, None))
            sets.append((fb, mb, ib, sb, dev.DeviceBuffer(B * n * 3), dev.DeviceBuffer(B * n)))
        nat.check(lib.ofl_device_sync())
        t = timed([(lambda s=s: dev.gather_bilinear_batch(s[2], np.uint8, 3, h, w, B, s[0], -1, smask=s[3], fmask=s[1], dst=s[4], valid=s[5],
                                                          arith=nat.ARITH_NATIVE, rule=nat.RULE_GE_HALF)) for s in sets], max(it // 2, 6))
        yield entry("c2_apply_t_32x1080p_u8_batch", "apply 't' RGB uint8 + target mask + valid, 32 fields x 32 images in one launch (K1 batch)",
                    (h, w), 17, *t, units=B, kernel="gather2_kernel", note="the 8-bit job batched")


def sec_config4(it):
    """config 4, one GPU's share: 32 independent 1080 x 1920 pairs in ONE launch of the fused compose kernel; pair i =
    rotation by -30 + 60 i / 255 degrees about the centre (+) translation (40 cos i, 40 sin i): the SAMPLING field of
    mode 3 / 't' is the translation (flow_class.py:1422)"""
    lib = nat.load()
    h, w, B = 1080, 1920, 32
    n = h * w
    sets = []
    for s in range(2):                                   # 2 x 1.79 GB: every launch streams from HBM
        va, ma, vb, mb = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n)
        for i in range(B):
            j = i + 32 * s
            fa = affine_flow([['rotation', w / 2, h / 2, -30 + 60 * j / 255]], h, w, 't')
            fb = affine_flow([['translation', 40 * math.cos(j), 40 * math.sin(j)]], h, w, 't')
            m = (np.random.default_rng(j).random((h, w)) > 0.05).astype(np.uint8)
            for dst, a in ((va.ptr + i * n * 8, fa), (vb.ptr + i * n * 8, fb), (ma.ptr + i * n, m), (mb.ptr + i * n, m)):
                a = np.ascontiguousarray(a)
                nat.check(lib.ofl_upload(dst, a.ctypes.data, a.nbytes, None))
                nat.check(lib.ofl_stream_sync(None))
        sets.append((va, ma, vb, mb, dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer.zeros(32 * B)))
    t = timed([(lambda s=s: nat.check(lib.ofl_compose3_dev(s[0].ptr, s[1].ptr, s[2].ptr, s[3].ptr, -1, h, w, B, s[4].ptr, s[5].ptr,
                                                            s[6].ptr, 0, None))) for s in sets], max(it, 10))
    yield entry("c4_share_32x1080p", "combine_with mode 3 't', 32 pairs of 1080p in one launch (K2)", (h, w), 27, *t, units=B,
                kernel="compose3_xpose_kernel", note="BASELINE config 4, one GPU's share of the 256 pairs; sampling field = translation of 40 px")
    if hasattr(dev, "compose3_bits_launch"):
        # the same launch on PACKED mask planes (device-resident chains): 24.4 B/px really move; the fraction is still quoted on
        # the 27 B/px of the contract
        psets = []
        for s in sets:
            psets.append((s[0], dev.mask_pack(s[1], h, w, B), s[2], dev.mask_pack(s[3], h, w, B), s[4], dev.DeviceBuffer(dev.mask_bits_bytes(h, w, B)), s[6]))
        t = timed([(lambda s=s: dev.compose3_bits_launch(s[0], s[1], s[2], s[3], -1, (h, w), s[4], s[5], s[6], batch=B)) for s in psets], max(it, 10))
        yield entry("c4_share_32x1080p_bits", "the same on packed mask planes (K2, ofl_compose3_bits_dev)", (h, w), 27, *t, units=B,
                    kernel="compose3_xpose_kernel<bits>", note="masks as one bit per pixel: 24.4 B/px move; fraction on the 27 B/px contract")
        # the roles swapped: the SAMPLING field is the rotation (the pattern whose 1-byte mask taps cost most in partially used lines)
        t = timed([(lambda s=s: nat.check(lib.ofl_compose3_dev(s[2].ptr, s[3].ptr, s[0].ptr, s[1].ptr, -1, h, w, B, s[4].ptr, s[5].ptr,
                                                                s[6].ptr, 0, None))) for s in sets], max(it, 10))
        yield entry("c4_rot_32x1080p", "32 pairs of 1080p, sampling field = the rotations (K2)", (h, w), 27, *t, units=B, kernel="compose3_xpose_kernel",
                    note="config 4's pairs with the roles swapped: rotated sampling grids (-30 .. -22 degrees)")
        t = timed([(lambda s=s: dev.compose3_bits_launch(s[2], s[3], s[0], s[1], -1, (h, w), s[4], s[5], s[6], batch=B)) for s in psets], max(it, 10))
        yield entry("c4_rot_32x1080p_bits", "the same on packed mask planes", (h, w), 27, *t, units=B, kernel="compose3_xpose_kernel<bits>",
                    note="rotated sampling grids, masks as bit planes")


def sec_config3(it):
    """config 3: 2160 x 3840 's': invert (1 scatter, 18 B/px) and combine mode 1 (72 B/px stage sum)"""
    h, w = 2160, 3840
    n = h * w
    f2 = affine_flow([['scaling', 1000, 800, 0.9]], h, w, 's')
    f3 = affine_flow([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], h, w, 's')
    d2s, d3s = flow_sets(f2, 's', n), flow_sets(f3, 's', n)
    t = timed([(lambda d=d: d.invert()) for d in d2s], max(it, 12))
    yield entry("c3_invert_4k", "invert s->s (K3)", (h, w), 18, *t, kernel="scatter_walk_kernel",
                note="BASELINE config 3; certified mesh: one kernel, no synchronisation (reference: scipy griddata, 64 s at 1080p on 1 core, SURVEY 6)")
    def fresh(d):                    # a field seen for the first time: its mesh certificate (one pass + one 144-byte read-back) is part of the call
        d._certs.clear()
        return d.invert()
    t = timed([(lambda d=d: fresh(d)) for d in d2s], max(it, 12))
    yield entry("c3_invert_4k_fresh", "invert s->s of a field seen for the first time (certificate + K3)", (h, w), 18 + 9, *t,
                kernel="scatter_certify_kernel + scatter_walk_kernel", note="the certificate reads the field once more and writes one bit per cell; it is cached per field afterwards (c3_invert_4k)")
    t = timed([(lambda a=a, b=b: a.combine_with(b, 1)) for a, b in zip(d2s, d3s)], max(6, it // 2))
    yield entry("c3_mode1_s_4k", "combine_with mode 1 's' (K3 + 2 x K2 + epilogue)", (h, w), 72, *t,
                kernel="scatter_walk_kernel + 2 x compose3_xpose_kernel + axpy_kernel", note="BASELINE config 3")
    t = timed([(lambda d=d: d.switch_ref()) for d in d2s], max(it, 12))
    yield entry("c3_switch_ref_4k", "switch_ref s->t (K3)", (h, w), 18, *t, kernel="scatter_walk_kernel")
    del d2s, d3s
    f2t = flow_sets(affine_flow([['scaling', 1000, 800, 0.9]], h, w, 't'), 't', n)
    f3t = flow_sets(affine_flow([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], h, w, 't'), 't', n)
    t = timed([(lambda a=a, b=b: a.combine_with(b, 1)) for a, b in zip(f2t, f3t)], max(6, it // 2))
    yield entry("c3_mode1_t_4k", "combine_with mode 1 't' (4 x K3 + K2 + epilogues)", (h, w), 4 * 18 + 27 + 3 * 27, *t,
                kernel="dl_* (3 of the 4 scatters take the Delaunay path) + scatter_walk_kernel + compose3_xpose_kernel",
                note="flow_class.py:1383-1385: 4 scatters (the reference: 4 griddata calls, minutes each at this size)")


def delaunay_fields(h, w):
    """K3 on fields the certificate refuses -- name -> (vecs, point mask or None)"""
    f3v = affine_flow([['rotation', w / 2, h / 2, -20], ['scaling', w / 3.84, h / 2.7, 0.9]], h, w, 's')
    yy, xx = np.mgrid[:h, :w].astype(np.float32)

    class f3:
        vecs = f3v

    def obj():
        v = np.zeros((h, w, 2), np.float32)
        v[h // 4:h // 4 * 3, w // 4:w // 4 * 3] = [30.0, -12.0]
        return v, None

    def stripes():
        v = np.zeros((h, w, 2), np.float32)
        v[..., 0] = (np.floor(xx / 64) % 2) * 20.0
        return v, None

    def speckle():
        m = np.random.default_rng(0).random((h, w)) > 0.05
        m[0, 0] = False
        return f3.vecs, m

    def wobble():
        v = f3.vecs.copy()
        v[..., 0] += 3.0 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 131)
        v[..., 1] += 3.0 * np.cos(2 * np.pi * xx / 97) * np.sin(2 * np.pi * yy / 131)
        m = np.random.default_rng(0).random((h, w)) > 0.05
        m[0, 0] = False
        return v, m

    def hole():
        m = np.ones((h, w), bool)
        m[h // 4:h // 4 + h // 5, w // 4:w // 4 + w // 5] = False
        return f3.vecs, m

    return {"object": obj, "stripes": stripes, "speckle": speckle, "wobble": wobble, "hole": hole}


DELAUNAY_NOTES = {"object": "rigid object +30 px over a static background (motion boundary: folds and a tear)",
                  "stripes": "64-px stripes of alternating 20-px motion",
                  "speckle": "similarity field, 5 % random invalid points incl. a corner",
                  "wobble": "the same mask on a non-affine field (3-px sinusoid): no co-circular cells, a wavy border",
                  "hole": "similarity field with a 432 x 768 hole in the point mask"}


def sec_delaunay(it, which=None):
    h, w = 2160, 3840
    n = h * w
    for name, make in delaunay_fields(h, w).items():
        if which and name not in which:
            continue
        v, m = make()
        ds = flow_sets(v, 's', n, m)
        del v, m
        t = timed([(lambda d=d: d.invert()) for d in ds], max(it // 2, 2 * len(ds)))
        yield entry("k3_{}_4k".format(name), "invert s->s, {} (K3, Delaunay path)".format(name), (h, w), 18, *t,
                    kernel="dl_* (ofl_delaunay.hip: bins, stars, raster, resolve)", note=DELAUNAY_NOTES[name])
        del ds


def sec_resize(it):
    h, w = 2160, 3840
    n = h * w
    class f2:
        vecs = affine_flow([['scaling', 1000, 800, 0.9]], h, w, 's')
    for scale in (0.5, 2, 1.5):
        ho, wo = dev.resized_shape(h, w, scale, scale)
        ds = flow_sets(f2.vecs, 's', n, bpp=9.0 * (n + ho * wo) / n)
        t = timed([(lambda d=d: d.resize(scale)) for d in ds], max(it, 3 * len(ds)))
        n_eq = (h * w + ho * wo) * 9 / (h * w)
        yield entry("k6_resize_x{}".format(scale), "resize x{} (K6)".format(scale), (h, w), n_eq, *t, kernel="resize_flow_kernel",
                    note="{}x{} -> {}x{}; bytes = 9 B x (source + output px)".format(h, w, ho, wo))
        del ds


def config5_field():
    flo = of.load_sintel(os.path.join(ROOT, "tests", "golden", "sintel.flo"))
    return np.ascontiguousarray(np.tile(flo, (432, 384, 1)))


def sec_config5(it, bands=True):
    """config 5: Sintel .flo tiled to 4320 x 7680, apply to an RGB f32 image: 't' (gather) and 's' (scatter).  One working
    set is 1.1 GB -- four Infinity Caches -- so two sets rotate"""
    big = config5_field()
    h, w = big.shape[:2]
    img = np.random.default_rng(2).random((h, w, 3), dtype=np.float32)
    dimgs = copies(img, 2, dev.DeviceImage.from_host)
    dts = [dev.DeviceFlow.from_host(big, 't') for _ in range(2)]
    t = timed([(lambda im=im, d=d: dev.gather_bilinear(im, d.vecs, (h, w), -1, fmask=d.mask, want_valid=True))
               for im, d in zip(dimgs, dts)], max(6, it // 2))
    yield entry("c5_t_8k", "apply 't' RGB f32 + valid, tiled Sintel 4320x7680 (K1)", (h, w), 34, *t, kernel="gather2_kernel",
                note="BASELINE config 5, wrapped as 't'")
    dimg, dt = dimgs[0], dts[0]
    vals = dimg.buf
    out = dev.DeviceBuffer(h * w * 12)
    valid = dev.DeviceBuffer(h * w)
    info = []
    t = timed(lambda: info.append(dev.scatter_linear(dt.vecs, +1, None, vals, 3, None, h, w, None, out, valid, 0)), 4, warm=1)
    yield entry("c5_s_8k", "apply 's' RGB f32 + valid, tiled Sintel 4320x7680 (K3, Delaunay path)", (h, w), 34, *t,
                kernel="dl_star_near_kernel (largest of the dl_* family)",
                note="BASELINE config 5 as loaded ('s'); sites / unfinished / left over: {}".format([int(v) for v in info[-1]]))
    if not bands:
        return
    # the same field split over 8 GPUs: what ONE rank computes (rows of band 3; inputs replicated)
    from oflibnumpy_amd import sharding
    r0, r1 = sharding.row_band(h, 3, 8)
    ob, vb = dev.DeviceBuffer((r1 - r0) * w * 12), dev.DeviceBuffer((r1 - r0) * w)
    t = timed(lambda: dev.scatter_rows(dt.vecs, +1, None, vals, 3, None, h, w, r0, r1 - r0, ob, vb), 4, warm=1)
    yield entry("c5_s_band3of8_replicated", "apply 's', row band 3 of 8 of the same field, replicated stars (K3)", (r1 - r0, w), 34, *t,
                kernel="dl_star_near_kernel", note="one rank's share of config 5 ('s') on 8 GPUs without the slab exchange")
    fbs = copies(np.ascontiguousarray(big[r0:r1]), 2, dev.DeviceBuffer.from_host)
    t = timed([(lambda im=im, fb=fb: dev.gather_rows(im, r0, r1 - r0, fb, -1, want_valid=True)) for im, fb in zip(dimgs, fbs)], max(6, it // 2))
    yield entry("c5_t_band3of8", "apply 't', row band 3 of 8 of the same field (K1)", (r1 - r0, w), 33, *t, kernel="gather2_kernel",
                note="one rank's share of config 5 ('t') on 8 GPUs")


SECTIONS = [("config2", sec_config2), ("config3", sec_config3), ("config4", sec_config4), ("delaunay", sec_delaunay),
            ("resize", sec_resize), ("config5", sec_config5)]


def collect(iters=12, sections=("config2", "config3", "config4", "delaunay", "config5"), budget_s=75.0, log=None):
    """The entries bench.py appends to its line: key -> {ms, algorithmic_bytes, frac, kernel, rotating_sets, note}.  Sections
    that would start after `budget_s` seconds are skipped (and named in "_skipped")."""
    t0 = time.perf_counter()
    out, skipped = {}, []
    for name, fn in SECTIONS:
        if name not in sections:
            continue
        if time.perf_counter() - t0 > budget_s:
            skipped.append(name)
            continue
        try:
            gen = fn(iters, bands=False) if name == "config5" else fn(iters)
            for e in gen:
                # (compact: the driver keeps the TAIL of stdout, and this rides in the one JSON line; the verbose entry -- op,
                # shape, note -- goes to stderr, and tools/bench_ops.py prints it when run on its own)
                if e["key"].endswith("_bits") or e["key"].startswith(("c4_rot", "c3_switch")):
                    if log:
                        log(json.dumps(e))                      # (A/B and duplicate entries: stderr only, the line stays short)
                    continue
                out[e["key"]] = {"ms": e["device_ms"], "algorithmic_bytes": e["algorithmic_bytes"], "frac": e["frac_of_8TBps"],
                                 "kernel": e["kernel"].split(" (")[0].split(" + ")[0], "sets": e["rotating_sets"]}
                if log:
                    log(json.dumps(e))
        finally:
            dev.empty_cache()
    if skipped:
        out["_skipped"] = skipped
    out["_seconds"] = round(time.perf_counter() - t0, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="run only the sections / keys whose name contains one of these (config2, config3, config4, "
                                               "delaunay, resize, config5, or a key such as k3_hole)")
    args = ap.parse_args()
    nat.ensure_device()
    picks = [s for s in args.only.split(',') if s]
    for name, fn in SECTIONS:
        want_section = not picks or any(p in name for p in picks)
        want_keys = [p for p in picks if p.startswith(("c2_", "c3_", "c4_", "c5_", "k3_", "k6_"))]
        if not want_section and not want_keys:
            continue
        if name == "delaunay" and not want_section:
            which = [k[3:].split("_")[0] for k in want_keys if k.startswith("k3_")]
            if not which:
                continue
            gen = fn(args.iters, which)
        elif not want_section:
            continue
        else:
            gen = fn(args.iters)
        for e in gen:
            print(json.dumps(e), flush=True)
        dev.empty_cache()


if __name__ == "__main__":
    main()
