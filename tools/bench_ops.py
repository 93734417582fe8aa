#!/usr/bin/env python
"""Secondary measurements (not the driver's bench.py): the other BASELINE.json configs, device-resident,
HIP-event timed.  Prints one JSON object per operation with the algorithmic bytes of SURVEY.md 8(d).

Every operation ROTATES over distinct input AND output buffers whose footprints sum to at least 3 x 256 MiB (the
Infinity Cache), exactly as bench.py does for the headline kernel: round 2 timed one resident working set in a loop,
and the rates of everything that fits the cache were cache rates.

    python tools/bench_ops.py [--iters 20] [--only substring]
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    return ms.value / iters * 1e-3, wall, len(fns)


def report(name, shape, bytes_per_px, dev_s, wall_s, sets=1, note="", units=1):
    n = shape[0] * shape[1] * units
    print(json.dumps({"op": name, "shape": list(shape), "algorithmic_bytes": bytes_per_px * n,
                      "device_ms": round(dev_s * 1e3, 4), "wall_ms": round(wall_s * 1e3, 4),
                      "GBps_algorithmic": round(bytes_per_px * n / dev_s / 1e9, 1),
                      "frac_of_8TBps": round(bytes_per_px * n / dev_s / 8e12, 4), "rotating_sets": sets, "note": note}), flush=True)


def copies(arr, n, wrap):
    """n device copies of one host array (the rotation is about where the bytes live, not about their values)"""
    return [wrap(arr) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="", help="run only the sections whose name contains this (config2, config3, config4, delaunay, resize, config5)")
    args = ap.parse_args()
    nat.ensure_device()
    it = args.iters
    lib = nat.load()
    want = lambda sec: (not args.only) or any(s in sec for s in args.only.split(','))

    if want("config2"):
        # config 2: 1080 x 1920 't' flow applied to an RGB float32 image with valid area (34 B/px)
        h, w = 1080, 1920
        n = h * w
        f1 = of.Flow.from_transforms([['rotation', 960, 540, -30]], [h, w], 't')
        img = np.random.default_rng(1).random((h, w, 3), dtype=np.float32)
        k = n_sets(34 * n)
        flows = [f1.to_device() for _ in range(k)]
        imgs = copies(img, k, dev.DeviceImage.from_host)
        t = timed([(lambda d=d, im=im: dev.gather_bilinear(im, d.vecs, (h, w), -1, fmask=d.mask, want_valid=True))
                   for d, im in zip(flows, imgs)], max(it, 3 * k))
        report("apply 't' RGB f32 + valid (K1)", (h, w), 34, *t, note="BASELINE config 2; rotated sampling pattern")
        k = n_sets(17 * n)
        flows = [f1.to_device() for _ in range(k)]
        u8 = copies((img * 255).astype(np.uint8), k, dev.DeviceImage.from_host)
        tms = copies((np.random.default_rng(3).random((h, w)) > 0.1).astype(np.uint8), k, dev.DeviceBuffer.from_host)
        t = timed([(lambda d=d, im=im, tm=tm: dev.gather_bilinear(im, d.vecs, (h, w), -1, smask=tm, fmask=d.mask, want_valid=True,
                                                                   arith=nat.ARITH_NATIVE, rule=nat.RULE_GE_HALF))
                   for d, im, tm in zip(flows, u8, tms)], max(it, 3 * k))
        report("apply 't' RGB uint8 + target mask + valid (K1)", (h, w), 17, *t, note="8 + 1 flow, 3 + 3 image, 1 + 1 masks")
        del flows, imgs, u8, tms

    if want("config4"):
        # config 4, one GPU's share: 32 independent 1080 x 1920 pairs in ONE launch of the fused compose kernel; pair i =
        # rotation by -30 + 60 i / 255 degrees about the centre (+) translation (40 cos i, 40 sin i): the SAMPLING field of
        # mode 3 / 't' is the translation (flow_class.py:1422)
        h, w, B = 1080, 1920, 32
        n = h * w
        sets = []
        for s in range(2):                                   # 2 x 1.79 GB: every launch streams from HBM
            va, ma, vb, mb = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n)
            for i in range(B):
                j = i + 32 * s
                fa = of.Flow.from_transforms([['rotation', w / 2, h / 2, -30 + 60 * j / 255]], [h, w], 't')
                fb = of.Flow.from_transforms([['translation', 40 * math.cos(j), 40 * math.sin(j)]], [h, w], 't')
                m = (np.random.default_rng(j).random((h, w)) > 0.05).astype(np.uint8)
                for dst, a in ((va.ptr + i * n * 8, fa.vecs), (vb.ptr + i * n * 8, fb.vecs), (ma.ptr + i * n, m), (mb.ptr + i * n, m)):
                    a = np.ascontiguousarray(a)
                    nat.check(lib.ofl_upload(dst, a.ctypes.data, a.nbytes, None))
                    nat.check(lib.ofl_stream_sync(None))
            sets.append((va, ma, vb, mb, dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer.zeros(32 * B)))
        t = timed([(lambda s=s: nat.check(lib.ofl_compose3_dev(s[0].ptr, s[1].ptr, s[2].ptr, s[3].ptr, -1, h, w, B, s[4].ptr, s[5].ptr,
                                                                s[6].ptr, 0, None))) for s in sets], max(it, 10))
        report("combine_with mode 3 't', 32 pairs of 1080p in one launch (K2)", (h, w), 27, *t, units=B,
               note="BASELINE config 4, one GPU's share of the 256 pairs; sampling field = translation of 40 px")
        del sets

    h, w = 2160, 3840
    n = h * w
    if want("config3") or want("delaunay") or want("resize"):
        f2 = of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], [h, w], 's')
        f3 = of.Flow.from_transforms([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], [h, w], 's')

    def flow_sets(vecs, ref, mask=None, bpp=18):
        ds = [dev.DeviceFlow.from_host(vecs, ref, mask) for _ in range(n_sets(bpp * n))]
        for d in ds:
            d.stats()
        return ds

    if want("config3"):
        # config 3: 2160 x 3840 's': invert (1 scatter, 18 B/px) and combine mode 1 (72 B/px stage sum)
        d2s, d3s = flow_sets(f2.vecs, 's'), flow_sets(f3.vecs, 's')
        t = timed([(lambda d=d: d.invert()) for d in d2s], max(it, 12))
        report("invert s->s (K3)", (h, w), 18, *t, note="BASELINE config 3; certified mesh: one kernel, no synchronisation (reference: scipy griddata, 64 s at 1080p on 1 core, SURVEY 6)")
        t = timed([(lambda a=a, b=b: a.combine_with(b, 1)) for a, b in zip(d2s, d3s)], max(6, it // 2))
        report("combine_with mode 1 's' (K3 + 2 x K1 + epilogues)", (h, w), 72, *t, note="BASELINE config 3")
        t = timed([(lambda d=d: d.switch_ref()) for d in d2s], max(it, 12))
        report("switch_ref s->t (K3)", (h, w), 18, *t)
        del d2s, d3s
        f2t = flow_sets(of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], [h, w], 't').vecs, 't')
        f3t = flow_sets(of.Flow.from_transforms([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], [h, w], 't').vecs, 't')
        t = timed([(lambda a=a, b=b: a.combine_with(b, 1)) for a, b in zip(f2t, f3t)], max(6, it // 2))
        report("combine_with mode 1 't' (4 x K3 + K1 + epilogues)", (h, w), 4 * 18 + 27 + 3 * 27, *t,
               note="flow_class.py:1383-1385: 4 scatters (the reference: 4 griddata calls, minutes each at this size)")
        del f2t, f3t

    if want("delaunay"):
        # K3 on discontinuous fields (motion boundaries fold and stretch cells): a rigid object moving 30 px over a
        # static background, and 64-px stripes of alternating 20-px motion
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        obj = np.zeros((h, w, 2), np.float32)
        obj[600:1500, 1000:2600] = [30.0, -12.0]
        stripes = np.zeros((h, w, 2), np.float32)
        stripes[..., 0] = (np.floor(xx / 64) % 2) * 20.0
        for name, v in (("rigid object +30 px", obj), ("64-px stripes of 20-px motion", stripes)):
            ds = flow_sets(v, 's')
            t = timed([(lambda d=d: d.invert()) for d in ds], max(it, 12))
            report("invert s->s, {} (K3)".format(name), (h, w), 18, *t, note="discontinuous field")
            del ds
        del yy, xx, obj, stripes
        spk = np.random.default_rng(0).random((h, w)) > 0.05
        spk[0, 0] = False
        ds = flow_sets(f3.vecs, 's', spk)
        t = timed([(lambda d=d: d.invert()) for d in ds], max(it, 12))
        report("invert s->s, 5 % random invalid points incl. a corner (K3)", (h, w), 18, *t, note="points dropped (consider_mask): Delaunay path")
        del ds
        # the same with SURVEY 8(d)'s non-affine term on top: no cell of this mesh is co-circular (the similarity transforms and
        # lattices above are, cell by cell -- every decision of theirs is a tie), and the certificate fails here and there
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        wob = f3.vecs.copy()
        wob[..., 0] += 3.0 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 131)
        wob[..., 1] += 3.0 * np.cos(2 * np.pi * xx / 97) * np.sin(2 * np.pi * yy / 131)
        del yy, xx
        ds = flow_sets(wob, 's', spk)
        t = timed([(lambda d=d: d.invert()) for d in ds], max(it, 12))
        report("invert s->s, the same mask on a non-affine field (3-px sinusoid) (K3)", (h, w), 18, *t, note="generic positions: Delaunay path")
        del wob, ds
        hole = np.ones((h, w), bool)
        hole[500:900, 1000:1800] = False
        ds = flow_sets(f3.vecs, 's', hole)
        t = timed([(lambda d=d: d.invert()) for d in ds], max(it, 12))
        report("invert s->s, 400 x 800 hole in the point mask (K3)", (h, w), 18, *t, note="hole triangulated like SciPy does: Delaunay path")
        del ds

    if want("resize"):
        # K6: Flow.resize of the 4K field: 9 B per source px read + 9 B per output px written
        for scale in (0.5, 2, 1.5):
            ho, wo = dev.resized_shape(h, w, scale, scale)
            ds = flow_sets(f2.vecs, 's', bpp=9.0 * (n + ho * wo) / n)
            t = timed([(lambda d=d: d.resize(scale)) for d in ds], max(it, 3 * len(ds)))
            n_eq = (h * w + ho * wo) * 9 / (h * w)
            report("resize x{} (K6)".format(scale), (h, w), n_eq, *t, note="{}x{} -> {}x{}; bytes = 9 B x (source + output px)".format(h, w, ho, wo))
            del ds

    if want("config5"):
        # config 5: Sintel .flo tiled to 4320 x 7680, apply to an RGB f32 image: 't' (gather) and 's' (scatter).  One working
        # set is 1.1 GB -- four Infinity Caches -- so two sets rotate
        flo = of.load_sintel(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sintel.flo"))
        big = np.tile(flo, (432, 384, 1))
        h, w = big.shape[:2]
        img = np.random.default_rng(2).random((h, w, 3), dtype=np.float32)
        dimgs = copies(img, 2, dev.DeviceImage.from_host)
        dts = [dev.DeviceFlow.from_host(big, 't') for _ in range(2)]
        t = timed([(lambda im=im, d=d: dev.gather_bilinear(im, d.vecs, (h, w), -1, fmask=d.mask, want_valid=True))
                   for im, d in zip(dimgs, dts)], max(6, it // 2))
        report("apply 't' RGB f32 + valid, tiled Sintel 4320x7680 (K1)", (h, w), 34, *t, note="BASELINE config 5, wrapped as 't'")
        dimg, dt = dimgs[0], dts[0]
        vals = dimg.buf
        out = dev.DeviceBuffer(h * w * 12)
        valid = dev.DeviceBuffer(h * w)
        info = []
        t = timed(lambda: info.append(dev.scatter_linear(dt.vecs, +1, None, vals, 3, None, h, w, None, out, valid, 0)), 4, warm=1)
        report("apply 's' RGB f32 + valid, tiled Sintel 4320x7680 (K3)", (h, w), 34, *t,
               note="BASELINE config 5 as loaded ('s'); large triangles: {}".format(info[-1][1:]))
        # the same field split over 8 GPUs: what ONE rank computes (rows of band 3; inputs replicated)
        from oflibnumpy_amd import sharding
        r0, r1 = sharding.row_band(h, 3, 8)
        ob, vb = dev.DeviceBuffer((r1 - r0) * w * 12), dev.DeviceBuffer((r1 - r0) * w)
        t = timed(lambda: dev.scatter_rows(dt.vecs, +1, None, vals, 3, None, h, w, r0, r1 - r0, ob, vb), 4, warm=1)
        report("apply 's', row band 3 of 8 of the same field (K3)", (r1 - r0, w), 34, *t, note="one rank's share of config 5 ('s') on 8 GPUs")
        fbs = copies(np.ascontiguousarray(big[r0:r1]), 2, dev.DeviceBuffer.from_host)
        t = timed([(lambda im=im, fb=fb: dev.gather_rows(im, r0, r1 - r0, fb, -1, want_valid=True)) for im, fb in zip(dimgs, fbs)], max(6, it // 2))
        report("apply 't', row band 3 of 8 of the same field (K1)", (r1 - r0, w), 33, *t, note="one rank's share of config 5 ('t') on 8 GPUs")


if __name__ == "__main__":
    main()
