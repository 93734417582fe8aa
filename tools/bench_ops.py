#!/usr/bin/env python
"""Secondary measurements (not the driver's bench.py): the other BASELINE.json configs, device-resident,
HIP-event timed.  Prints one JSON object per operation with the algorithmic bytes of SURVEY.md 8(d).

    python tools/bench_ops.py [--iters 20]
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev

nat = of.native


def timed(fn, iters, warm=3):
    lib = nat.load()
    for _ in range(warm):
        fn()
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    nat.check(lib.ofl_event_create(ctypes.byref(e0)))
    nat.check(lib.ofl_event_create(ctypes.byref(e1)))
    nat.check(lib.ofl_device_sync())
    t0 = time.perf_counter()
    nat.check(lib.ofl_event_record(e0, None))
    for _ in range(iters):
        fn()
    nat.check(lib.ofl_event_record(e1, None))
    nat.check(lib.ofl_device_sync())
    wall = (time.perf_counter() - t0) / iters
    ms = ctypes.c_float()
    nat.check(lib.ofl_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    return ms.value / iters * 1e-3, wall


def report(name, shape, bytes_per_px, dev_s, wall_s, note=""):
    n = shape[0] * shape[1]
    print(json.dumps({"op": name, "shape": list(shape), "algorithmic_bytes": bytes_per_px * n,
                      "device_ms": round(dev_s * 1e3, 4), "wall_ms": round(wall_s * 1e3, 4),
                      "GBps_algorithmic": round(bytes_per_px * n / dev_s / 1e9, 1),
                      "frac_of_8TBps": round(bytes_per_px * n / dev_s / 8e12, 4), "note": note}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    nat.ensure_device()
    it = args.iters

    # config 2: 1080 x 1920 't' flow applied to an RGB float32 image with valid area (34 B/px)
    h, w = 1080, 1920
    f1 = of.Flow.from_transforms([['rotation', 960, 540, -30]], [h, w], 't')
    img = np.random.default_rng(1).random((h, w, 3), dtype=np.float32)
    d1 = f1.to_device()
    dimg = dev.DeviceImage.from_host(img)
    t = timed(lambda: dev.gather_bilinear(dimg, d1.vecs, (h, w), -1, fmask=d1.mask, want_valid=True), it)
    report("apply 't' RGB f32 + valid (K1)", (h, w), 34, *t, note="BASELINE config 2; rotated sampling pattern")

    u8 = dev.DeviceImage.from_host((img * 255).astype(np.uint8))
    tm = dev.DeviceBuffer.from_host((np.random.default_rng(3).random((h, w)) > 0.1).astype(np.uint8))
    t = timed(lambda: dev.gather_bilinear(u8, d1.vecs, (h, w), -1, smask=tm, fmask=d1.mask, want_valid=True,
                                          arith=nat.ARITH_NATIVE, rule=nat.RULE_GE_HALF), it)
    report("apply 't' RGB uint8 + target mask + valid (K1)", (h, w), 17, *t, note="8 + 1 flow, 3 + 3 image, 1 + 1 masks")

    # config 3: 2160 x 3840 's': invert (1 scatter, 18 B/px) and combine mode 1 (72 B/px stage sum)
    h, w = 2160, 3840
    f2 = of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], [h, w], 's')
    f3 = of.Flow.from_transforms([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], [h, w], 's')
    d2, d3 = f2.to_device(), f3.to_device()
    d2.stats(); d3.stats()
    t = timed(lambda: d2.invert(), it)
    report("invert s->s (K3)", (h, w), 18, *t, note="BASELINE config 3; certified mesh: one kernel, no synchronisation (reference: scipy griddata, 64 s at 1080p on 1 core, SURVEY 6)")
    t = timed(lambda: d2.combine_with(d3, 1), max(3, it // 4))
    report("combine_with mode 1 's' (K3 + 2 x K1 + epilogues)", (h, w), 72, *t, note="BASELINE config 3")
    t = timed(lambda: d2.switch_ref(), it)
    report("switch_ref s->t (K3)", (h, w), 18, *t)
    f2t = of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], [h, w], 't').to_device()
    f3t = of.Flow.from_transforms([['rotation', 1920, 1080, -20], ['scaling', 1000, 800, 0.9]], [h, w], 't').to_device()
    f2t.stats(); f3t.stats()
    t = timed(lambda: f2t.combine_with(f3t, 1), max(3, it // 4))
    report("combine_with mode 1 't' (4 x K3 + K1 + epilogues)", (h, w), 4 * 18 + 27 + 3 * 27, *t,
           note="flow_class.py:1383-1385: 4 scatters (the reference: 4 griddata calls, minutes each at this size)")

    # K3 on discontinuous fields (motion boundaries fold and stretch cells): a rigid object moving 30 px over a
    # static background, and 64-px stripes of alternating 20-px motion
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    obj = np.zeros((h, w, 2), np.float32)
    obj[600:1500, 1000:2600] = [30.0, -12.0]
    stripes = np.zeros((h, w, 2), np.float32)
    stripes[..., 0] = (np.floor(xx / 64) % 2) * 20.0
    for name, v in (("rigid object +30 px", obj), ("64-px stripes of 20-px motion", stripes)):
        dd = dev.DeviceFlow.from_host(v, 's')
        dd.stats()
        t = timed(lambda: dd.invert(), it)
        report("invert s->s, {} (K3)".format(name), (h, w), 18, *t, note="discontinuous field")
    del yy, xx, obj, stripes
    spk = np.random.default_rng(0).random((h, w)) > 0.05
    spk[0, 0] = False
    ds = dev.DeviceFlow.from_host(f3.vecs, 's', spk)
    ds.stats()
    t = timed(lambda: ds.invert(), it)
    report("invert s->s, 5 % random invalid points incl. a corner (K3)", (h, w), 18, *t, note="points dropped (consider_mask): Delaunay path")
    # the same with SURVEY 8(d)'s non-affine term on top: no cell of this mesh is co-circular (the similarity transforms and
    # lattices above are, cell by cell -- every decision of theirs is a tie), and the certificate fails here and there
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    wob = f3.vecs.copy()
    wob[..., 0] += 3.0 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 131)
    wob[..., 1] += 3.0 * np.cos(2 * np.pi * xx / 97) * np.sin(2 * np.pi * yy / 131)
    del yy, xx
    dw = dev.DeviceFlow.from_host(wob, 's', spk)
    dw.stats()
    t = timed(lambda: dw.invert(), it)
    report("invert s->s, the same mask on a non-affine field (3-px sinusoid) (K3)", (h, w), 18, *t, note="generic positions: Delaunay path")
    del wob, dw
    hole = np.ones((h, w), bool)
    hole[500:900, 1000:1800] = False
    dh = dev.DeviceFlow.from_host(f3.vecs, 's', hole)
    dh.stats()
    t = timed(lambda: dh.invert(), it)
    report("invert s->s, 400 x 800 hole in the point mask (K3)", (h, w), 18, *t, note="hole triangulated like SciPy does: Delaunay path")

    # K6: Flow.resize of the 4K field: 9 B per source px read + 9 B per output px written
    for scale in (0.5, 2, 1.5):
        ho, wo = dev.resized_shape(h, w, scale, scale)
        t = timed(lambda: d2.resize(scale), it)
        n_eq = (h * w + ho * wo) * 9 / (h * w)
        report("resize x{} (K6)".format(scale), (h, w), n_eq, *t, note="{}x{} -> {}x{}; bytes = 9 B x (source + output px)".format(h, w, ho, wo))

    # config 5: Sintel .flo tiled to 4320 x 7680, apply to an RGB f32 image: 't' (gather) and 's' (scatter)
    flo = of.load_sintel(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sintel.flo"))
    big = np.tile(flo, (432, 384, 1))
    h, w = big.shape[:2]
    img = np.random.default_rng(2).random((h, w, 3), dtype=np.float32)
    dimg = dev.DeviceImage.from_host(img)
    dt = dev.DeviceFlow.from_host(big, 't')
    t = timed(lambda: dev.gather_bilinear(dimg, dt.vecs, (h, w), -1, fmask=dt.mask, want_valid=True), max(3, it // 2))
    report("apply 't' RGB f32 + valid, tiled Sintel 4320x7680 (K1)", (h, w), 34, *t, note="BASELINE config 5, wrapped as 't'")
    vals = dimg.buf
    out = dev.DeviceBuffer(h * w * 12)
    valid = dev.DeviceBuffer(h * w)
    info = []
    t = timed(lambda: info.append(dev.scatter_linear(dt.vecs, +1, None, vals, 3, None, h, w, None, out, valid, 0)), 3, warm=1)
    report("apply 's' RGB f32 + valid, tiled Sintel 4320x7680 (K3)", (h, w), 34, *t,
           note="BASELINE config 5 as loaded ('s'); large triangles: {}".format(info[-1][1:]))
    # the same field split over 8 GPUs: what ONE rank computes (rows of band 3; inputs replicated)
    from oflibnumpy_amd import sharding
    r0, r1 = sharding.row_band(h, 3, 8)
    ob, vb = dev.DeviceBuffer((r1 - r0) * w * 12), dev.DeviceBuffer((r1 - r0) * w)
    t = timed(lambda: dev.scatter_rows(dt.vecs, +1, None, vals, 3, None, h, w, r0, r1 - r0, ob, vb), 3, warm=1)
    report("apply 's', row band 3 of 8 of the same field (K3)", (r1 - r0, w), 34, *t, note="one rank's share of config 5 ('s') on 8 GPUs")
    fb = dev.DeviceBuffer.from_host(np.ascontiguousarray(big[r0:r1]))
    t = timed(lambda: dev.gather_rows(dimg, r0, r1 - r0, fb, -1, want_valid=True), max(3, it // 2))
    report("apply 't', row band 3 of 8 of the same field (K1)", (r1 - r0, w), 33, *t, note="one rank's share of config 5 ('t') on 8 GPUs")


if __name__ == "__main__":
    main()
