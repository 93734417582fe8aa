#!/usr/bin/env python
"""Slab-wise scatter (ofl_scatter_slab_stars_dev / _finish_dev) rehearsed on ONE GPU: every rank of a `world` is played in
turn with one workspace (step 1 of all ranks first -- the all-gather is a concatenation -- then step 1 + step 2 rank by
rank), the bands are compared bit for bit with the whole-field call, and step 1 + step 2 of one band are timed against it.

    python tools/slab_check.py [--op config5|speckle|hole|stripes|object|wobble] [--size H W] [--world 8] [--band 3]
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
from oflibnumpy_amd import _native as nat

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from slab_util import Slab, probe_values


def field(op, h, w):
    f = of.Flow.from_transforms([['rotation', w / 2, h / 2, -20], ['scaling', w / 3.84, h / 2.7, 0.9]], [h, w], 's')
    m = None
    if op == "speckle":
        v, m = f.vecs, np.random.default_rng(0).random((h, w)) > 0.05
    elif op == "wobble":
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        v = f.vecs.copy()
        v[..., 0] += 3.0 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 131)
        v[..., 1] += 3.0 * np.cos(2 * np.pi * xx / 97) * np.sin(2 * np.pi * yy / 131)
        m = np.random.default_rng(0).random((h, w)) > 0.05
    elif op == "hole":
        v, m = f.vecs, np.ones((h, w), bool)
        m[h // 4:h // 4 + h // 5, w // 4:w // 4 + w // 5] = False
    elif op == "stripes":
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        v = np.zeros((h, w, 2), np.float32)
        v[..., 0] = (np.floor(xx / 64) % 2) * 20.0
    elif op == "object":
        v = np.zeros((h, w, 2), np.float32)
        v[h // 4:h // 4 * 3, w // 4:w // 4 * 3] = [30.0, -12.0]
    elif op == "config5":
        flo = of.load_sintel(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sintel.flo"))
        v = np.ascontiguousarray(np.tile(flo, (h // flo.shape[0], w // flo.shape[1], 1)))
    else:
        raise SystemExit("unknown op")
    return np.ascontiguousarray(v, np.float32), m


def wall_ms(fn, iters=5):
    lib = nat.load()
    fn()
    nat.check(lib.ofl_stream_sync(None))
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    nat.check(lib.ofl_stream_sync(None))
    return (time.perf_counter() - t0) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="config5")
    ap.add_argument("--size", type=int, nargs=2, default=[4320, 7680])
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--band", type=int, default=3)
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--profile", action="store_true", help="only the gathered lists and 5 x (step 1 + step 2) of the band: for rocprofv3")
    args = ap.parse_args()
    of.native.ensure_device()
    h, w = args.size
    vecs, m = field(args.op, h, w)
    h, w = vecs.shape[:2]
    sl = Slab(vecs, m, probe_values(h, w))
    res = {"op": args.op, "shape": [h, w], "world": args.world}
    if not args.no_check:
        fo, fv, info = sl.full()
        fo = fo.to_host((h, w, 2), np.float32)
        fv = fv.to_host((h, w), np.uint8)
        out, valid, lists, bands = sl.play(args.world)
        counts = sl.counts(lists, args.world)[0]
        res.update({"kept_unfinished_left": list(map(int, info)), "unfinished_per_rank": counts,
                    "bands_equal_full": bool(np.array_equal(out.view(np.uint32), fo.view(np.uint32)) and np.array_equal(valid, fv)),
                    "valid_fraction": float(fv.mean())})
        if not res["bands_equal_full"]:
            d = (out.view(np.uint32) != fo.view(np.uint32)).any(-1) | (valid != fv)
            ys = np.nonzero(d.any(1))[0]
            res["differing_nodes"] = int(d.sum())
            res["differing_rows"] = [int(ys.min()), int(ys.max())]
            res["bands"] = bands
    else:
        bands = sl.bands(args.world)
        lists = sl.gather(bands)
    r0, r1 = bands[args.band]
    o, v = dev.DeviceBuffer((r1 - r0) * w * 2 * 4), dev.DeviceBuffer((r1 - r0) * w)
    scratch = dev.DeviceBuffer(sl.nb)
    if args.profile:
        t_ab = wall_ms(lambda: (sl.stars(r0, r1 - r0, scratch.ptr), sl.finish(r0, r1 - r0, lists, args.world, o, v)))
        print(json.dumps({"op": args.op, "band": [r0, r1], "slab_band_ms": round(t_ab, 3)}))
        return
    t_full = wall_ms(lambda: sl.full())
    t_a = wall_ms(lambda: sl.stars(r0, r1 - r0, scratch.ptr))
    t_ab = wall_ms(lambda: (sl.stars(r0, r1 - r0, scratch.ptr), sl.finish(r0, r1 - r0, lists, args.world, o, v)))
    t_rows = wall_ms(lambda: sl.full(r0, r1 - r0))
    res.update({"band": [r0, r1], "full_ms": round(t_full, 3), "replicated_band_ms": round(t_rows, 3), "slab_step1_ms": round(t_a, 3),
                "slab_band_ms": round(t_ab, 3), "slab_over_full": round(t_ab / t_full, 3)})
    print(json.dumps(res))


if __name__ == "__main__":
    main()
