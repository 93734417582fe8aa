#!/usr/bin/env python
"""One operation in a loop for profiling: python tools/bench_invert.py [--op invert|switch|speckle|wobble|hole|stripes|object|config5] [--iters N]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev
from bench_ops import timed, report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="invert")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--size", type=int, nargs=2, default=[2160, 3840])
    args = ap.parse_args()
    of.native.ensure_device()
    h, w = args.size
    f = of.Flow.from_transforms([['rotation', w / 2, h / 2, -20], ['scaling', w / 3.84, h / 2.7, 0.9]], [h, w], 's')
    if args.op in ("invert", "switch"):
        d = of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], [h, w], 's').to_device()
    elif args.op == "invert_rot":                            # a certified mesh whose source cells lie along rotated rows
        d = f.to_device()
    elif args.op == "speckle":
        m = np.random.default_rng(0).random((h, w)) > 0.05
        d = dev.DeviceFlow.from_host(f.vecs, 's', m)
    elif args.op == "wobble":                                # the speckle mask on a non-affine field: no co-circular cells
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        v = f.vecs.copy()
        v[..., 0] += 3.0 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 131)
        v[..., 1] += 3.0 * np.cos(2 * np.pi * xx / 97) * np.sin(2 * np.pi * yy / 131)
        m = np.random.default_rng(0).random((h, w)) > 0.05
        d = dev.DeviceFlow.from_host(v, 's', m)
    elif args.op == "hole":
        m = np.ones((h, w), bool)
        m[h // 4:h // 4 + h // 5, w // 4:w // 4 + w // 5] = False
        d = dev.DeviceFlow.from_host(f.vecs, 's', m)
    elif args.op == "stripes":
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        v = np.zeros((h, w, 2), np.float32)
        v[..., 0] = (np.floor(xx / 64) % 2) * 20.0
        d = dev.DeviceFlow.from_host(v, 's')
    elif args.op == "config5":
        flo = of.load_sintel(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sintel.flo"))
        big = np.tile(flo, (h // flo.shape[0], w // flo.shape[1], 1))
        h, w = big.shape[:2]
        d = dev.DeviceFlow.from_host(big, 's')
    elif args.op == "cluster":                               # a 400 x 400 block contracted 100 times inside a static field: 160 000 distinct sites in 4 x 4 px
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        v = np.zeros((h, w, 2), np.float32)
        blk = (slice(700, 1100), slice(1500, 1900))
        ys, xs = yy[blk].astype(np.float64), xx[blk].astype(np.float64)
        rip = 1.0 + 0.05 * np.sin(xs / 5.3 + ys / 7.1)
        v[blk + (0,)] = (1700.37 + (xs - xs.mean()) / 100.0 * rip - xs).astype(np.float32)
        v[blk + (1,)] = (900.61 + (ys - ys.mean()) / 100.0 * rip - ys).astype(np.float32)
        d = dev.DeviceFlow.from_host(v, 's')
    elif args.op == "object":
        v = np.zeros((h, w, 2), np.float32)
        v[h // 4:h // 4 * 3, w // 4:w // 4 * 3] = [30.0, -12.0]
        d = dev.DeviceFlow.from_host(v, 's')
    else:
        raise SystemExit("unknown op")
    d.stats()
    fn = (lambda: d.switch_ref()) if args.op == "switch" else (lambda: d.invert())
    if args.op == "invert_rot":
        fn()
        assert d._certs and all(c.certified for c in d._certs.values()), "the rotated field was expected to take the certified walk"
    t = timed(fn, args.iters)
    report(args.op, (h, w), 18, *t)


if __name__ == "__main__":
    main()
