#!/usr/bin/env python
"""K1 (general bilinear gather) over dtypes and sampling patterns: python tools/bench_k1.py [--iters 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev
from bench_ops import timed, report, n_sets, copies

nat = of.native


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--dtypes", default="u8,u16,f32")
    ap.add_argument("--patterns", default="scale,shift,rot30")
    ap.add_argument("--sizes", default="1080,2160")
    args = ap.parse_args()
    nat.ensure_device()
    for h, w in [(int(v), int(v) * 16 // 9) for v in args.sizes.split(',')]:
        flows = {
            "scale": of.Flow.from_transforms([['scaling', w / 2, h / 2, 0.8]], [h, w], 't'),
            "shift": of.Flow.from_transforms([['translation', 3.3, -2.6]], [h, w], 't'),
            "rot30": of.Flow.from_transforms([['rotation', w / 2, h / 2, -30]], [h, w], 't'),
        }
        rng = np.random.default_rng(1)
        img = rng.random((h, w, 3), dtype=np.float32)
        tmask = (rng.random((h, w)) > 0.1).astype(np.uint8)
        for name, f in flows.items():
            if name not in args.patterns.split(','):
                continue
            for dt, bpp in (("u8", 3), ("u16", 6), ("f32", 12)):
                if dt not in args.dtypes.split(","):
                    continue
                arr = {"u8": (img * 255).astype(np.uint8), "u16": (img * 65535).astype(np.uint16), "f32": img}[dt]
                # distinct flow / image / mask / result buffers per launch, >= 3 Infinity Caches in total (bench_ops.n_sets)
                k = n_sets((8 + 2 * bpp) * h * w)
                ds = [f.to_device() for _ in range(k)]
                dis = copies(arr, k, dev.DeviceImage.from_host)
                tms = copies(tmask, k, dev.DeviceBuffer.from_host)
                kw = dict(arith=nat.ARITH_NATIVE, rule=nat.RULE_GE_HALF) if dt == "u8" else (dict(rule=nat.RULE_GT_HALF) if dt == "u16" else {})
                t = timed([(lambda d=d, di=di, tm=tm: dev.gather_bilinear(di, d.vecs, (h, w), -1, smask=tm, fmask=d.mask, want_valid=True, **kw))
                           for d, di, tm in zip(ds, dis, tms)], max(args.iters, 3 * k))
                report("K1 RGB {} + target mask + valid, {}".format(dt, name), (h, w), 8 + 1 + 2 * bpp + 2, *t)
                t = timed([(lambda d=d, di=di: dev.gather_bilinear(di, d.vecs, (h, w), -1, **({a: v for a, v in kw.items() if a == 'arith'})))
                           for d, di in zip(ds, dis)], max(args.iters, 3 * k))
                report("K1 RGB {} image only, {}".format(dt, name), (h, w), 8 + 2 * bpp, *t)
                del ds, dis, tms

if __name__ == "__main__":
    main()
