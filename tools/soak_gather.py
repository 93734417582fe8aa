#!/usr/bin/env python
"""Soak of the bilinear-gather family (K1 apply 't', K2 combine_with mode 3, both refs, K6 resize) against the CPU oracle, BIT FOR BIT:
random shapes (odd widths, partial tiles, single rows), flows that mix exact whole / half / 1-32nd pixels, sub-threshold,
ordinary, rotated (the transposed-gather path) and far-out-of-range vectors, random masks, 1 - 5 channels, uint8 / int16 /
uint16 / float32 / float64 images, with and without a target mask.

    python tools/soak_gather.py [--seconds 120] [--seed 0] [--max 200 520]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def one_case(of, O, seed, hmax, wmax):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(1, hmax)), int(rng.integers(1, wmax))
    yy, xx = np.mgrid[:h, :w].astype(np.float64)
    kind = rng.integers(0, 5, size=(h, w, 1))
    base = rng.standard_normal((h, w, 2)) * rng.choice([0.3, 3.0, 40.0])
    v = np.where(kind == 0, np.round(base * 2) / 2,                          # exact half / whole pixels
        np.where(kind == 1, base,
        np.where(kind == 2, rng.uniform(-9e-4, 9e-4, (h, w, 2)),
        np.where(kind == 3, np.round(base * 32) / 32 + rng.choice([0.0, 1 / 64, -1 / 64]), base * 1e3))))   # 1/32-px snapping boundaries
    style = int(rng.integers(0, 6))
    if style == 0:                                                           # a rotation: the transposed gather
        a = rng.uniform(-1.2, 1.2)
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        v = np.stack([(np.cos(a) - 1) * (xx - cx) - np.sin(a) * (yy - cy), np.sin(a) * (xx - cx) + (np.cos(a) - 1) * (yy - cy)], -1)
    elif style == 1:
        v = rng.uniform(-9e-4, 9e-4, (h, w, 2))                              # thresholded-zero sampling field
    elif style == 2:
        v = np.zeros((h, w, 2))
    v = v.astype(np.float32)
    u = (rng.standard_normal((h, w, 2)) * 5).astype(np.float32)
    m1, m2 = rng.random((h, w)) > rng.choice([0.0, 0.2, 0.7]), rng.random((h, w)) > rng.choice([0.0, 0.2])
    bad, msgs, n = 0, [], 0
    for ref in ('t', 's'):
        a_, b_ = (u, v) if ref == 't' else (v, u)                            # v always plays the sampling field
        got = of.Flow(a_, ref, m1).combine_with(of.Flow(b_, ref, m2), 3)
        want = O.OFlow(a_, ref, m1).combine_with(O.OFlow(b_, ref, m2), 3)
        d = int((got.vecs.view(np.uint32) != want.vecs.view(np.uint32)).any(-1).sum() + (got.mask != want.mask).sum())
        n += h * w
        if d:
            bad += d
            msgs.append("compose ref {} {}x{} style {}: {} px".format(ref, h, w, style, d))
    c = int(rng.integers(1, 6))
    dt = rng.choice([np.uint8, np.int16, np.uint16, np.float32, np.float64])
    img = (rng.random((h, w, c)) * (255 if dt == np.uint8 else 30000)).astype(dt)
    tm = m2 if rng.random() < 0.6 else None
    if dt == np.uint16 and tm is None:
        tm = m2                                                              # (uint16 with the default int8 mask: the reference cannot either)
    try:
        ow, ov = O.OFlow(v, 't', m1).apply(img, tm, return_valid_area=True)
        gw, gv = of.Flow(v, 't', m1).apply(img, tm, return_valid_area=True)
        same = gw.dtype == ow.dtype and np.array_equal(gw.view(np.uint8), ow.view(np.uint8)) and np.array_equal(gv, ov)
        n += h * w
        if not same:
            d = int((gw != ow).any(-1).sum() + (gv != ov).sum()) if gw.shape == ow.shape else -1
            bad += max(d, 1)
            msgs.append("apply {} c{} {}x{} style {} mask {}: {} px".format(np.dtype(dt).name, c, h, w, style, tm is not None, d))
    except (TypeError, ValueError) as e:
        # the oracle and the product refuse the same inputs
        try:
            of.Flow(v, 't', m1).apply(img, tm, return_valid_area=True)
            bad += 1
            msgs.append("oracle refused, product accepted: " + str(e)[:80])
        except (TypeError, ValueError):
            pass
    # resize (K6): cv2.resize semantics, any scale pair that leaves at least one pixel
    if h >= 2 and w >= 2:
        fy, fx = float(rng.choice([0.3, 0.5, 0.77, 1.0, 1.5, 2.0, 3.1])), float(rng.choice([0.4, 0.5, 0.9, 1.0, 1.3, 2.0, 2.6]))
        if round(h * fy) >= 1 and round(w * fx) >= 1:
            gr = of.Flow(v, 't', m1).resize((fy, fx))
            wr = O.OFlow(v, 't', m1).resize((fy, fx))
            n += gr.vecs.shape[0] * gr.vecs.shape[1]
            if gr.vecs.shape != wr.vecs.shape or not (np.array_equal(gr.vecs.view(np.uint32), wr.vecs.view(np.uint32)) and np.array_equal(gr.mask, wr.mask)):
                bad += 1
                msgs.append("resize {}x{} by ({}, {})".format(h, w, fy, fx))
    return n, bad, msgs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max", type=int, nargs=2, default=[200, 520])
    args = ap.parse_args()
    import oflibnumpy_amd as of
    from oracle import np_oracle as O
    of.native.ensure_device()
    O.build()
    t0, cases, px, bad, msgs = time.time(), 0, 0, 0, []
    seed = args.seed * 1_000_000
    while time.time() - t0 < args.seconds:
        n, b, m = one_case(of, O, seed, args.max[0], args.max[1])
        cases += 1; px += n; bad += b
        msgs += ["seed {}: {}".format(seed, x) for x in m]
        seed += 1
    print(json.dumps({"soak": "gather family vs the CPU oracle, bit for bit", "seed_base": args.seed * 1_000_000, "cases": cases,
                      "pixels_compared": px, "mismatching_pixels_or_cases": bad, "details": msgs[:20]}))


if __name__ == "__main__":
    main()
