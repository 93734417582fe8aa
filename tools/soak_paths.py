#!/usr/bin/env python
"""Soak of the two scatter paths AGAINST EACH OTHER: on fields whose mesh certifies (a random similarity / shear with a
ripple: no co-circular cells to speak of), the one-kernel walk on the certified mesh and the GPU Delaunay triangulation
(forced with OFL_SCATTER_UNCERTIFIED) are independent implementations of the same function -- validity must agree exactly
and random image values within rtol 1e-4 / atol 2e-5 at every node except those within 1e-6 px (measured along an edge
function) of a triangle edge of either... here simply: at every node, bar a share below 1e-5 that is printed.  No SciPy: sizes
up to 1080p.

    python tools/soak_paths.py [--seconds 120] [--seed 0] [--max 1080 1920]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max", type=int, nargs=2, default=[1080, 1920])
    args = ap.parse_args()
    import oflibnumpy_amd as of
    from oflibnumpy_amd import device as dev
    nat = of.native
    nat.ensure_device()
    t0, cases, nodes, certified, bad_valid, bad_val, msgs = time.time(), 0, 0, 0, 0, 0, []
    seed = args.seed * 1_000_000
    while time.time() - t0 < args.seconds:
        rng = np.random.default_rng(seed)
        h, w = int(rng.integers(32, args.max[0])), int(rng.integers(32, args.max[1]))
        ts = [['rotation', float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(-30, 30))],
              ['scaling', float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(0.7, 1.4))]]
        v = np.array(of.Flow.from_transforms(ts[:int(rng.integers(1, 3))], [h, w], 's').vecs)
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        amp = float(rng.uniform(0.02, 0.25))
        win = (np.sin(np.pi * xx / (w - 1)) * np.sin(np.pi * yy / (h - 1))) ** 2          # the border stays straight: the certificate asks for that
        v[..., 0] += win * amp * np.sin(xx / rng.uniform(6, 40)) * np.cos(yy / rng.uniform(6, 40))
        v[..., 1] += win * amp * np.cos(xx / rng.uniform(6, 40)) * np.sin(yy / rng.uniform(6, 40))
        v = v.astype(np.float32)
        sign = 1 if rng.random() < 0.7 else -1
        C = int(rng.integers(1, 4))
        vals = rng.random((h, w, C), dtype=np.float32)
        vm = rng.random((h, w)) > 0.1
        f, dv, dm = dev.DeviceBuffer.from_host(v), dev.DeviceBuffer.from_host(vals), dev.DeviceBuffer.from_host(vm.astype(np.uint8))
        o1, v1, o2, v2 = dev.DeviceBuffer(h * w * C * 4), dev.DeviceBuffer(h * w), dev.DeviceBuffer(h * w * C * 4), dev.DeviceBuffer(h * w)
        i1 = dev.scatter_linear(f, sign, None, dv, C, dm, h, w, None, o1, v1, 0)
        i2 = dev.scatter_linear(f, sign, None, dv, C, dm, h, w, None, o2, v2, nat.SCATTER_UNCERTIFIED)
        cases += 1
        if not (i1[1] == 0 and i1[2] == 0 and i2[1] > 0):
            seed += 1
            continue                                                 # (the first call did not take the certified path: nothing to compare)
        certified += 1
        a, av = o1.to_host((h, w, C), np.float32), v1.to_host((h, w), np.uint8)
        b, bv = o2.to_host((h, w, C), np.float32), v2.to_host((h, w), np.uint8)
        dvl = av != bv
        dva = ~np.isclose(a, b, rtol=1e-4, atol=2e-5).all(-1) & ~dvl
        nodes += h * w
        if dvl.any() or dva.any():
            bad_valid += int(dvl.sum()); bad_val += int(dva.sum())
            y, x = np.argwhere(dvl | dva)[0]
            msgs.append("seed {}: {}x{} sign {}: {} validity, {} value nodes; first ({}, {}) walk {} {} delaunay {} {}".format(
                seed, h, w, sign, int(dvl.sum()), int(dva.sum()), y, x, a[y, x].tolist(), int(av[y, x]), b[y, x].tolist(), int(bv[y, x])))
        seed += 1
    print(json.dumps({"soak": "certified walk vs GPU Delaunay on certified fields", "seed_base": args.seed * 1_000_000, "cases": cases,
                      "certified_cases": certified, "nodes_compared": nodes, "validity_mismatches": bad_valid, "value_mismatches": bad_val,
                      "details": msgs[:12]}))


if __name__ == "__main__":
    main()
