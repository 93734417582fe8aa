#!/usr/bin/env python
"""Soak of the slab-wise scatter: random fields of every make of tests/scatter_soak_util.py, random point masks, any
number of ranks, band edges on and off the 8-row tiles -- the bands of all ranks (played on this one GPU, tests/slab_util.py)
must equal the whole-field result bit for bit, values and validity.

    python tools/soak_slab.py [--seconds 120] [--seed 0] [--max 200 280]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max", type=int, nargs=2, default=[200, 280])
    args = ap.parse_args()
    import oflibnumpy_amd as of
    from oflibnumpy_amd import _native as nat
    from slab_util import Slab, probe_values
    from scatter_soak_util import make_case
    of.native.ensure_device()
    t0, cases, nodes, bad, msgs = time.time(), 0, 0, 0, []
    seed = args.seed * 1_000_000
    while time.time() - t0 < args.seconds:
        h, w, kind, vecs, pm, sign, C, vals, vm = make_case(seed, args.max[0], args.max[1])
        rng = np.random.default_rng(seed + 3_000_000_000)
        world, align = int(rng.integers(2, 12)), int(rng.choice([1, 8]))
        try:
            sl = Slab(vecs, pm, probe_values(h, w), entries=1 << 15, sign=sign, vmask=vm, valid_rule=int(rng.choice([0, 1, 2])))
            fo, fv, info = sl.full()
            fo, fv = fo.to_host((h, w, 2), np.float32), fv.to_host((h, w), np.uint8)
        except (RuntimeError, ValueError):
            seed += 1
            continue                                                 # (a point set the whole-field call refuses: nothing to compare)
        # the whole-field call is deterministic, and the replicated-stars bands (ofl_scatter_rows_dev) tile it as well
        fo2, fv2, _ = sl.full()
        if not (np.array_equal(fo2.to_host((h, w, 2), np.float32).view(np.uint32), fo.view(np.uint32)) and np.array_equal(fv2.to_host((h, w), np.uint8), fv)):
            bad += 1; msgs.append("seed {}: kind {} {}x{}: two whole-field calls differ".format(seed, kind, h, w))
        for r0, r1 in sl.bands(world, align)[:3]:
            if r1 > r0:
                bo, bv, _ = sl.full(r0, r1 - r0)
                if not (np.array_equal(bo.to_host((r1 - r0, w, 2), np.float32).view(np.uint32), fo[r0:r1].view(np.uint32)) and
                        np.array_equal(bv.to_host((r1 - r0, w), np.uint8), fv[r0:r1])):
                    bad += 1; msgs.append("seed {}: kind {} {}x{}: replicated band [{}, {}) differs from the whole field".format(seed, kind, h, w, r0, r1))
        out, valid, lists, bands = sl.play(world, align)
        counts, errs = sl.counts(lists, world)
        d = (out.view(np.uint32) != fo.view(np.uint32)).any(-1) | (valid != fv)
        cases += 1; nodes += h * w
        if d.any() or any(errs) or sum(counts) != info[1]:
            bad += int(d.sum()) + (1 if any(errs) or sum(counts) != info[1] else 0)
            msgs.append("seed {}: kind {} {}x{} world {} align {}: {} nodes, lists {} vs {} unfinished, errs {}".format(
                seed, kind, h, w, world, align, int(d.sum()), sum(counts), info[1], errs))
        seed += 1
    print(json.dumps({"soak": "slab bands == whole field, bit for bit", "seed_base": args.seed * 1_000_000, "cases": cases,
                      "nodes_compared": nodes, "mismatching_nodes_or_cases": bad, "details": msgs[:20]}))


if __name__ == "__main__":
    main()
