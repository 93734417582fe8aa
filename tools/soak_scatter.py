#!/usr/bin/env python
"""Soak of the scattered -> grid path against SciPy (the oracle's scipy.interpolate.griddata, utils.py:253): random fields,
sizes, point masks, folds, holes, signs, value masks -- validity bit-exact and values within 1e-4 outside SciPy's non-unique
simplices, for as long as asked.

    python tools/soak_scatter.py [--seconds 120] [--seed 0] [--max 160 240]

Prints one JSON line: cases run, nodes compared, mismatches (a mismatch also dumps the case's seed for replay).
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


from scatter_soak_util import one_case, one_query_case, one_track_case


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max", type=int, nargs=2, default=[160, 240])
    ap.add_argument("--cluster", action="store_true", help="dense clusters of distinct sites: a block of every field contracted 5 .. 200 times (seeds base + 900 000 ...)")
    ap.add_argument("--mode", default="grid", choices=["grid", "query", "track"], help="grid nodes (apply 's', invert, ...) or scattered query positions (mode 2 't')")
    args = ap.parse_args()
    import oflibnumpy_amd as of
    from oflibnumpy_amd import device as dev
    from oracle import np_oracle as O
    from scatter_util import nonunique_nodes, hull_band
    of.native.ensure_device()
    O.build()
    t0, cases, nodes, bad, msgs = time.time(), 0, 0, 0, []
    seed = args.seed * 1_000_000 + (900_000 if args.cluster else 0)
    while time.time() - t0 < args.seconds:
        if args.mode == "track":
            n, b, msg = one_track_case(of, O, seed, args.max[0], args.max[1])
        else:
            n, b, msg = (one_case if args.mode == "grid" else one_query_case)(dev, O, nonunique_nodes, hull_band, seed, args.max[0], args.max[1])
        cases += 1; nodes += n; bad += b
        if msg:
            msgs.append("seed {}: {}".format(seed, msg))
        seed += 1
    print(json.dumps({"soak": "scatter path vs SciPy" + (" (dense clusters)" if args.cluster else "") + (" (query positions)" if args.mode == "query" else " (track_pts)" if args.mode == "track" else ""), "seed_base": args.seed * 1_000_000, "cases": cases, "nodes_compared": nodes,
                      "mismatching_nodes_or_cases": bad, "details": msgs[:20]}))


if __name__ == "__main__":
    main()
