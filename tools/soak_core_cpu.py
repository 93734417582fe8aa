#!/usr/bin/env python
"""Soak of the geometry core on the CPU (no GPU needed): tests/native/dl_core_cpu.cpp compiles the very header the kernels use
(ofl_delaunay_core.h: mesh cells, mesh fans, clip pass, second per-thread pass; its far pass is a plain sequential clip) and
builds all stars of a random field of tests/scatter_soak_util.py; every simplex of scipy.spatial.Delaunay that is uniquely
Delaunay must be among the triangles the stars list.

    python tools/soak_core_cpu.py [--seconds 600] [--seed 0] [--jobs 8] [--max 140 200]
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time
from multiprocessing import Pool

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

LIB = None


def init(path):
    global LIB
    LIB = ctypes.CDLL(path)
    LIB.dl_stars_grid_cpu.restype = ctypes.c_int


def work(job):
    seed, hmax, wmax = job
    from scipy.spatial import Delaunay
    from scatter_soak_util import make_case
    import test_delaunay_core as T
    h, w, kind, vecs, pm, sign, C, vals, vm = make_case(seed, hmax, wmax)
    yy, xx = np.mgrid[:h, :w]
    P = np.stack([(xx + sign * vecs[..., 0].astype(np.float64)).ravel(), (yy + sign * vecs[..., 1].astype(np.float64)).ravel()], 1)
    kept = (pm if pm is not None else np.ones((h, w), bool)).ravel().copy()
    # duplicates: the smallest index of a location stays a site (as the library's dedupe)
    _, first = np.unique(P[kept], axis=0, return_index=True)
    idx_kept = np.flatnonzero(kept)
    keep2 = np.zeros(len(P), bool)
    keep2[idx_kept[first]] = True
    if keep2.sum() < 4:
        return seed, 0, 0, None
    try:
        tri, info = T.stars_grid(LIB, P, keep2.reshape(h, w), h, w)
    except AssertionError as e:
        return seed, 0, 1, "kind {} {}x{}: harness failed ({})".format(kind, h, w, e)
    idx = np.flatnonzero(keep2)
    try:
        d = Delaunay(P[idx])
    except Exception:
        return seed, 0, 0, None
    uniq = T.unique_simplices(P[idx], d.simplices, max(1e-9, 2.5e-11 * float(np.abs(P[idx]).max())))
    # (near-duplicate sites -- two float32 roundings of the same position, 1e-7 px apart -- make triangles of no area: which of the
    # equally empty slivers a triangulation lists there is immaterial, nothing can be interpolated in them)
    A, B, Cc = (P[idx][d.simplices[:, k]] for k in range(3))
    area = 0.5 * np.abs((B[:, 0] - A[:, 0]) * (Cc[:, 1] - A[:, 1]) - (B[:, 1] - A[:, 1]) * (Cc[:, 0] - A[:, 0]))
    edge = np.minimum(np.minimum(np.hypot(*(B - A).T), np.hypot(*(Cc - B).T)), np.hypot(*(A - Cc).T))
    uniq &= (area > 1e-6) & (edge > 1e-4)
    want = {tuple(sorted(int(idx[v]) for v in t)) for t, u in zip(d.simplices, uniq) if u}
    got = {tuple(sorted(int(v) for v in t)) for t in tri}
    miss = want - got
    msg = None
    if miss:
        msg = "kind {} {}x{} sign {} mask {}: {} of {} unique simplices missing, e.g. {}".format(kind, h, w, sign, pm is not None, len(miss), len(want), sorted(miss)[:2])
    return seed, len(want), len(miss), msg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--max", type=int, nargs=2, default=[140, 200])
    args = ap.parse_args()
    so = os.path.join(tempfile.mkdtemp(), "libdlcore.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "dl_core_cpu.cpp")])
    t0, cases, simplices, bad, msgs = time.time(), 0, 0, 0, []
    seed = args.seed * 1_000_000
    with Pool(args.jobs, initializer=init, initargs=(so,)) as pool:
        while time.time() - t0 < args.seconds:
            batch = [(seed + k, args.max[0], args.max[1]) for k in range(args.jobs * 4)]
            seed += len(batch)
            for s, n, b, msg in pool.map(work, batch):
                cases += 1; simplices += n; bad += b
                if msg:
                    msgs.append("seed {}: {}".format(s, msg))
    print(json.dumps({"soak": "geometry core on the CPU vs scipy.spatial.Delaunay", "seed_base": args.seed * 1_000_000, "cases": cases,
                      "unique_simplices_checked": simplices, "missing": bad, "details": msgs[:16]}))


if __name__ == "__main__":
    main()
