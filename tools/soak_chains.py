#!/usr/bin/env python
"""Soak of the composed paths -- combine_with modes 1 and 2, invert, switch_ref, both references -- STAGE BY STAGE against the
oracle (tests/test_gpu_chains.py's machinery): the oracle runs a chain on random operands (a random similarity / shear each,
optionally with a sinusoidal perturbation and random masks), then the PRODUCT runs every stage on the oracle's operands of that
stage: exact stages and gathers must agree bit for bit, scatters on every node outside SciPy's non-unique simplices and the
hull band (validity exactly, values at rtol 1e-4 / atol 2e-5).

    python tools/soak_chains.py [--seconds 120] [--seed 0] [--max 120 160]
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


def random_transforms(rng, h, w):
    t = []
    for _ in range(int(rng.integers(1, 3))):
        k = int(rng.integers(0, 3))
        if k == 0:
            t.append(['rotation', float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(-25, 25))])
        elif k == 1:
            t.append(['scaling', float(rng.uniform(0, w)), float(rng.uniform(0, h)), float(rng.uniform(0.8, 1.2))])
        else:
            t.append(['translation', float(rng.uniform(-8, 8)), float(rng.uniform(-8, 8))])
    return t


def one_case(of, O, C, seed, hmax, wmax):
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(16, hmax)), int(rng.integers(16, wmax))
    shape = (h, w)
    kind, ref = [('mode1', 's'), ('mode1', 't'), ('mode2', 's'), ('mode2', 't'), ('invert', 's'), ('invert', 't'), ('switch', 's'), ('switch', 't')][seed % 8]
    t1, t2 = random_transforms(rng, h, w), random_transforms(rng, h, w)
    f1, f2, f3 = (of.Flow.from_transforms(t, list(shape), ref) for t in (t1, t2, t1 + t2))
    a, b = (f2, f3) if kind == 'mode1' else ((f1, f3) if kind == 'mode2' else (f3, f3))
    yy, xx = np.mgrid[:h, :w].astype(np.float32)

    def dress(f, salt):
        v = np.array(f.vecs)
        if rng.random() < 0.5:                                              # a non-affine ripple: co-circular cells become rare
            amp = float(rng.uniform(0.05, 0.8))
            v[..., 0] += amp * np.sin(xx / rng.uniform(5, 30) + salt) * np.cos(yy / rng.uniform(5, 30))
            v[..., 1] += amp * np.cos(xx / rng.uniform(5, 30)) * np.sin(yy / rng.uniform(5, 30) + salt)
        m = np.ones(shape, bool) if rng.random() < 0.5 else rng.random(shape) > rng.choice([0.03, 0.2])
        return v.astype(np.float32), f.ref, m

    a, b = dress(a, 0.3), dress(b, 1.1)
    if kind in ('invert', 'switch'):
        b = a
    chain = C.CHAINS[(kind, ref)]
    Ob, Pb = C.OracleBackend(O), C.ProductBackend(of)
    try:
        want = C.run_chain(Ob, chain, a, b)
    except Exception as e:                                                   # (a chain SciPy refuses: no points kept, flat simplex)
        return 0, 0, []
    n, bad, msgs = 0, 0, []
    for name, op, *args in chain:
        xs = [want[k] for k in args]
        try:
            got = C.triple(C.run_stage(Pb, op, [Pb.flow(*x) for x in xs]))
        except Exception as e:
            bad += 1
            msgs.append("{} {} stage {} ({}) {}x{}: product raised {}".format(kind, ref, name, op, h, w, str(e)[:80]))
            continue
        k, sign = C.stage_kind(op, xs)
        if got[1] != want[name][1]:
            bad += 1; msgs.append("{} {} stage {}: reference label".format(kind, ref, name)); continue
        if k == 'scatter':
            try:
                amb, band = C.scatter_ambiguity(op, sign, xs, shape)
            except Exception:
                continue                                                     # (hull of a degenerate point set)
            d = C.close(got, want[name]) & ~amb & ~band
            n += int((~amb & ~band).sum())
        else:
            d = (got[0].view(np.uint32) != want[name][0].view(np.uint32)).any(-1) | (got[2] != want[name][2])
            n += h * w
        if d.any():
            bad += int(d.sum())
            y, x = np.argwhere(d)[0]
            msgs.append("{} {} stage {} ({}, {}) {}x{}: {} nodes, first ({}, {}) got {} {} want {} {}".format(
                kind, ref, name, op, k, h, w, int(d.sum()), y, x, got[0][y, x].tolist(), bool(got[2][y, x]), want[name][0][y, x].tolist(), bool(want[name][2][y, x])))
    return n, bad, msgs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max", type=int, nargs=2, default=[120, 160])
    args = ap.parse_args()
    import oflibnumpy_amd as of
    from oracle import np_oracle as O
    import test_gpu_chains as C
    of.native.ensure_device()
    O.build()
    t0, cases, nodes, bad, msgs = time.time(), 0, 0, 0, []
    seed = args.seed * 1_000_000
    while time.time() - t0 < args.seconds:
        n, b, m = one_case(of, O, C, seed, args.max[0], args.max[1])
        cases += 1; nodes += n; bad += b
        msgs += ["seed {}: {}".format(seed, x) for x in m]
        seed += 1
    print(json.dumps({"soak": "composed paths, stage by stage on the oracle's operands", "seed_base": args.seed * 1_000_000, "cases": cases,
                      "nodes_compared": nodes, "mismatching_nodes_or_cases": bad, "details": msgs[:20]}))


if __name__ == "__main__":
    main()
