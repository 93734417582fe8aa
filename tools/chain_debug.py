"""debug aid for one failing soak_chains seed: the failing stage alone, through the grid path, the query path and row bands"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import soak_chains as S
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev
from oracle import np_oracle as O
import test_gpu_chains as C
of.native.ensure_device(); O.build()
nat = of.native
seed, stage, ny, nx = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(seed)
h, w = int(rng.integers(16, 120)), int(rng.integers(16, 160)); shape = (h, w)
kind, ref = [('mode1', 's'), ('mode1', 't'), ('mode2', 's'), ('mode2', 't'), ('invert', 's'), ('invert', 't'), ('switch', 's'), ('switch', 't')][seed % 8]
t1, t2 = S.random_transforms(rng, h, w), S.random_transforms(rng, h, w)
f1, f2, f3 = (of.Flow.from_transforms(t, list(shape), ref) for t in (t1, t2, t1 + t2))
a, b = (f2, f3) if kind == 'mode1' else ((f1, f3) if kind == 'mode2' else (f3, f3))
yy, xx = np.mgrid[:h, :w].astype(np.float32)
def dress(f, salt):
    v = np.array(f.vecs)
    if rng.random() < 0.5:
        amp = float(rng.uniform(0.05, 0.8))
        v[..., 0] += amp * np.sin(xx / rng.uniform(5, 30) + salt) * np.cos(yy / rng.uniform(5, 30))
        v[..., 1] += amp * np.cos(xx / rng.uniform(5, 30)) * np.sin(yy / rng.uniform(5, 30) + salt)
    m = np.ones(shape, bool) if rng.random() < 0.5 else rng.random(shape) > rng.choice([0.03, 0.2])
    return v.astype(np.float32), f.ref, m
a, b = dress(a, 0.3), dress(b, 1.1)
if kind in ('invert', 'switch'):
    b = a
chain = C.CHAINS[(kind, ref)]
want = C.run_chain(C.OracleBackend(O), chain, a, b)
for name, op, *args in chain:
    if name != stage: continue
    xs = [want[k] for k in args]
    vecs, r, mask = xs[0]
    k, sign = C.stage_kind(op, xs)
    print("stage", name, op, k, sign, "mask share", mask.mean())
    Pb = C.ProductBackend(of)
    got = C.triple(C.run_stage(Pb, op, [Pb.flow(*x) for x in xs]))
    print("grid path  :", got[0][ny, nx].tolist(), bool(got[2][ny, nx]), " want", want[name][0][ny, nx].tolist())
    # the scatter by hand: values = -vecs for switch_ref ('s' -> 't'): positions x + vecs
    fb = dev.DeviceBuffer.from_host(np.ascontiguousarray(vecs)); mb = dev.DeviceBuffer.from_host(mask.astype(np.uint8))
    out = dev.DeviceBuffer(h * w * 8); valid = dev.DeviceBuffer(h * w)
    info = dev.scatter_linear(fb, sign, mb, fb, 2, mb, h, w, None, out, valid, 0)
    res = out.to_host((h, w, 2), np.float32); print("by hand    :", res[ny, nx].tolist(), "info", info)
    # the query path at the same position
    q = np.array([[nx, ny]], np.float32); qb = dev.DeviceBuffer.from_host(q)
    oq = dev.DeviceBuffer(8); vq = dev.DeviceBuffer(1)
    dev.scatter_linear(fb, sign, mb, fb, 2, mb, h, w, qb, oq, vq, 0)
    print("query path :", oq.to_host((2,), np.float32).tolist(), vq.to_host((1,), np.uint8).tolist())
    # row bands
    for r0, rows in ((8, 16), (0, 24), (8, 8)):
        ob = dev.DeviceBuffer(rows * w * 8); vb = dev.DeviceBuffer(rows * w)
        dev.scatter_rows(fb, sign, mb, fb, 2, mb, h, w, r0, rows, ob, vb, 0)
        rb = ob.to_host((rows, w, 2), np.float32); print("band", r0, rows, ":", rb[ny - r0, nx].tolist())
    print("want-like by hand (SciPy on the same inputs):")
    from scipy.interpolate import griddata
    pts = np.stack([(xx.astype(np.float64) + sign * vecs[..., 0].astype(np.float64))[mask], (yy.astype(np.float64) + sign * vecs[..., 1].astype(np.float64))[mask]], 1)
    gv = griddata(pts, vecs[mask].astype(np.float64), np.array([[nx, ny]], np.float64), method='linear')
    print("            ", gv.tolist())
