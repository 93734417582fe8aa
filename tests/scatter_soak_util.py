"""One random case of the scattered -> grid path against SciPy (tools/soak_scatter.py runs them by the thousand; the seeds that
once failed are replayed by tests/test_gpu_scatter_exact.py::test_soak_regressions)."""
import numpy as np


def make_case(seed, hmax, wmax):
    """-> h, w, kind, vecs float32 [h][w][2], point mask or None, sign, C, vals, value mask (all from the seed alone)"""
    rng = np.random.default_rng(seed)
    h, w = int(rng.integers(4, hmax)), int(rng.integers(4, wmax))
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    kind = int(rng.integers(0, 6))
    if seed % 1_000_000 >= 900_000:                                 # (seed ranges of their own -- tools/soak_scatter.py --cluster: the cases of all
        kind = 6                                                    # other seeds, the regression seeds of tests/ among them, stay what they were)
    amp = float(rng.uniform(0.1, 6))
    vecs = np.stack([amp * np.sin(xx / rng.uniform(3, 40) + yy / rng.uniform(5, 50)) + rng.uniform(-0.2, 0.2) * yy,
                     amp * np.cos(xx / rng.uniform(4, 35) - yy / rng.uniform(6, 45)) + rng.uniform(-0.2, 0.2) * xx], -1).astype(np.float32)
    vecs += rng.uniform(-6, 6, 2).astype(np.float32)
    if kind == 1:                                                   # isolated outliers: folds
        vecs[rng.random((h, w)) < 0.03] += np.float32(rng.uniform(3, 15))
    elif kind == 2:                                                 # a block moving on its own: tear + fold
        y0, x0 = int(rng.integers(0, max(h // 2, 1))), int(rng.integers(0, max(w // 2, 1)))
        vecs[y0:y0 + max(h // 3, 1), x0:x0 + max(w // 3, 1)] += rng.uniform(-10, 10, 2).astype(np.float32)
    elif kind == 3:                                                 # integer vectors: lattice sites, duplicates, co-circular cells
        vecs = np.round(vecs)
    elif kind == 4:                                                 # a similarity: every cell close to co-circular
        a, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.7, 1.3)
        cx, cy = rng.uniform(0, w), rng.uniform(0, h)
        vecs = np.stack([(sc * np.cos(a) - 1) * (xx - cx) - sc * np.sin(a) * (yy - cy),
                         sc * np.sin(a) * (xx - cx) + (sc * np.cos(a) - 1) * (yy - cy)], -1).astype(np.float32)
    elif kind == 5:                                                 # lattices of other makes (the regression seeds are kinds 3 and 4)
        sub = int(rng.integers(0, 3))
        if sub == 0:                                                # BASELINE config 5's make: rows that slide against each other
            px, py = int(rng.integers(3, 24)), int(rng.integers(2, 12))
            vecs = np.stack([((xx % px) * (yy % py)).astype(np.float32) * np.float32(rng.choice([1.0, 0.5, 0.25])), np.zeros((h, w), np.float32)], -1)
        elif sub == 1:                                              # a translation: every cell an exact square
            vecs = np.broadcast_to(rng.uniform(-7, 7, 2).astype(np.float32), (h, w, 2)).copy()
            if rng.random() < 0.5:
                y0, x0 = int(rng.integers(0, max(h // 2, 1))), int(rng.integers(0, max(w // 2, 1)))
                vecs[y0:y0 + max(h // 3, 1), x0:x0 + max(w // 3, 1)] += np.round(rng.uniform(-9, 9, 2)).astype(np.float32)
        else:                                                       # piecewise constant integer vectors: sheets that fold and tear
            by, bx = int(rng.integers(3, 20)), int(rng.integers(3, 20))
            gy, gx = (yy // by).astype(int), (xx // bx).astype(int)
            tab = np.round(rng.uniform(-6, 6, (gy.max() + 1, gx.max() + 1, 2))).astype(np.float32)
            vecs = tab[gy, gx]
        vecs = np.ascontiguousarray(vecs, np.float32)
    elif kind == 6:                                                 # a dense cluster of DISTINCT sites: a block contracted 5 .. 200 times
        y0, x0 = int(rng.integers(0, max(h // 2, 1))), int(rng.integers(0, max(w // 2, 1)))
        bh, bw = max(h // 2, 2), max(w // 2, 2)
        f = float(np.exp(rng.uniform(np.log(5), np.log(200))))
        cy, cx = rng.uniform(0, h), rng.uniform(0, w)
        sl = (slice(y0, y0 + bh), slice(x0, x0 + bw))
        ys, xs = yy[sl], xx[sl]
        rip = 1.0 + 0.05 * np.sin(xs / 5.3 + ys / 7.1)             # (no cell of the contracted lattice exactly co-circular)
        vecs[sl + (0,)] = (cx + (xs - xs.mean()) / f * rip - xs).astype(np.float32)
        vecs[sl + (1,)] = (cy + (ys - ys.mean()) / f * rip - ys).astype(np.float32)
    pm = None
    if rng.random() < 0.6:
        pm = rng.random((h, w)) > rng.uniform(0, 0.5)
        if rng.random() < 0.4 and h > 12 and w > 12:
            pm[h // 4:h // 4 + h // 3, w // 5:w // 5 + w // 2] = False   # a hole
        if pm.sum() < 8:
            pm = None
    sign = 1 if rng.random() < 0.7 else -1
    C = int(rng.integers(1, 4))
    vals = rng.random((h, w, C), dtype=np.float32)
    vm = rng.random((h, w)) > 0.15
    return h, w, kind, vecs, pm, sign, C, vals, vm


def one_case(dev, O, nonunique_nodes, hull_band, seed, hmax, wmax):
    h, w, kind, vecs, pm, sign, C, vals, vm = make_case(seed, hmax, wmax)
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    f, dv, dm = dev.DeviceBuffer.from_host(vecs), dev.DeviceBuffer.from_host(vals), dev.DeviceBuffer.from_host(vm.astype(np.uint8))
    dpm = dev.DeviceBuffer.from_host(pm.astype(np.uint8)) if pm is not None else None
    out, valid = dev.DeviceBuffer(h * w * C * 4), dev.DeviceBuffer(h * w)
    try:
        dev.scatter_linear(f, sign, dpm, dv, C, dm, h, w, None, out, valid, 0)
    except (RuntimeError, ValueError) as e:
        # a refused point set: SciPy must refuse it as well (QhullError) -- or it is a mismatch
        try:
            O.scatter_griddata(sign * vecs, np.concatenate([vals, vm[..., None].astype(np.float32)], -1), pm)
        except Exception:
            return 0, 0, None
        return 0, 1, "refused by the library, accepted by SciPy: " + str(e)[:120]
    got, gv = out.to_host((h, w, C), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
    try:
        want = O.scatter_griddata(sign * vecs, np.concatenate([vals, vm[..., None].astype(np.float32)], -1), pm)
    except Exception as e:
        return 0, 1, "accepted by the library, refused by SciPy: " + str(e)[:120]
    keep = np.ones((h, w), bool) if pm is None else pm
    pts = np.stack([(xx + sign * vecs[..., 0].astype(np.float64)).ravel(), (yy + sign * vecs[..., 1].astype(np.float64)).ravel()], 1)[keep.ravel()]
    # (Qhull's own roundoff grows with the coordinates: what counts as "the fourth site lies on the circle" scales with them)
    amb, inside = nonunique_nodes(pts, (h, w), tol=max(1e-9, 2.5e-11 * float(np.abs(pts).max())))
    try:
        band = hull_band(pts, (h, w))
    except Exception:
        band = np.zeros((h, w), bool)
    sel = ~amb & ~band
    bad_v = (gv != (want[..., -1] == 1)) & sel
    bad = ~np.isclose(got, want[..., :C], rtol=1e-4, atol=2e-5).all(-1) & sel
    n_bad = int(bad_v.sum() + bad.sum())
    msg = None
    if n_bad:
        msg = "kind {} {}x{} sign {} mask {}: {} validity, {} value nodes, first {}".format(
            kind, h, w, sign, pm is not None, int(bad_v.sum()), int(bad.sum()), np.argwhere(bad_v | bad)[:3].tolist())
    return int(sel.sum()), n_bad, msg


def one_query_case(dev, O, nonunique_nodes, hull_band, seed, hmax, wmax):
    """combine_with mode 2, ref 't' (flow_class.py:1398-1410): f1 resampled from the float32 points x - f1 onto the positions
    x - f3 -- scattered QUERY positions, every point kept, the mask a value channel, validity = interpolated mask > 0.99.
    The product's DeviceFlow._resample_to against the oracle's _mode2_t_resample (SciPy griddata at the same positions)."""
    h, w, kind, f1, pm, sign, C, vals, vm = make_case(seed, hmax, wmax)
    rng = np.random.default_rng(seed + 7_000_000_000)
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    amp = float(rng.uniform(0.1, 5))
    f3 = np.stack([amp * np.cos(xx / rng.uniform(4, 40) + yy / rng.uniform(6, 50)) + rng.uniform(-3, 3),
                   amp * np.sin(yy / rng.uniform(4, 40) - xx / rng.uniform(6, 50)) + rng.uniform(-3, 3)], -1).astype(np.float32)
    m1 = vm if pm is None else pm
    try:
        got = dev.DeviceFlow.from_host(f1, 't', m1)._resample_to(dev.DeviceFlow.from_host(f3, 't'))
        gv, gm = got.to_host()
    except (RuntimeError, ValueError) as e:
        try:
            O._mode2_t_resample(O.OFlow(f1, 't', m1), O.OFlow(f3, 't'))
        except Exception:
            return 0, 0, None
        return 0, 1, "refused by the library, accepted by SciPy: " + str(e)[:120]
    try:
        want = O._mode2_t_resample(O.OFlow(f1, 't', m1), O.OFlow(f3, 't'))
    except Exception as e:
        return 0, 1, "accepted by the library, refused by SciPy: " + str(e)[:120]
    # the points and the positions as the reference builds them: float32 arrays (flow_class.py:1398-1406)
    c1 = np.copy(-f1); c1[:, :, 0] += np.arange(w); c1[:, :, 1] += np.arange(h)[:, None]
    c3 = np.copy(-f3); c3[:, :, 0] += np.arange(w); c3[:, :, 1] += np.arange(h)[:, None]
    pts, q = c1.reshape(-1, 2).astype(np.float64), c3.reshape(-1, 2).astype(np.float64)
    amb, inside = nonunique_nodes(pts, (h, w), queries=q, tol=max(1e-9, 2.5e-11 * float(np.abs(pts).max())))
    try:
        band = hull_band(pts, (h, w), queries=q)
    except Exception:
        band = np.zeros((h, w), bool)
    sel = ~amb & ~band
    bad = ~np.isclose(gv, want.vecs, rtol=1e-4, atol=2e-5).all(-1) & sel
    bad_m = (gm != want.mask) & sel
    n_bad = int(bad.sum() + bad_m.sum())
    msg = None
    if n_bad:
        msg = "query kind {} {}x{}: {} mask, {} value positions, first {}".format(kind, h, w, int(bad_m.sum()), int(bad.sum()),
                                                                                   np.argwhere(bad | bad_m)[:3].tolist())
    return int(sel.sum()), n_bad, msg


def one_track_case(of, O, seed, hmax, wmax):
    """track_pts (utils.py:547-622) through the packaged mirror against the oracle: random points (inside, on and outside the
    image; float64, float32 and integer), both references, s_exact_mode, int_out.  's' bilinear samples must agree to float
    rounding (1e-5 px), the Delaunay forms ('t', s_exact_mode) to 1e-4 relative wherever the covering simplex is unique."""
    from scipy.spatial import Delaunay
    from test_delaunay_core import unique_simplices
    h, w, kind, vecs, pm, sign, C, vals, vm = make_case(seed, hmax, wmax)
    rng = np.random.default_rng(seed + 11_000_000_000)
    n = int(rng.integers(1, 400))
    pts = np.stack([rng.uniform(-3, h + 2, n), rng.uniform(-3, w + 2, n)], 1)
    ref = 's' if rng.random() < 0.5 else 't'
    exact = bool(rng.random() < 0.5)
    int_out = bool(rng.random() < 0.3)
    dt = rng.choice(['f8', 'f4', 'i'])
    if dt == 'i':
        pts = np.stack([rng.integers(0, h, n), rng.integers(0, w, n)], 1).astype(np.int64)
    else:
        if ref == 's' and not exact:                           # bilinear_interpolation indexes the image: positions must lie inside it
            pts = np.stack([rng.uniform(0, h - 1, n), rng.uniform(0, w - 1, n)], 1)
        pts = pts.astype(dt)
    try:
        want = O.track_pts(vecs, ref, pts.copy(), int_out, exact)
    except Exception:
        return 0, 0, None                                      # (the reference's own arithmetic refuses the case: nothing to compare)
    try:
        got = of.track_pts(vecs, ref, pts.copy(), int_out, exact)
    except Exception as e:
        return 0, 1, "track {} {} {}x{} exact {}: product raised {}".format(ref, dt, h, w, exact, str(e)[:100])
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    sel = np.ones(n, bool)
    if (ref == 't' or (exact and dt != 'i')) and not O.is_zero_flow(vecs, True):
        # the Delaunay forms: exempt positions whose covering simplex is not unique, and those within 1e-5 px of the hull border
        yy, xx = np.mgrid[:h, :w]
        src = np.stack([yy.ravel(), xx.ravel()], 1).astype(np.float64)
        P = src - vecs[..., ::-1].reshape(-1, 2).astype(np.float64) if ref == 't' else src
        up, counts = np.unique(P, axis=0, return_counts=True)
        try:
            d = Delaunay(up)
        except Exception:
            return 0, 0, None
        q = pts.astype(np.float64)
        s_ = d.find_simplex(q)
        uniq = unique_simplices(up, d.simplices, max(1e-9, 2.5e-11 * float(np.abs(up).max())))
        sel = (s_ >= 0)
        sel[s_ >= 0] = uniq[s_[s_ >= 0]] & ~(counts[d.simplices[s_[s_ >= 0]]] > 1).any(1)
        from scipy.spatial import ConvexHull
        hull = ConvexHull(up)
        dist = (q @ hull.equations[:, :2].T + hull.equations[:, 2]).max(1)
        sel &= np.abs(dist) > 1e-5
        sel |= (s_ < 0) & (dist > 1e-5)                        # clearly outside: both must say so (0)
    tol = 0.51 if int_out else None
    if int_out:
        bad = (np.abs(got - want) > 1).any(1) & sel            # (a rounding boundary may fall either way)
    else:
        bad = ~np.isclose(got, want, rtol=1e-4, atol=1e-4 if (ref == 's' and not exact) else 2e-4).all(1) & sel
    msg = None
    if bad.any():
        i = int(np.flatnonzero(bad)[0])
        msg = "track ref {} pts {} exact {} int_out {} {}x{} kind {}: {} of {} points, first {} got {} want {}".format(
            ref, dt, exact, int_out, h, w, kind, int(bad.sum()), n, pts[i].tolist(), got[i].tolist(), want[i].tolist())
    return int(sel.sum()), int(bad.sum()), msg
