"""GPU parity tests of the two round-2 scatter paths behind ofl_scatter_linear_dev:

  * the CERTIFIED fast path (ofl_scatter_certify_dev + ofl_scatter_certified_dev): when the cell-wise mesh of the
    warped grid provably is the Delaunay triangulation SciPy builds, one kernel resolves the grid;
  * the EXACT path for everything else: a real Delaunay triangulation of the kept points on the GPU.

Expected values come from the oracle (which calls the same scipy.interpolate.griddata as the reference,
src/oflibnumpy/utils.py:253) and from outputs of the real reference in tests/golden/.  Bars: validity masks
bit-exact, values within 1e-4 relative wherever the Delaunay triangulation is unique (co-circular cells -- exact
squares of translations / axis-aligned scalings -- are the only exemption: Qhull itself is arbitrary there).
"""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 2e-5


def certify(of, vecs, sign=1, pmask=None, pp=0, with_bits=True):
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    h, w = vecs.shape[:2]
    f = dev.DeviceBuffer.from_host(np.ascontiguousarray(vecs, np.float32))
    pm = dev.DeviceBuffer.from_host(pmask.astype(np.uint8)) if pmask is not None else None
    ws = dev._workspace(h, w, 0)
    c = nat.MeshCert()
    nb = ctypes.c_size_t(0)
    nat.check(lib.ofl_scatter_diag_bytes(h, w, ctypes.byref(nb)))
    bits = dev.DeviceBuffer(nb.value) if with_bits else None
    nat.check(lib.ofl_scatter_certify_dev(f.ptr, sign, pp, pm.ptr if pm is not None else None, h, w, ws.ptr, ws.nbytes,
                                          ctypes.byref(c), bits.ptr if bits is not None else None, None))
    c._diag_buf = bits          # the plane must outlive the certificate
    return c


def wobble(shape, ax=0.5, ay=0.4):
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float32)
    return np.stack([ax * np.sin(xx / 9) * np.cos(yy / 7), ay * np.cos(xx / 8) * np.sin(yy / 6)], -1).astype(np.float32)


@pytest.fixture
def near2_always():
    """Marks the tests that cross the star passes.  The second per-thread pass only runs for fields with tens of thousands of
    unfinished points; test_near2_route_on_the_experiments_build re-runs exactly these tests in a child process on the
    experiments build (libofl_hip_exp.so) with OFL_DL_NEAR2_MIN=0, which sends the small fixtures through that pass as
    well -- the shipped library has no such knob."""
    return "near2" if os.environ.get("OFL_DL_NEAR2_MIN") == "0" else "default"


def test_near2_route_on_the_experiments_build(gpu):
    """Both routes through the star passes give the same results: the near2-marked tests of this file, once more, in a
    child process that loads the experiments build with the per-thread pass forced on (see near2_always)."""
    import subprocess
    import sys
    if os.environ.get("OFL_DL_NEAR2_MIN") == "0":
        pytest.skip("this IS the child run")
    from oflibnumpy_amd import build_native
    if not os.path.exists(build_native.EXP_OUT):
        build_native.build_experiments()
    env = dict(os.environ, OFL_LIB=build_native.EXP_OUT, OFL_DL_NEAR2_MIN="0")
    here = os.path.abspath(__file__)
    p = subprocess.run([sys.executable, "-m", "pytest", here, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "reference_outputs or bands_and_determinism or notch_open or medium_field or random_fields"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=os.path.dirname(os.path.dirname(here)))
    tail = p.stdout[-1500:] + p.stderr[-1500:]
    assert p.returncode == 0 and " passed" in p.stdout and "failed" not in p.stdout, tail


def test_certificates(gpu):
    """What the certify pass reports for the field families of the reference's tests and of BASELINE.json."""
    of = gpu
    shape = (48, 64)
    for tr in ([['rotation', 20, 30, 25]], [['scaling', 10, 8, 0.85]], [['translation', 3, -2]],
               [['translation', 3.3, -2.7]], [['rotation', 16, 12, 15], ['scaling', 5, 5, 1.1]]):
        for sign in (1, -1):
            c = certify(of, of.from_transforms(tr, shape, 's'), sign)
            assert c.certified == 1 and c.folded_cells == 0 and c.bad_edges == 0 and c.dropped == 0, (tr, sign)
            assert c.border_dev < 1e-4
    v = of.from_transforms([['rotation', 20, 30, 10]], shape, 's')
    c = certify(of, v + wobble(shape, 1.5, 1.2))
    assert c.certified == 0 and c.border_dev > 0.1 and c.folded_cells == 0           # curved border: hull != mesh border
    blk = np.zeros(shape + (2,), np.float32)
    blk[12:28, 16:40] = [4.3, 2.6]
    c = certify(of, blk)
    assert c.certified == 0 and c.folded_cells > 0 and c.bad_edges > 0               # motion boundary: folds and stretched cells
    shear = np.zeros(shape + (2,), np.float32)
    shear[..., 0] = 1.7 * np.mgrid[:shape[0], :shape[1]][0]                          # x += 1.7 y: grid edges stop being Delaunay
    c = certify(of, shear)
    assert c.certified == 0 and c.folded_cells == 0 and c.bad_edges > 1000
    m = np.ones(shape, bool)
    m[5, 7] = False
    assert certify(of, v, pmask=m).certified == 0 and certify(of, v, pmask=m).dropped == 1
    # BASELINE configs 1-4 at reduced size: all affine
    for tr in ([['rotation', 200, 150, -30]], [['scaling', 100, 80, 0.9]], [['rotation', 192, 108, -20], ['scaling', 100, 80, 0.9]]):
        assert certify(of, of.from_transforms(tr, (216, 384), 's')).certified == 1


def affine_field(shape, a, b, c, d, tx=0.0, ty=0.0):
    """flow of the affine map (x, y) -> (x, y) + [[a, b], [c, d]] (x - w/2, y - h/2) + (tx, ty)"""
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float64)
    xx -= shape[1] / 2.0
    yy -= shape[0] / 2.0
    return np.stack([a * xx + b * yy + tx, c * xx + d * yy + ty], -1).astype(np.float32)


def test_certified_walk_matches_scipy(gpu, oracle):
    """Image-valued 's' warps of certified fields against SciPy: random (non-affine) values, a speckled value mask,
    1-3 channels, both signs.  Generic affine maps (anisotropic scaling + mild shear) have no co-circular cell, so
    EVERY node must agree; similarity transforms leave exactly co-circular cells (their float32 rounding is the
    same at all four corners of ~13 % of the cells), where Qhull's choice is arbitrary and only the rest is compared."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    from test_gpu_scatter import ambiguous_nodes
    rng = np.random.default_rng(5)
    fields = [((40, 56), affine_field((40, 56), 0.15, 0.2, -0.1, 0.25, 1.3, -0.4), True),
              ((37, 91), affine_field((37, 91), -0.2, 0.1, 0.15, -0.1, -3.0, 2.0), True),
              ((64, 48), affine_field((64, 48), 0.4, -0.15, 0.1, 0.3), True),
              ((40, 56), of.from_transforms([['rotation', 20, 25, 17]], (40, 56), 's'), False),
              ((50, 50), of.from_transforms([['scaling', 20, 30, 0.8], ['rotation', 25, 25, 45]], (50, 50), 's'), False)]
    for it, (shape, vecs, unique) in enumerate(fields):
        tr = it
        h, w = shape
        sign = 1 if it % 2 == 0 else -1
        C = 1 + it % 3
        vals = rng.standard_normal((h, w, C)).astype(np.float32)
        vm = rng.random((h, w)) > 0.2
        c = certify(of, vecs, sign)
        assert c.certified == 1
        f, dv, dm = (dev.DeviceBuffer.from_host(a) for a in (vecs, vals, vm.astype(np.uint8)))
        out, valid, cnt = dev.DeviceBuffer(h * w * C * 4), dev.DeviceBuffer(h * w), dev.DeviceBuffer.zeros(16)
        of.native.check(of.native.load().ofl_scatter_certified_dev(f.ptr, sign, 0, dv.ptr, C, dm.ptr, h, w, 0, h, out.ptr, valid.ptr,
                                                                   0, ctypes.byref(c), cnt.ptr, None))
        got, gv = out.to_host((h, w, C), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
        assert int(cnt.to_host((1,), np.uint32)[0]) == 0
        want = O.scatter_griddata(sign * vecs, np.concatenate([vals, vm[..., None].astype(np.float32)], -1), None)
        amb = ambiguous_nodes(vecs, sign)
        assert amb.sum() == 0 if unique else amb.mean() < 0.7
        np.testing.assert_array_equal(gv[~amb], (want[..., -1] == 1)[~amb], err_msg=str(tr))
        bad = ~np.isclose(got, want[..., :C], rtol=RTOL, atol=ATOL).all(-1)
        assert not (bad & ~amb).any(), (tr, int((bad & ~amb).sum()))
        # the generic entry takes the same path and gives the same bits; so do row bands
        out2, valid2 = dev.DeviceBuffer(h * w * C * 4), dev.DeviceBuffer(h * w)
        dev.scatter_linear(f, sign, None, dv, C, dm, h, w, None, out2, valid2, 0)
        np.testing.assert_array_equal(out2.to_host((h, w, C), np.float32), got)
        np.testing.assert_array_equal(valid2.to_host((h, w), np.uint8).astype(bool), gv)
        parts = []
        for r0, r1 in ((0, 13), (13, 14), (14, h)):
            ob, vb = dev.DeviceBuffer((r1 - r0) * w * C * 4), dev.DeviceBuffer((r1 - r0) * w)
            dev.scatter_rows(f, sign, None, dv, C, dm, h, w, r0, r1 - r0, ob, vb)
            parts.append(ob.to_host((r1 - r0, w, C), np.float32))
        np.testing.assert_array_equal(np.concatenate(parts), got)


def test_certified_4k_no_walk_failures(gpu):
    """BASELINE config 3 at full size through the certified entry: the failure counter stays 0, the inverse matches
    the analytic one, and the fused negation equals the two-step form bit for bit."""
    of = gpu
    from oflibnumpy_amd import device as dev
    shape = [2160, 3840]
    for tr, inv_tr in (([['scaling', 1000, 800, 0.9]], [['scaling', 1000, 800, 1 / 0.9]]),
                       ([['rotation', 1920, 1080, -20]], [['rotation', 1920, 1080, 20]])):
        f = of.Flow.from_transforms(tr, shape, 's')
        d = f.to_device()
        c = d.mesh_cert(+1)
        assert c.certified == 1, (tr, c.folded_cells, c.bad_edges, c.border_dev)
        h, w = shape
        out, valid, cnt = dev.DeviceBuffer(h * w * 8), dev.DeviceBuffer(h * w), dev.DeviceBuffer.zeros(16)
        of.native.check(of.native.load().ofl_scatter_certified_dev(d.vecs.ptr, 1, 0, d.vecs.ptr, 2, d.mask.ptr, h, w, 0, h, out.ptr,
                                                                   valid.ptr, of.native.SCATTER_NEGATE, ctypes.byref(c), cnt.ptr, None))
        assert int(cnt.to_host((1,), np.uint32)[0]) == 0
        got, gm = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
        want = of.Flow.from_transforms(inv_tr, shape, 's')
        assert gm.mean() > 0.5
        np.testing.assert_allclose(got[gm], want.vecs[gm], rtol=1e-4, atol=2e-4)
        inv = d.invert()
        v2, m2 = inv.to_host()
        np.testing.assert_array_equal(v2, got)
        np.testing.assert_array_equal(m2, gm)
        two_step = d.apply(-d).to_host()
        np.testing.assert_array_equal(two_step[0], got)
        np.testing.assert_array_equal(two_step[1], gm)


def test_nodes_in_the_noise_band_of_the_border(gpu, oracle):
    """A translation that puts the warped top border ON a row of nodes, plus 1e-6 px of noise: whether such a node
    is inside the convex hull is decided by the (almost collinear) border points, exactly as SciPy's hull does."""
    of, O = gpu, oracle
    rng = np.random.default_rng(8)
    shape = (24, 32)
    for k in range(6):
        vecs = np.zeros(shape + (2,), np.float32)
        vecs[..., 0] = 0.25
        vecs[..., 1] = 1.0
        vecs[0, :, 1] += (rng.random(shape[1]) - 0.5).astype(np.float32) * 2e-6      # top border at y = 1 +- 1e-6
        vecs[:, 0, 0] += (rng.random(shape[0]) - 0.5).astype(np.float32) * 2e-6
        c = certify(of, vecs)
        assert c.certified == 1 and 0 < c.border_dev < 1e-5
        f, o = of.Flow(vecs, 's'), O.OFlow(vecs, 's')
        np.testing.assert_array_equal(f.valid_target(), o.valid_target(), err_msg=str(k))
        img = rng.random(shape + (2,), dtype=np.float32)
        gw, gv = f.apply(img, return_valid_area=True)
        ow, ov = o.apply(img, return_valid_area=True)
        np.testing.assert_array_equal(gv, ov)
        # nodes on the noise band (row 1): SciPy interpolates inside a sliver triangle of height 1e-6 between border
        # points, the kernel along the sliver's chord -- both convex combinations of values of the border row, equal for
        # affine data; everything else agrees to rounding
        np.testing.assert_allclose(gw[2:], ow[2:], rtol=RTOL, atol=ATOL)
        lo, hi = img[:2].min((0, 1)) - 1e-6, img[:2].max((0, 1)) + 1e-6
        assert (gw[1][gv[1]] >= lo).all() and (gw[1][gv[1]] <= hi).all()
        gi, oi = f.invert(), o.invert()
        np.testing.assert_array_equal(gi.mask, oi.mask)
        np.testing.assert_allclose(gi.vecs[gi.mask], oi.vecs[oi.mask], rtol=RTOL, atol=ATOL)


def test_certified_walk_first_estimate_affine_and_not(gpu, oracle):
    """The walk's first estimate of a node's source cell: the inverse of the affine map through the warped corners when the
    certificate finds every point of the field within a quarter of a cell of that map (its "not affine" word stays 0), the
    Newton step on the node's own flow otherwise.  Both against SciPy through the oracle: a similarity transform (the short
    cut), the same field with a smooth interior bulge of 0.4 .. 3 cells that keeps the corners and the certificate (the
    fall-back -- with the short cut's estimate such nodes would start cells away), and a bulge of 0.1 cells (the short cut
    with estimates that are off by a fraction of a cell: the walk's own steps make up for it)."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    h, w = 150, 210
    yy, xx = np.mgrid[:h, :w].astype(np.float64)
    base = of.Flow.from_transforms([['rotation', 90, 70, 11], ['scaling', 100, 60, 0.93]], [h, w], 's').vecs.astype(np.float64)
    bump = np.sin(np.pi * xx / (w - 1)) ** 2 * np.sin(np.pi * yy / (h - 1)) ** 2          # 0 on the border (and flat there), 1 in the middle
    for amp, want_word in ((0.0, 0), (0.1, 0), (0.4, 1), (3.0, 1)):
        v = base.copy()
        v[..., 0] += amp * bump
        v[..., 1] -= 0.7 * amp * bump
        v = v.astype(np.float32)
        c = certify(of, v)
        assert c.certified == 1, (amp, c.folded_cells, c.bad_edges, c.border_dev)
        d = dev.DeviceFlow.from_host(v, 's')
        cert = d.mesh_cert(+1)
        assert cert.certified
        word = cert._diag_buf.to_host((h * ((w + 31) // 32),), np.uint32)[(h - 1) * ((w + 31) // 32)]
        assert int(word != 0) == want_word, (amp, word)
        f, o = of.Flow(v, 's'), O.OFlow(v, 's')
        gi, oi = f.invert(), o.invert()
        np.testing.assert_array_equal(gi.mask, oi.mask, err_msg=str(amp))
        np.testing.assert_allclose(gi.vecs[gi.mask], oi.vecs[oi.mask], rtol=RTOL, atol=ATOL, err_msg=str(amp))
        gs, os_ = f.switch_ref(), o.switch_ref()
        np.testing.assert_array_equal(gs.mask, os_.mask, err_msg=str(amp))
        np.testing.assert_allclose(gs.vecs[gs.mask], os_.vecs[os_.mask], rtol=RTOL, atol=ATOL, err_msg=str(amp))


# ---------------------------------------------------------------------------------------------- exact path
from scatter_util import nonunique_nodes, warped_points      # noqa: E402


def fixture_case(g, tag):
    vecs, mask, img = g[tag + '/in_vecs'], g[tag + '/in_mask'], g[tag + '/img']
    h, w = vecs.shape[:2]
    yy, xx = np.mgrid[:h, :w]
    pts = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)[mask.ravel()]
    return vecs, mask, img, pts


EXACT_TAGS = ["affine_generic", "affine_generic_hole", "block_generic", "curved", "curved_in", "hole_img", "shear",
              "speckle_img", "wobble3", "sintel4x4"]


@pytest.mark.parametrize("tag", EXACT_TAGS)
def test_exact_path_matches_reference_outputs(gpu, golden2, tag, near2_always):
    """Outputs of the REAL reference (tests/golden/make_golden.py::main_delaunay) for fields whose cell-wise mesh is
    not the Delaunay triangulation: folds (incl. BASELINE config 5 as loaded, 4 x 4 tiles), holes with random image
    values, curved borders, sheared cells, speckled point masks.  Masks bit-exact everywhere; values within 1e-4
    wherever SciPy's triangulation is unique."""
    of = gpu
    g = golden2
    vecs, mask, img, pts = fixture_case(g, tag)
    f = of.Flow(vecs, 's', mask)
    amb, inside = nonunique_nodes(pts, vecs.shape[:2])
    w, v = f.apply(img, return_valid_area=True)
    np.testing.assert_array_equal(v, g[tag + '/apply_valid'], err_msg=tag)
    np.testing.assert_array_equal(f.valid_target(), g[tag + '/valid_target'], err_msg=tag)
    bad = ~np.isclose(w, g[tag + '/apply'], rtol=RTOL, atol=ATOL).all(-1)
    assert not (bad & ~amb).any(), (tag, int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:5].tolist())
    r = f.invert()
    np.testing.assert_array_equal(r.mask, g[tag + '/invert_mask'], err_msg=tag)
    badv = ~np.isclose(r.vecs, g[tag + '/invert_vecs'], rtol=RTOL, atol=ATOL).all(-1)
    assert not (badv & ~amb).any(), (tag, int((badv & ~amb).sum()))
    if tag not in ("sintel4x4", "hole_img"):
        assert amb.mean() < 0.12, (tag, amb.mean())           # the comparison above is not vacuous


def test_exact_path_bands_and_determinism(gpu, golden2, near2_always):
    """Row bands of the exact path concatenate to the full result bit for bit, and repeated calls give the same bits
    (bucket order, far-point order and triangle ids are functions of the input only)."""
    of = gpu
    from oflibnumpy_amd import device as dev, sharding
    rng = np.random.default_rng(77)
    for tag in ("block_generic", "speckle_img", "sintel4x4", "curved"):
        vecs, mask, img, _ = fixture_case(golden2, tag)
        h, w = vecs.shape[:2]
        f, vals = dev.DeviceBuffer.from_host(vecs), dev.DeviceBuffer.from_host(img)
        pm = dev.DeviceBuffer.from_host(mask.astype(np.uint8)) if not mask.all() else None
        vm = dev.DeviceBuffer.from_host((rng.random((h, w)) > 0.1).astype(np.uint8))
        out, valid = dev.DeviceBuffer(h * w * 12), dev.DeviceBuffer(h * w)
        info = dev.scatter_linear(f, +1, pm, vals, 3, vm, h, w, None, out, valid, 0)
        assert info[0] == int(mask.sum())
        full, fullv = out.to_host((h, w, 3), np.float32), valid.to_host((h, w), np.uint8)
        dev.scatter_linear(f, +1, pm, vals, 3, vm, h, w, None, out, valid, 0)
        np.testing.assert_array_equal(out.to_host((h, w, 3), np.float32), full)
        for world in (3, 5):
            parts, vparts = [], []
            for r in range(world):
                r0, r1 = sharding.row_band(h, r, world)
                if r1 <= r0:
                    continue
                ob, vb = dev.DeviceBuffer((r1 - r0) * w * 12), dev.DeviceBuffer((r1 - r0) * w)
                dev.scatter_rows(f, +1, pm, vals, 3, vm, h, w, r0, r1 - r0, ob, vb)
                parts.append(ob.to_host((r1 - r0, w, 3), np.float32)); vparts.append(vb.to_host((r1 - r0, w), np.uint8))
            np.testing.assert_array_equal(np.concatenate(parts), full, err_msg=tag)
            np.testing.assert_array_equal(np.concatenate(vparts), fullv, err_msg=tag)


def test_both_paths_agree_at_full_size(gpu):
    """A generic affine field at 1080p through BOTH paths: the certificate + walk kernel, and -- with
    OFL_SCATTER_UNCERTIFIED, which makes the entry skip its certificate pass -- the Delaunay path (mesh fans for the
    interior, the clip and cooperative passes for the border).  The triangulation is unique here (no co-circular
    cells), so masks must be identical and values equal within the float32 tolerance; a hole and a 20-px tear are then
    cut into the same field and the Delaunay path must reproduce the walk result wherever the cut cannot reach."""
    of = gpu
    from oflibnumpy_amd import device as dev
    nat = of.native
    h, w = 1080, 1920
    vecs = affine_field((h, w), -0.07, 0.11, -0.065, 0.04, 30.0, -12.0)
    cert = certify(of, vecs)
    assert cert.certified
    rng = np.random.default_rng(5)
    vals = rng.random((h, w, 2), dtype=np.float32)
    vm = (rng.random((h, w)) > 0.05).astype(np.uint8)
    fb, vb, mb = dev.DeviceBuffer.from_host(vecs), dev.DeviceBuffer.from_host(vals), dev.DeviceBuffer.from_host(vm)
    out, valid = dev.DeviceBuffer(h * w * 8), dev.DeviceBuffer(h * w)
    info = dev.scatter_linear(fb, +1, None, vb, 2, mb, h, w, None, out, valid, 0)
    assert info[1] == 0                                                    # certified: nothing unfinished
    walk, walkv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8)
    info = dev.scatter_linear(fb, +1, None, vb, 2, mb, h, w, None, out, valid, nat.SCATTER_UNCERTIFIED)
    assert info[0] == h * w and 0 < info[1] < 4 * (h + w)                  # the Delaunay path ran: border points unfinished
    exact, exactv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8)
    np.testing.assert_array_equal(exactv, walkv)
    np.testing.assert_allclose(exact, walk, rtol=RTOL, atol=ATOL)
    # dropped points: away from them the triangulation is the same
    pm = np.ones((h, w), np.uint8)
    pm[300:420, 500:800] = 0
    pm[:, 1200:1202] = 0
    info = dev.scatter_linear(fb, +1, dev.DeviceBuffer.from_host(pm), vb, 2, mb, h, w, None, out, valid, 0)
    assert info[0] == int(pm.sum())
    cut, cutv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8)
    yy, xx = np.mgrid[:h, :w]
    px, py = xx + vecs[..., 0], yy + vecs[..., 1]                           # where the sources land
    far = np.ones((h, w), bool)
    for ys, xs in ((slice(298, 423), slice(498, 803)), (slice(0, h), slice(1198, 1205))):
        x0, x1, y0, y1 = px[ys, xs].min() - 3, px[ys, xs].max() + 3, py[ys, xs].min() - 3, py[ys, xs].max() + 3
        far &= ~((xx >= x0) & (xx <= x1) & (yy >= y0) & (yy <= y1))
    assert far.mean() > 0.8
    np.testing.assert_array_equal(cutv[far], walkv[far])
    np.testing.assert_allclose(cut[far], walk[far], rtol=RTOL, atol=ATOL)
    assert (cutv[~far] == 1).mean() > 0.5                                  # the hole and the tear are triangulated across


def test_notch_open_to_the_border(gpu, oracle, near2_always):
    """A 110 x 170 notch of dropped points that is OPEN to the image border: the two border points at its mouth are hull
    points (workgroup pass), the rim points at its bottom close only after a dozen coarse rings (wave pass), and they are
    Delaunay neighbours of each other across more than the coarse rings the workgroup pass searches -- the wave pass
    leaves such wide-reaching stars in the candidate list of the sweeps.  Whole result against SciPy."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    rng = np.random.default_rng(41)
    h, w = 230, 330
    vecs = affine_field((h, w), 0.06, 0.11, -0.07, 0.045, 4.0, -3.0)
    pm = np.ones((h, w), bool)
    pm[:110, 80:250] = False
    pm[150:, 140:143] = False                                                # and a slit open to the bottom border
    vals = rng.random((h, w, 2), dtype=np.float32)
    vm = rng.random((h, w)) > 0.1
    f, dv, dm, dpm = (dev.DeviceBuffer.from_host(a) for a in (vecs, vals, vm.astype(np.uint8), pm.astype(np.uint8)))
    out, valid = dev.DeviceBuffer(h * w * 8), dev.DeviceBuffer(h * w)
    dev.scatter_linear(f, +1, dpm, dv, 2, dm, h, w, None, out, valid, 0)
    got, gv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
    want = O.scatter_griddata(vecs, np.concatenate([vals, vm[..., None].astype(np.float32)], -1), pm)
    yy, xx = np.mgrid[:h, :w]
    pts = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)[pm.ravel()]
    amb, inside = nonunique_nodes(pts, (h, w))
    assert amb.mean() < 0.01
    np.testing.assert_array_equal(gv[~amb], (want[..., -1] == 1)[~amb])
    bad = ~np.isclose(got, want[..., :2], rtol=RTOL, atol=ATOL).all(-1)
    assert not (bad & ~amb).any(), (int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:4].tolist())
    assert gv[20:90, 100:230].mean() > 0.5                                   # the notch is triangulated across (inside the hull)


def test_garbage_vectors_do_not_hurt(gpu, golden2):
    """Behind the bare C ABI nothing has validated the vectors (the reference's Flow refuses NaN / Inf): points with a
    non-finite position are dropped like masked-out points -- the result is bit-identical to the call that masks them --
    and a handful of 1e9 vectors ("unknown flow" in some files) neither blow the buckets up to thousands of sites (the
    call stays fast) nor change the result away from the hull."""
    import time
    of = gpu
    from oflibnumpy_amd import device as dev
    vecs, mask, img, _ = fixture_case(golden2, "curved")
    h, w = vecs.shape[:2]
    rng = np.random.default_rng(8)
    bad = rng.random((h, w)) < 0.01
    bad[0, 0] = bad[h // 2, w // 2] = True
    v2 = vecs.copy()
    v2[bad] = rng.choice(np.float32([np.nan, np.inf, -np.inf]), (int(bad.sum()), 2))
    vals = dev.DeviceBuffer.from_host(img)
    res = []
    for fv, pm in ((v2, None), (vecs, ~bad)):
        out, valid = dev.DeviceBuffer(h * w * 12), dev.DeviceBuffer(h * w)
        info = dev.scatter_linear(dev.DeviceBuffer.from_host(fv), +1, None if pm is None else dev.DeviceBuffer.from_host(pm.astype(np.uint8)),
                                  vals, 3, None, h, w, None, out, valid, 0)
        assert info[0] == int((~bad).sum())
        res.append((out.to_host((h, w, 3), np.float32), valid.to_host((h, w), np.uint8)))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # outliers at 1080p: time and the interior
    H, W = 1080, 1920
    big = affine_field((H, W), 0.05, 0.02, -0.03, 0.04, 2.0, 1.0)
    img2 = rng.random((H, W, 1), dtype=np.float32)
    fb, vb = dev.DeviceBuffer.from_host(big), dev.DeviceBuffer.from_host(img2)
    out, valid = dev.DeviceBuffer(H * W * 4), dev.DeviceBuffer(H * W)
    dev.scatter_linear(fb, +1, None, vb, 1, None, H, W, None, out, valid, of.native.SCATTER_UNCERTIFIED)
    clean = out.to_host((H, W, 1), np.float32)
    wild = big.copy()
    for (y, x) in ((5, 7), (500, 900), (1079, 1919), (700, 3)):
        wild[y, x] = [1e9, -3e8]
    for k in range(300):                                                     # and a spray of them in every direction: long slivers
        ang = 2 * np.pi * k / 300                                            # from the hull to far away, each with the frame as its box
        wild[3 + (k * 7) % (H - 6), 40 + (k * 131) % (W - 80)] = [4e5 * np.cos(ang), 4e5 * np.sin(ang)]
    t0 = time.perf_counter()
    dev.scatter_linear(dev.DeviceBuffer.from_host(wild), +1, None, vb, 1, None, H, W, None, out, valid, 0)
    got = out.to_host((H, W, 1), np.float32)
    assert time.perf_counter() - t0 < 2.0
    far = np.ones((H, W), bool)
    far[:40] = far[-40:] = False
    far[:, :40] = far[:, -40:] = False
    far[480:560, 880:960] = False
    for k in range(300):                                                     # (where the displaced points USED to land)
        y, x = 3 + (k * 7) % (H - 6), 40 + (k * 131) % (W - 80)
        ly, lx = int(round(y + big[y, x, 1])), int(round(x + big[y, x, 0]))
        far[max(ly - 3, 0):ly + 4, max(lx - 3, 0):lx + 4] = False
    np.testing.assert_allclose(got[far], clean[far], rtol=RTOL, atol=ATOL)
    assert valid.to_host((H, W), np.uint8)[far].all()


def test_degenerate_point_sets_are_refused_not_endured(gpu):
    """Qhull refuses a flat point set ("initial simplex is flat" -- the reference raises QhullError, a RuntimeError); here a
    field that maps a million points onto one spot, or onto one line, is reported (OFL_E_INVALID, flag 16) instead of being
    ground through -- quickly, and with an all-invalid result.  The same for a small flat set."""
    import time
    of = gpu
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    h, w = 1080, 1920
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    ws = dev._workspace(h, w, 1)
    vals = dev.DeviceBuffer.from_host(np.ones((h, w, 1), np.float32))
    for name, vecs in (("one spot", np.stack([100 - xx, 50 - yy], -1)), ("one line", np.stack([0 * xx, 200 - yy], -1))):
        f = dev.DeviceBuffer.from_host(np.ascontiguousarray(vecs, np.float32))
        out, valid = dev.DeviceBuffer(h * w * 4), dev.DeviceBuffer(h * w)
        info = (ctypes.c_uint64 * 3)()
        t0 = time.perf_counter()
        rc = lib.ofl_scatter_linear_dev(f.ptr, 1, 0, None, vals.ptr, 1, None, h, w, None, out.ptr, valid.ptr, 0, ws.ptr, ws.nbytes, info, None)
        assert rc == nat.E_INVALID and time.perf_counter() - t0 < 20.0, (name, rc, time.perf_counter() - t0)
        assert not valid.to_host((h, w), np.uint8).any(), name
    hs, wsm = 12, 16
    ys, xs = np.mgrid[:hs, :wsm].astype(np.float32)
    f = dev.DeviceBuffer.from_host(np.ascontiguousarray(np.stack([0 * xs, 5 - ys], -1), np.float32))
    out, valid = dev.DeviceBuffer(hs * wsm * 4), dev.DeviceBuffer(hs * wsm)
    with pytest.raises(RuntimeError):
        dev.scatter_linear(f, +1, None, dev.DeviceBuffer.from_host(np.ones((hs, wsm, 1), np.float32)), 1, None, hs, wsm, None, out, valid, 0)
    assert not valid.to_host((hs, wsm), np.uint8).any()


def test_block_collapsed_onto_one_pixel_matches_scipy(gpu, oracle):
    """A flow that sends a whole 70 x 70 block of a 96 x 128 image to ONE position is legal input to Flow.apply: scipy's
    griddata (Qhull option Qc: coincident points are one vertex) triangulates the 7 389 distinct sites, and so does the
    Delaunay path -- 4 900 coincident points are one bucket far beyond the pairwise dedupe, which a hash table of positions
    reduces to its smallest index.  Masks bit-exact; values exact wherever SciPy's triangulation is unique and does not
    touch the collapsed vertex (whose value is that of whichever duplicate Qhull happened to keep)."""
    of, O = gpu, oracle
    rng = np.random.default_rng(31)
    shape = (96, 128)
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float32)
    vecs = of.from_transforms([['rotation', 60, 50, 7], ['scaling', 30, 40, 1.04]], list(shape), 's') + wobble(shape, 0.6, 0.5)
    vecs[10:80, 20:90, 0] = 55.25 - xx[10:80, 20:90]
    vecs[10:80, 20:90, 1] = 45.5 - yy[10:80, 20:90]
    img = rng.random(shape + (3,), dtype=np.float32)
    f = of.Flow(vecs, 's')
    got, valid = f.apply(img, return_valid_area=True)
    want, wvalid = O.OFlow(vecs, 's').apply(img, return_valid_area=True)
    np.testing.assert_array_equal(valid, wvalid)
    np.testing.assert_array_equal(f.valid_target(), wvalid)
    assert valid.mean() > 0.5
    amb, inside = nonunique_nodes(warped_points(vecs), shape)
    assert 0.2 < amb.mean() < 0.5 and (inside & ~amb).mean() > 0.3    # the fan around the collapsed vertex fills the block's old place; the rest is unique
    bad = ~np.isclose(got, want, rtol=RTOL, atol=ATOL).all(-1)
    assert not (bad & ~amb).any(), (int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:5].tolist())
    # a second, smaller collapse (below the hash-table threshold: pairwise dedupe) next to it, and determinism
    vecs[82:92, 100:112, 0] = 108.5 - xx[82:92, 100:112]
    vecs[82:92, 100:112, 1] = 88.25 - yy[82:92, 100:112]
    f = of.Flow(vecs, 's')
    np.testing.assert_array_equal(f.valid_target(), O.OFlow(vecs, 's').valid_target())
    a, b = f.apply(img), f.apply(img)
    np.testing.assert_array_equal(a, b)


def contracted_block(shape, blk, factor, centre):
    """background: a rotation + scaling with a ripple; the block `blk` of the grid contracted `factor` times around `centre`
    (every site distinct, the contracted lattice rippled so that none of its cells is co-circular)"""
    import oflibnumpy_amd as of
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float32)
    vecs = of.from_transforms([['rotation', shape[1] * 0.47, shape[0] * 0.52, 7], ['scaling', shape[1] * 0.23, shape[0] * 0.42, 1.04]], list(shape), 's')
    vecs = vecs + wobble(shape, 0.6, 0.5)
    ys, xs = yy[blk].astype(np.float64), xx[blk].astype(np.float64)
    rip = 1.0 + 0.05 * np.sin(xs / 5.3 + ys / 7.1)
    vecs[blk + (0,)] = (centre[0] + (xs - xs.mean()) / factor * rip - xs).astype(np.float32)
    vecs[blk + (1,)] = (centre[1] + (ys - ys.mean()) / factor * rip - ys).astype(np.float32)
    return np.ascontiguousarray(vecs, np.float32)


@pytest.mark.parametrize("factor", [20.0, 100.0])
def test_dense_clusters_of_distinct_sites_match_scipy(gpu, oracle, factor):
    """A flow that CONTRACTS a 60 x 60 block of a 96 x 128 image 20 or 100 times puts 3 600 distinct sites into nine buckets -- or
    one -- of the uniform grid; scipy's griddata (utils.py:253) triangulates them like any other point set.  Buckets of more
    than 64 entries get grids of their own (dl_sub_bin_kernel; ofl_dl::apply_heavy_run) instead of counting against a limit:
    masks bit-exact, values within 1e-4 wherever SciPy's triangulation is unique, with and without a point mask, and the
    same bits on a second run (the order of a cluster's sites is fixed by index, not by the fill pass's atomics)."""
    of, O = gpu, oracle
    rng = np.random.default_rng(43)
    shape = (96, 128)
    blk = (slice(20, 80), slice(30, 90))
    vecs = contracted_block(shape, blk, factor, (61.3, 47.7))
    img = rng.random(shape + (3,), dtype=np.float32)
    for pm in (None, rng.random(shape) > 0.1):
        f = of.Flow(vecs, 's', pm)
        got, valid = f.apply(img, return_valid_area=True)
        want, wvalid = O.OFlow(vecs, 's', pm).apply(img, return_valid_area=True)
        np.testing.assert_array_equal(valid, wvalid)
        keep = None if pm is None else pm
        amb, inside = nonunique_nodes(warped_points(vecs, keep), shape)
        assert (inside & ~amb).mean() > 0.5
        bad = ~np.isclose(got, want, rtol=RTOL, atol=ATOL).all(-1) & ~amb
        assert not bad.any(), (factor, pm is not None, int(bad.sum()), np.argwhere(bad)[:5].tolist())
        np.testing.assert_array_equal(f.apply(img), got)
    # the nodes INSIDE the cluster are covered by its tiny triangles: the warped image there is the block's content, shrunk
    cy, cx = 47.7, 61.3
    assert valid[int(cy), int(cx)] and valid[int(cy) + 1, int(cx) + 1]


def test_dense_cluster_at_4k_is_triangulated_in_milliseconds(gpu):
    """2160 x 3840 static field with a 400 x 400 block contracted 100 times (160 000 distinct sites in 4 x 4 px, ten thousand per
    bucket; round 3 refused it: more than 65 536 distinct sites in buckets of more than 4 096): OFL_OK, every site kept, the
    block's old place filled by the fan around the cluster, in well under 50 ms"""
    import time
    of = gpu
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    h, w = 2160, 3840
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    vecs = np.zeros((h, w, 2), np.float32)
    blk = (slice(700, 1100), slice(1500, 1900))
    ys, xs = yy[blk].astype(np.float64), xx[blk].astype(np.float64)
    rip = 1.0 + 0.05 * np.sin(xs / 5.3 + ys / 7.1)
    vecs[blk + (0,)] = (1700.37 + (xs - xs.mean()) / 100.0 * rip - xs).astype(np.float32)
    vecs[blk + (1,)] = (900.61 + (ys - ys.mean()) / 100.0 * rip - ys).astype(np.float32)
    f = dev.DeviceBuffer.from_host(vecs)
    vals = dev.DeviceBuffer.from_host((xx + 2 * yy)[..., None].astype(np.float32))
    out, valid = dev.DeviceBuffer(h * w * 4), dev.DeviceBuffer(h * w)
    info = dev.scatter_linear(f, +1, None, vals, 1, None, h, w, None, out, valid, nat.SCATTER_UNCERTIFIED)      # warm-up + the counts
    assert info[0] == h * w
    nat.check(lib.ofl_stream_sync(None))
    t0 = time.perf_counter()
    for _ in range(3):
        dev.scatter_linear(f, +1, None, vals, 1, None, h, w, None, out, valid, nat.SCATTER_UNCERTIFIED)
    nat.check(lib.ofl_stream_sync(None))
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print("4K field with a 400 x 400 block contracted 100 x: {:.2f} ms, unfinished {}, left over {}".format(ms, info[1], info[2]))
    v = valid.to_host((h, w), np.uint8)
    assert v.all()                                         # the hull is the image; the block's old place is covered by large triangles
    o = out.to_host((h, w), np.float32)
    far = np.ones((h, w), bool)
    far[690:1110, 1490:1910] = False
    np.testing.assert_allclose(o[far], (xx + 2 * yy)[far], rtol=1e-6, atol=1e-3)       # away from the block the field is the identity
    assert ms < 50.0, ms


def test_delaunay_path_without_counts(gpu, golden2):
    """info_host == NULL: the Delaunay path sizes every launch on the device and reads nothing back (the entry only
    enqueues); the result is bit-identical to the call that asks for the counts.  What can only be known after the
    fact -- no point kept at all -- is reported to callers that ask, and is an all-invalid result for those that do not."""
    of = gpu
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    vecs, mask, img, _ = fixture_case(golden2, "block_generic")
    h, w = vecs.shape[:2]
    f, vals = dev.DeviceBuffer.from_host(vecs), dev.DeviceBuffer.from_host(img)
    ws = dev._workspace(h, w, 3)
    outs = []
    for with_info in (True, False):
        out, valid = dev.DeviceBuffer(h * w * 12), dev.DeviceBuffer(h * w)
        info = (ctypes.c_uint64 * 3)()
        nat.check(lib.ofl_scatter_linear_dev(f.ptr, 1, 0, None, vals.ptr, 3, None, h, w, None, out.ptr, valid.ptr, 0, ws.ptr, ws.nbytes,
                                             info if with_info else None, None))
        outs.append((out.to_host((h, w, 3), np.float32), valid.to_host((h, w), np.uint8)))
        assert (info[0] == h * w and info[1] > 0) if with_info else info[0] == 0
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    none = dev.DeviceBuffer.from_host(np.zeros((h, w), np.uint8))
    out, valid = dev.DeviceBuffer(h * w * 12), dev.DeviceBuffer(h * w)
    info = (ctypes.c_uint64 * 3)()
    assert lib.ofl_scatter_linear_dev(f.ptr, 1, 0, none.ptr, vals.ptr, 3, None, h, w, None, out.ptr, valid.ptr, 0, ws.ptr, ws.nbytes,
                                      info, None) == nat.E_NOPOINTS
    nat.check(lib.ofl_scatter_linear_dev(f.ptr, 1, 0, none.ptr, vals.ptr, 3, None, h, w, None, out.ptr, valid.ptr,
                                         nat.SCATTER_UNCERTIFIED, ws.ptr, ws.nbytes, None, None))
    assert not valid.to_host((h, w), np.uint8).any() and not out.to_host((h, w, 3), np.float32).any()


def test_exact_path_medium_field_against_scipy(gpu, oracle, near2_always):
    """One field large enough (240 x 320) for every star pass to see real work -- a smooth non-affine warp with a 15-px
    tear along a slanted line (folds on one side, a gap on the other), a 40 x 60 hole and 3 % speckle in the point mask,
    random image values and a speckled value mask: the whole result equals SciPy's outside non-unique simplices."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    rng = np.random.default_rng(99)
    h, w = 240, 320
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    vecs = np.stack([2.5 * np.sin(xx / 23 + yy / 31) + 0.11 * yy + 0.04 * xx,
                     2.0 * np.cos(xx / 19) * np.sin(yy / 27) - 0.06 * xx + 0.03 * yy], -1).astype(np.float32)
    vecs[(xx + 0.4 * yy) > 210] += np.float32([15.0, -6.0])                 # the tear
    pm = rng.random((h, w)) > 0.03
    pm[90:130, 60:120] = False
    vals = rng.random((h, w, 2), dtype=np.float32)
    vm = rng.random((h, w)) > 0.1
    f, dv, dm, dpm = (dev.DeviceBuffer.from_host(a) for a in (vecs, vals, vm.astype(np.uint8), pm.astype(np.uint8)))
    out, valid = dev.DeviceBuffer(h * w * 8), dev.DeviceBuffer(h * w)
    info = dev.scatter_linear(f, +1, dpm, dv, 2, dm, h, w, None, out, valid, 0)
    assert info[0] == int(pm.sum()) and info[1] > 500                       # rims of the tear and the hole, the border
    got, gv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
    want = O.scatter_griddata(vecs, np.concatenate([vals, vm[..., None].astype(np.float32)], -1), pm)
    pts = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)[pm.ravel()]
    amb, inside = nonunique_nodes(pts, (h, w))
    assert amb.mean() < 0.02
    np.testing.assert_array_equal(gv[~amb], (want[..., -1] == 1)[~amb])
    bad = ~np.isclose(got, want[..., :2], rtol=RTOL, atol=ATOL).all(-1)
    assert not (bad & ~amb).any(), (int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:4].tolist())


def test_wavy_border_against_scipy(gpu, oracle):
    """The benchmark's non-affine case at 480 x 640 (tools/bench_ops.py: a rotation and scaling plus SURVEY 8(d)'s 3-px
    sinusoid, 5 % of the points dropped): the image border is a wavy line whose bays are bridged by sliver triangles with
    circumcentres hundreds of pixels out -- the cells the far threshold, the far cone and the sweep over nine chunks of
    left-over points are there for.  The whole result equals SciPy's outside non-unique simplices."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    h, w = 480, 640
    f = of.Flow.from_transforms([['rotation', w / 2, h / 2, -20], ['scaling', w / 3.84, h / 2.7, 0.9]], [h, w], 's')
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    vecs = f.vecs.copy()
    vecs[..., 0] += 3.0 * np.sin(2 * np.pi * xx / 97) * np.cos(2 * np.pi * yy / 131)
    vecs[..., 1] += 3.0 * np.cos(2 * np.pi * xx / 97) * np.sin(2 * np.pi * yy / 131)
    rng = np.random.default_rng(5)
    pm = rng.random((h, w)) > 0.05
    vals = rng.random((h, w, 2), dtype=np.float32)
    fb, dv, dpm = (dev.DeviceBuffer.from_host(a) for a in (vecs, vals, pm.astype(np.uint8)))
    out, valid = dev.DeviceBuffer(h * w * 8), dev.DeviceBuffer(h * w)
    info = dev.scatter_linear(fb, +1, dpm, dv, 2, None, h, w, None, out, valid, 0)
    assert info[0] == int(pm.sum()) and info[2] > 1500                      # the border goes through the left-over pass
    got, gv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
    want = O.scatter_griddata(vecs, np.concatenate([vals, np.ones((h, w, 1), np.float32)], -1), pm)
    pts = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)[pm.ravel()]
    amb, inside = nonunique_nodes(pts, (h, w))
    assert amb.mean() < 0.02
    np.testing.assert_array_equal(gv[~amb], (want[..., -1] == 1)[~amb])
    bad = ~np.isclose(got, want[..., :2], rtol=RTOL, atol=ATOL).all(-1)
    assert not (bad & ~amb).any(), (int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:4].tolist())


def test_exact_path_random_fields_against_scipy(gpu, oracle, near2_always):
    """Seeded sweep on ragged shapes: smooth non-affine fields, folds, random point masks with holes, both signs, random
    image values -- the full result (values and validity of a random value mask) equals SciPy's outside non-unique
    simplices."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    rng = np.random.default_rng(2024)
    for it in range(24):
        h, w = int(rng.integers(5, 70)), int(rng.integers(5, 90))
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        amp = rng.uniform(0.2, 4)
        vecs = np.stack([amp * np.sin(xx / rng.uniform(3, 30) + yy / rng.uniform(5, 40)) + 0.13 * yy,
                         amp * np.cos(xx / rng.uniform(4, 25)) - 0.07 * xx], -1).astype(np.float32)
        vecs += rng.uniform(-5, 5, 2).astype(np.float32)
        if it % 3 == 0:
            vecs[rng.random((h, w)) < 0.03] += rng.uniform(3, 12)          # isolated outliers: folds
        pm = None
        if it % 2:
            pm = rng.random((h, w)) > rng.uniform(0, 0.3)
            if it % 4 == 1 and h > 12 and w > 12:
                pm[h // 4:h // 4 + h // 3, w // 5:w // 5 + w // 2] = False   # a hole
            if pm.sum() < 8:
                pm = None
        sign = 1 if it % 5 else -1
        C = int(rng.integers(1, 4))
        vals = rng.random((h, w, C), dtype=np.float32)
        vm = rng.random((h, w)) > 0.15
        f, dv, dm = dev.DeviceBuffer.from_host(vecs), dev.DeviceBuffer.from_host(vals), dev.DeviceBuffer.from_host(vm.astype(np.uint8))
        dpm = dev.DeviceBuffer.from_host(pm.astype(np.uint8)) if pm is not None else None
        out, valid = dev.DeviceBuffer(h * w * C * 4), dev.DeviceBuffer(h * w)
        dev.scatter_linear(f, sign, dpm, dv, C, dm, h, w, None, out, valid, 0)
        got, gv = out.to_host((h, w, C), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
        want = O.scatter_griddata(sign * vecs, np.concatenate([vals, vm[..., None].astype(np.float32)], -1), pm)
        keep = np.ones((h, w), bool) if pm is None else pm
        pts = np.stack([(xx + sign * vecs[..., 0].astype(np.float64)).ravel(), (yy + sign * vecs[..., 1].astype(np.float64)).ravel()], 1)[keep.ravel()]
        amb, inside = nonunique_nodes(pts, (h, w))
        covered = want[..., -1] != 0
        np.testing.assert_array_equal(gv[~amb], (want[..., -1] == 1)[~amb], err_msg=str((it, h, w)))
        bad = ~np.isclose(got, want[..., :C], rtol=RTOL, atol=ATOL).all(-1)
        assert not (bad & ~amb).any(), (it, h, w, int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:4].tolist())
        assert amb.mean() < 0.2, (it, amb.mean())


def test_certified_field_the_walk_cannot_follow(gpu, oracle):
    """A certificate proves that the mesh is the triangulation, not that the walk kernel's Newton steps reach every node's
    triangle: x' = g(x), y' = g(y) with g' between 0.03 and 1.97 keeps every cell a positively oriented rectangle (a tensor
    grid IS its own Delaunay triangulation) and the border straight, and squeezes a third of the image into a sliver.
    Whatever the walk kernel loses (its failure counter says how many nodes) is recomputed on the Delaunay path -- through
    the one-shot C entry and through the Flow API with its cached certificate alike -- and the result is SciPy's."""
    of, O = gpu, oracle
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    h, w = 64, 96
    yy, xx = np.mgrid[:h, :w].astype(np.float64)
    g = lambda t, L: -0.97 * np.sin(2 * np.pi * t / (L - 1)) * (L - 1) / (2 * np.pi)
    vecs = np.stack([g(xx, w), g(yy, h)], -1).astype(np.float32)
    c = certify(of, vecs)
    assert c.certified == 1, (c.folded_cells, c.bad_edges, c.border_dev)
    f = dev.DeviceBuffer.from_host(vecs)
    out, valid, cnt = dev.DeviceBuffer(h * w * 8), dev.DeviceBuffer(h * w), dev.DeviceBuffer.zeros(16)
    nat.check(lib.ofl_scatter_certified_dev(f.ptr, 1, 0, f.ptr, 2, None, h, w, 0, h, out.ptr, valid.ptr, 0, ctypes.byref(c), cnt.ptr, None))
    lost = int(cnt.to_host((1,), np.uint32)[0])
    print("walk kernel lost", lost, "of", h * w, "nodes on the squeezed tensor grid")
    want = O.scatter_griddata(vecs, np.concatenate([vecs, np.ones((h, w, 1), np.float32)], -1), None)
    # one-shot entry: certificate, walk, counter, Delaunay path if needed
    ws = dev._workspace(h, w, 2)
    info = (ctypes.c_uint64 * 3)()
    nat.check(lib.ofl_scatter_linear_dev(f.ptr, 1, 0, None, f.ptr, 2, None, h, w, None, out.ptr, valid.ptr, 0, ws.ptr, ws.nbytes, info, None))
    got, gv = out.to_host((h, w, 2), np.float32), valid.to_host((h, w), np.uint8).astype(bool)
    np.testing.assert_array_equal(gv, want[..., 2] == 1)
    np.testing.assert_allclose(got, want[..., :2], rtol=RTOL, atol=ATOL)
    # Flow API (cached certificate, checked on its first use)
    fl = of.Flow(vecs, 's')
    wv, wm = fl.apply(of.Flow(vecs, 's')).vecs, fl.valid_target()
    np.testing.assert_array_equal(wm, want[..., 2] == 1)
    np.testing.assert_allclose(wv, want[..., :2], rtol=RTOL, atol=ATOL)
    d = fl.to_device()
    r1, r2 = d.apply(d).to_host(), d.apply(d).to_host()          # second call: the cached decision
    np.testing.assert_array_equal(r1[0], r2[0])
    np.testing.assert_allclose(r1[0], want[..., :2], rtol=RTOL, atol=ATOL)
    assert lost == 0 or d.mesh_cert(+1).certified == 0


def test_diagonal_bit_plane_changes_nothing(gpu):
    """The certificate's per-cell diagonal bits (ofl_scatter_certify_dev, diag_bits) replace the walk kernel's per-node
    float64 in-circle determinant: the same predicate on the same numbers, so outputs with and without the plane are
    bit-identical -- on a similarity (every cell co-circular to rounding: the decision is a coin flip of the last bit, and
    must be the SAME coin), an exact lattice and a generic affine map, both signs, at a size with partial words per row."""
    of = gpu
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    rng = np.random.default_rng(3)
    shape = (61, 77)
    h, w = shape
    fields = [of.from_transforms([['rotation', 30, 25, 33]], list(shape), 's'), of.from_transforms([['translation', 3, -2]], list(shape), 's'),
              of.from_transforms([['scaling', 10, 20, 0.8], ['rotation', 40, 30, -17]], list(shape), 's'), affine_field(shape, 0.15, 0.2, -0.1, 0.25, 1.3, -0.4)]
    for vecs in fields:
        for sign in (1, -1):
            outs = []
            for with_bits in (True, False):
                c = certify(of, vecs, sign, with_bits=with_bits)
                assert c.certified == 1 and bool(c.diag_bits) == with_bits
                vals = rng.standard_normal((h, w, 3)).astype(np.float32) if not outs else vals
                f, dv = dev.DeviceBuffer.from_host(np.ascontiguousarray(vecs, np.float32)), dev.DeviceBuffer.from_host(vals)
                out, valid, cnt = dev.DeviceBuffer(h * w * 12), dev.DeviceBuffer(h * w), dev.DeviceBuffer.zeros(16)
                nat.check(lib.ofl_scatter_certified_dev(f.ptr, sign, 0, dv.ptr, 3, None, h, w, 0, h, out.ptr, valid.ptr, 0, ctypes.byref(c), cnt.ptr, None))
                assert int(cnt.to_host((1,), np.uint32)[0]) == 0
                outs.append((out.to_host((h, w, 3), np.float32), valid.to_host((h, w), np.uint8)))
            np.testing.assert_array_equal(outs[0][0], outs[1][0])
            np.testing.assert_array_equal(outs[0][1], outs[1][1])


def test_random_collapses_against_scipy(gpu, oracle):
    """Seeded sweep over fields in which rectangles of 2 .. 40 pixels a side collapse onto single positions -- some onto
    the position of a background site (duplicates across sheets), some onto each other, the largest beyond the pairwise
    dedupe (hash table): valid areas equal SciPy's bit for bit, values agree wherever its triangulation is unique."""
    of, O = gpu, oracle
    rng = np.random.default_rng(77)
    shape = (72, 96)
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float32)
    for it in range(6):
        vecs = np.zeros(shape + (2,), np.float32) if it % 2 == 0 else \
            (of.from_transforms([['rotation', 40, 30, 5.0 * it]], list(shape), 's') + wobble(shape, 0.4, 0.3))
        for _ in range(int(rng.integers(1, 4))):
            hh, ww = int(rng.integers(2, 41)), int(rng.integers(2, 41))
            y0, x0 = int(rng.integers(0, shape[0] - hh)), int(rng.integers(0, shape[1] - ww))
            # even rounds: onto a lattice node (an exact duplicate of a background site when that one stays put)
            ty, tx = (float(rng.integers(5, shape[0] - 5)), float(rng.integers(5, shape[1] - 5))) if it % 2 == 0 else \
                     (float(rng.uniform(5, shape[0] - 5)), float(rng.uniform(5, shape[1] - 5)))
            vecs[y0:y0 + hh, x0:x0 + ww, 0] = tx - xx[y0:y0 + hh, x0:x0 + ww]
            vecs[y0:y0 + hh, x0:x0 + ww, 1] = ty - yy[y0:y0 + hh, x0:x0 + ww]
        img = rng.random(shape + (2,), dtype=np.float32)
        f = of.Flow(vecs, 's')
        got, valid = f.apply(img, return_valid_area=True)
        want, wvalid = O.OFlow(vecs, 's').apply(img, return_valid_area=True)
        np.testing.assert_array_equal(valid, wvalid, err_msg="round {}".format(it))
        amb, inside = nonunique_nodes(warped_points(vecs), shape)
        bad = ~np.isclose(got, want, rtol=RTOL, atol=ATOL).all(-1)
        assert not (bad & ~amb).any(), (it, int((bad & ~amb).sum()), np.argwhere(bad & ~amb)[:5].tolist())


def test_soak_regressions(gpu, oracle):
    """Seeds of tools/soak_scatter.py that once differed from SciPy at a handful of nodes: similarity transforms and integer
    fields (every cell co-circular) under random point masks.  A bisector that passes exactly through two coincident cell
    vertices had them decided "cut" and "not cut" by the tie rule; the clip then removed that one-vertex run instead of the
    run the half-plane really takes away, the site's neighbour was never applied and its star disagreed with its neighbours'
    (ofl_dl::poly_cutmask, far_apply).  0 mismatching nodes now, here and in 12 million nodes of fresh seeds."""
    from oflibnumpy_amd import device as dev
    from scatter_soak_util import one_case
    from scatter_util import nonunique_nodes, hull_band
    total = 0
    for seed in (1000020, 1000058, 1000112, 1000200, 1000212, 1000344, 1000485, 1000492):
        n, bad, msg = one_case(dev, oracle, nonunique_nodes, hull_band, seed, 160, 240)
        assert bad == 0, msg
        total += n
    # larger fields: sites that are EXACTLY co-circular must be a tie for all four stars that ask (incircle_origin_filtered;
    # seed 3000265), and Qhull's own tolerance grows with the coordinates (seed 3000143: scatter_util.nonunique_nodes)
    for seed in (3000143, 3000265):
        n, bad, msg = one_case(dev, oracle, nonunique_nodes, hull_band, seed, 320, 420)
        assert bad == 0, msg
        total += n
    assert total > 150_000


def test_query_soak_regressions(gpu, oracle):
    """combine_with mode 2 / ref 't' on integer-valued fields (tools/soak_scatter.py --mode query): every cell is co-circular, so
    stars legitimately disagree on diagonals, and hull stars repeat a neighbour around collinear runs (a, b, a).  The
    visibility walk of the query path needs the edge it leaves through listed by the star it enters: it used to give up there
    and returned "outside" for 0.17 % of the positions SciPy finds inside -- up to long hull slivers 0.02 px wide.  Now: the
    other endpoint's star, then a geometric choice of the apex among both endpoints' neighbours, then the triangles of all
    sites within two bucket rings (dl_locate, dl_locate_brute): 0 of 5.2 M positions off."""
    from oflibnumpy_amd import device as dev
    from scatter_soak_util import one_query_case
    from scatter_util import nonunique_nodes, hull_band
    total = 0
    for seed in (5000011, 5000017, 5000022, 5000030, 5000043, 5000078):
        n, bad, msg = one_query_case(dev, oracle, nonunique_nodes, hull_band, seed, 160, 240)
        assert bad == 0, msg
        total += n
    assert total > 20_000

