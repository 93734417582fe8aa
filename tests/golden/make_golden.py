#!/usr/bin/env python
"""Generates tests/golden/ref_scipy_paths.npz by RUNNING THE REAL REFERENCE in the build container.

The reference (/root/reference, oflibnumpy 1.1.1) imports cv2 at module import time and cv2 is not
installed here (ordinary ModuleNotFoundError, no permission denial).  A stub module whose functions
raise is registered as `cv2` so that the package imports; every path that reaches cv2.remap therefore
raises and is NOT captured here.  What is captured is the reference's genuine arithmetic on every
SciPy-backed path of the hot path ('s' apply, invert s->s / t->t, switch_ref, valid_target('s'),
valid_source('t'), combine_with mode 2).

Only inputs and expected outputs are stored (data, not source).  The reference never travels to the
GPU box; the tests read the .npz only.

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    cv2 = types.ModuleType("cv2")

    def _unavailable(*a, **k):
        raise RuntimeError("cv2 is not available in this container")

    for name in ("remap", "imread", "resize", "cvtColor", "imshow", "waitKey", "findContours",
                 "drawContours", "estimateAffine2D", "estimateAffinePartial2D", "findHomography",
                 "arrowedLine", "line"):
        setattr(cv2, name, _unavailable)
    cv2.INTER_LINEAR = 1
    sys.modules["cv2"] = cv2
    sys.path.insert(0, "/root/reference/src")
    import oflibnumpy
    return oflibnumpy


def main():
    of = _import_reference()
    Flow = of.Flow
    out = {}
    rng = np.random.default_rng(7)

    def put(tag, flow, res):
        out[tag + "/in_vecs"] = flow.vecs
        out[tag + "/in_mask"] = flow.mask
        out[tag + "/in_ref"] = np.array(flow.ref)
        if isinstance(res, Flow):
            out[tag + "/out_vecs"] = res.vecs
            out[tag + "/out_mask"] = res.mask
            out[tag + "/out_ref"] = np.array(res.ref)
        else:
            out[tag + "/out"] = res

    shape = (24, 32)
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype('f')
    wobble = np.stack([0.5 * np.sin(2 * np.pi * xx / 17.0) * np.cos(2 * np.pi * yy / 13.0),
                       0.4 * np.cos(2 * np.pi * xx / 19.0) * np.sin(2 * np.pi * yy / 11.0)], -1).astype('f')
    hole = np.ones(shape, bool)
    hole[8:14, 10:18] = False
    speckle = rng.random(shape) > 0.1

    flows = {}
    for ref in ('s', 't'):
        flows["rot_" + ref] = Flow.from_transforms([['rotation', 12, 10, -20]], shape, ref)
        flows["scale_" + ref] = Flow.from_transforms([['scaling', 10, 8, 0.85]], shape, ref)
        flows["rotscale_" + ref] = Flow.from_transforms([['rotation', 16, 12, 15], ['scaling', 5, 5, 1.1]], shape, ref)
        flows["transl_" + ref] = Flow.from_transforms([['translation', 3, -2]], shape, ref)
        f = Flow.from_transforms([['rotation', 16, 12, -10], ['scaling', 16, 12, 0.95]], shape, ref)
        flows["wobble_" + ref] = Flow(f.vecs + wobble, ref)
        flows["rot_hole_" + ref] = Flow.from_transforms([['rotation', 12, 10, -20]], shape, ref, hole)
        flows["rot_speckle_" + ref] = Flow.from_transforms([['rotation', 12, 10, -20]], shape, ref, speckle)

    img = rng.random(shape + (3,), dtype=np.float32)
    img_u8 = (rng.random(shape + (3,)) * 255).astype(np.uint8)
    out["img_f32"] = img
    out["img_u8"] = img_u8

    for name, f in flows.items():
        put("invert/" + name, f, f.invert())
        put("switch_ref/" + name, f, f.switch_ref())
        if f.ref == 's':
            put("valid_target/" + name, f, f.valid_target())
            put("valid_target_nomask/" + name, f, f.valid_target(consider_mask=False))
            w, v = f.apply(img, return_valid_area=True)
            put("apply_img/" + name, f, w)
            out["apply_img/" + name + "/out_valid"] = v
            w, v = f.apply(img, return_valid_area=True, consider_mask=False)
            put("apply_img_nomask/" + name, f, w)
            out["apply_img_nomask/" + name + "/out_valid"] = v
            put("apply_u8/" + name, f, f.apply(img_u8))
        else:
            put("valid_source/" + name, f, f.valid_source())
            put("valid_source_nomask/" + name, f, f.valid_source(consider_mask=False))

    # combine_with mode 2 (both refs): f1 (+) f2 = f3, recover f2
    for ref in ('s', 't'):
        t1 = [['rotation', 16, 12, -12]]
        t2 = [['scaling', 8, 6, 0.9]]
        f1 = Flow.from_transforms(t1, shape, ref)
        f3 = Flow.from_transforms(t1 + t2, shape, ref)
        r = f1.combine_with(f3, 2)
        put("combine2/" + ref, f1, r)
        out["combine2/" + ref + "/in2_vecs"] = f3.vecs
        out["combine2/" + ref + "/in2_mask"] = f3.mask
        f1w = flows["wobble_" + ref]
        r = f1w.combine_with(f3, 2)
        put("combine2_wobble/" + ref, f1w, r)
        out["combine2_wobble/" + ref + "/in2_vecs"] = f3.vecs
        out["combine2_wobble/" + ref + "/in2_mask"] = f3.mask

    # 7x7 known-answer geometry of tests/test_flow_class.py:852-980, captured from the reference itself
    s7 = (7, 7)
    m_s = np.ones(s7, bool)
    m_s[4:, :3] = False
    m_t = np.ones(s7, bool)
    m_t[:3, 4:] = False
    f_s = Flow.from_transforms([['rotation', 0, 0, 45]], s7, 's')
    f_sm = Flow.from_transforms([['rotation', 0, 0, 45]], s7, 's', m_s)
    f_t = Flow.from_transforms([['rotation', 0, 0, 45]], s7, 't')
    f_tm = Flow.from_transforms([['rotation', 0, 0, 45]], s7, 't', m_t)
    put("k7/valid_target_s", f_s, f_s.valid_target())
    put("k7/valid_target_s_masked", f_sm, f_sm.valid_target())
    put("k7/valid_target_s_masked_nomask", f_sm, f_sm.valid_target(False))
    put("k7/valid_source_t", f_t, f_t.valid_source())
    put("k7/valid_source_t_masked", f_tm, f_tm.valid_source())
    put("k7/valid_source_t_masked_nomask", f_tm, f_tm.valid_source(False))

    # sparse point tracking (utils.py:547-622, flow_class.py:755-795) -- appended last so that the seeded
    # arrays above stay byte-identical
    tshape = (64, 80)
    pts_f = np.array([[20.5, 10.5], [8.3, 7.2], [50.4, 60.2], [0.0, 0.0], [63.0, 79.0], [31.25, 40.75]])
    pts_i = np.array([[20, 10], [8, 7], [63, 79], [0, 0]])
    tyy, txx = np.mgrid[:tshape[0], :tshape[1]].astype('f')
    twob = np.stack([0.8 * np.sin(txx / 9.0) * np.cos(tyy / 7.0), 0.6 * np.cos(txx / 8.0) * np.sin(tyy / 6.0)], -1).astype('f')
    out["track/pts_f"] = pts_f
    out["track/pts_i"] = pts_i
    for name, tr, extra in (("rot", [['rotation', 0, 0, 12]], None), ("wob", [['rotation', 30, 40, -8], ['scaling', 30, 40, 0.95]], twob)):
        for ref in ('s', 't'):
            fv = of.from_transforms(tr, tshape, ref)
            if extra is not None:
                fv = (fv + extra).astype('f')
            tag = "track/{}_{}".format(name, ref)
            out[tag + "/flow"] = fv
            out[tag + "/float"] = of.track_pts(fv, ref, pts_f)
            out[tag + "/float_int_out"] = of.track_pts(fv, ref, pts_f, int_out=True)
            if ref == 's':
                out[tag + "/int"] = of.track_pts(fv, ref, pts_i)
                out[tag + "/exact"] = of.track_pts(fv, ref, pts_f, s_exact_mode=True)
    ft = Flow.from_transforms([['rotation', 0, 0, 12]], tshape, 't')
    ft.mask[:, 50:] = False
    status_pts = np.array([[0, 20], [0, 75], [8.3, 7.2], [40.4, 30.2], [50, 60]])
    warped, status = ft.track(status_pts, get_valid_status=True)
    out["track/status_t/flow"] = ft.vecs
    out["track/status_t/mask"] = ft.mask
    out["track/status_t/pts"] = status_pts
    out["track/status_t/warped"] = warped
    out["track/status_t/status"] = status

    # discontinuous fields (motion boundaries): a block moving over a static background, 's' reference -- appended
    # after everything else for the same reason
    dshape = (40, 56)
    for name, (du, dv) in (("block_int", (3.0, -2.0)), ("block_frac", (4.3, 2.6))):
        v = np.zeros(dshape + (2,), np.float32)
        v[12:28, 16:40] = [du, dv]
        f = Flow(v, 's')
        dimg = np.random.default_rng(11).random(dshape + (3,), dtype=np.float32)
        out["disc/" + name + "/img"] = dimg
        w, va = f.apply(dimg, return_valid_area=True)
        put("disc_apply/" + name, f, w)
        out["disc_apply/" + name + "/out_valid"] = va
        put("disc_invert/" + name, f, f.invert())
        put("disc_valid_target/" + name, f, f.valid_target())

    path = os.path.join(HERE, "ref_scipy_paths.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


def main_delaunay(of):
    """Round 2: cases where the cell-wise mesh of the warped grid is NOT the Delaunay triangulation SciPy builds --
    folds (the tiled Sintel field of BASELINE config 5 as loaded), holes of the point mask with random image values,
    curved mesh borders, sheared cells, larger smooth deformations, speckled point masks.  Written to a second file so
    that ref_scipy_paths.npz stays byte-identical."""
    Flow = of.Flow
    out = {}
    rng = np.random.default_rng(29)

    def case(tag, f, img, consider_mask=True, with_invert=True):
        out[tag + "/in_vecs"] = f.vecs
        out[tag + "/in_mask"] = f.mask
        out[tag + "/img"] = img
        w, v = f.apply(img, return_valid_area=True, consider_mask=consider_mask)
        out[tag + "/apply"] = w
        out[tag + "/apply_valid"] = v
        out[tag + "/valid_target"] = f.valid_target(consider_mask=consider_mask)
        if with_invert:
            r = f.invert()
            out[tag + "/invert_vecs"] = r.vecs
            out[tag + "/invert_mask"] = r.mask

    # 1. BASELINE config 5 as loaded: tests/sintel.flo (10 x 20, u = r * c, v = 0) tiled 4 x 4, ref 's' (from_sintel)
    flo = Flow.from_sintel("/root/reference/tests/sintel.flo")
    assert flo.ref == 's'
    tiled = Flow(np.tile(flo.vecs, (4, 4, 1)), 's')
    case("sintel4x4", tiled, rng.random((40, 80, 3), dtype=np.float32))

    shape = (60, 80)
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype('f')
    img = rng.random(shape + (3,), dtype=np.float32)
    base = Flow.from_transforms([['rotation', 40, 30, 14], ['scaling', 20, 20, 0.93]], shape, 's')

    # 2. a 20 x 30 hole in the point mask, random image values
    hole = np.ones(shape, bool)
    hole[20:40, 25:55] = False
    case("hole_img", Flow(base.vecs, 's', hole), img)

    # 3. curved mesh border (radial distortion: the convex hull reaches beyond the warped border in pockets)
    r2 = ((xx - 40) ** 2 + (yy - 30) ** 2) / 2500.0
    barrel = np.stack([(xx - 40) * 0.06 * r2, (yy - 30) * 0.06 * r2], -1).astype('f')
    case("curved", Flow(base.vecs + barrel, 's'), img)
    pincushion = -barrel
    case("curved_in", Flow(base.vecs + pincushion, 's'), img)

    # 4. sheared cells: x += 1.3 y -- grid edges stop being Delaunay everywhere
    shear = np.zeros(shape + (2,), 'f')
    shear[..., 0] = 1.3 * yy - 30
    case("shear", Flow(shear, 's'), img)

    # 5. a larger smooth deformation (3 px sinusoid, near fold-over) on top of a similarity
    wob = np.stack([3.0 * np.sin(2 * np.pi * xx / 47.0) * np.cos(2 * np.pi * yy / 31.0),
                    2.0 * np.cos(2 * np.pi * xx / 37.0) * np.sin(2 * np.pi * yy / 41.0)], -1).astype('f')
    case("wobble3", Flow(base.vecs + wob, 's'), img)

    # 6. speckled point mask (10 % dropped) with random image values, smooth non-affine field
    small = np.stack([0.5 * np.sin(xx / 9) * np.cos(yy / 7), 0.4 * np.cos(xx / 8) * np.sin(yy / 6)], -1).astype('f')
    speck = rng.random(shape) > 0.1
    case("speckle_img", Flow(base.vecs + small, 's', speck), img)

    # 7. anisotropic scaling + shear (affine, but no cell is co-circular: the triangulation is unique)
    aff = np.stack([0.15 * (xx - 30) + 0.2 * (yy - 20), -0.1 * (xx - 30) + 0.25 * (yy - 20)], -1).astype('f')
    case("affine_generic", Flow(aff, 's'), img)
    case("affine_generic_hole", Flow(aff, 's', hole), img)

    # 8. motion boundary with a generic (non-integer, non-axis-aligned) background motion
    blk = (base.vecs + small).copy()
    blk[18:40, 22:58] += np.array([6.4, -3.7], 'f')
    case("block_generic", Flow(blk, 's'), img)

    path = os.path.join(HERE, "ref_delaunay_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
    main_delaunay(_import_reference())
