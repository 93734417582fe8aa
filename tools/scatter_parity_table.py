#!/usr/bin/env python
"""Parity of the scatter path (K3) against OUTPUTS OF THE REAL REFERENCE, case by case (tests/golden/*.npz):
validity-mask mismatches and the share of nodes whose values differ by more than 1e-4 relative, split into nodes
where SciPy's own triangulation is unique and nodes where it is not (co-circular sites / duplicated sites: Qhull's
choice is arbitrary there).  Round 3 adds the composed paths and the other target dtypes: combine_with mode 2 and uint8 images
against the reference's own outputs, mode 1 / invert / switch_ref chains and float64 images against the oracle (the
reference needs cv2.remap for mode 1; the oracle calls the same scipy griddata).
Run on the GPU box:  python tools/scatter_parity_table.py > profiles/r03_scatter_parity.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oflibnumpy_amd as of
from scatter_util import warped_points, nonunique_nodes

RTOL, ATOL = 1e-4, 2e-5


def row(name, path, got, want, gmask, wmask, amb):
    bad = ~np.isclose(got, want, rtol=RTOL, atol=ATOL)
    if bad.ndim == 3:
        bad = bad.any(-1)
    n = bad.size
    mb = gmask ^ wmask
    print("{:<34} {:<8} {:>6} {:>9} {:>9} {:>10.4f} {:>12.4f} {:>14.4f}".format(
        name, path, n, int((mb & ~amb).sum()), int((mb & amb).sum()), 100.0 * amb.mean(), 100.0 * (bad & ~amb).sum() / n,
        100.0 * (bad & amb).sum() / n))


def main():
    of.native.ensure_device()
    from oflibnumpy_amd import device as dev
    print("{:<34} {:<8} {:>6} {:>9} {:>9} {:>10} {:>12} {:>14}".format("case (operation/field)", "path", "nodes", "mask_uniq", "mask_nonu", "nonuniq_%", "bad_unique_%", "bad_nonuniq_%"))
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "ref_delaunay_cases.npz"))
    for tag in sorted({k.split('/')[0] for k in g2.files}):
        vecs, mask, img = g2[tag + '/in_vecs'], g2[tag + '/in_mask'], g2[tag + '/img']
        f = of.Flow(vecs, 's', mask)
        amb, _ = nonunique_nodes(warped_points(vecs, mask), vecs.shape[:2])
        d = f.to_device()
        path = "walk" if (mask.all() and d.mesh_cert(+1).certified) else "delaunay"
        w, v = f.apply(img, return_valid_area=True)
        row("apply(img)/" + tag, path, w, g2[tag + '/apply'], v, g2[tag + '/apply_valid'], amb)
        r = f.invert()
        row("invert/" + tag, path, r.vecs, g2[tag + '/invert_vecs'], r.mask, g2[tag + '/invert_mask'], amb)
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_scipy_paths.npz"))
    img = g['img_f32']
    names = sorted({k.split('/')[1] for k in g.files if k.startswith('apply_img/')})
    for name in names:
        for op, cm in (("apply_img", True), ("apply_img_nomask", False)):
            tag = op + '/' + name
            vecs, mask = g[tag + '/in_vecs'], g[tag + '/in_mask']
            f = of.Flow(vecs, 's', mask)
            keep = mask if (cm and not mask.all()) else None
            amb, _ = nonunique_nodes(warped_points(vecs, keep), vecs.shape[:2])
            d = f.to_device()
            path = "walk" if (keep is None and d.mesh_cert(+1).certified) else "delaunay"
            w, v = f.apply(img, return_valid_area=True, consider_mask=cm)
            row(tag, path, w, g[tag + '/out'], v, g[tag + '/out_valid'], amb)
    for name in ("block_int", "block_frac"):
        tag = "disc_apply/" + name
        vecs = g[tag + '/in_vecs']
        f = of.Flow(vecs, 's', g[tag + '/in_mask'])
        amb, _ = nonunique_nodes(warped_points(vecs), vecs.shape[:2])
        w, v = f.apply(g['disc/' + name + '/img'], return_valid_area=True)
        row(tag, "delaunay", w, g[tag + '/out'], v, g[tag + '/out_valid'], amb)
    # ---- round 3: composed paths and other dtypes
    print()
    print("{:<34} {:<8} {:>6} {:>9} {:>9} {:>10} {:>12} {:>14}".format("composed paths / dtypes", "expected", "nodes", "mask_uniq", "mask_nonu", "nonuniq_%", "bad_unique_%", "bad_nonuniq_%"))
    for fam in ("combine2", "combine2_wobble"):
        for ref in ("s", "t"):
            tag = fam + '/' + ref
            a = of.Flow(g[tag + '/in_vecs'], ref, g[tag + '/in_mask'])
            b = of.Flow(g[tag + '/in2_vecs'], ref, g[tag + '/in2_mask'])
            r = a.combine_with(b, 2)
            shape = a.shape
            if ref == 's':
                amb, _ = nonunique_nodes(warped_points(a.vecs, None if a.mask.all() else a.mask, +1), shape)
            else:           # flow_class.py:1398-1407: float32 points x - f1, queried at x - f3
                yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float32)
                pts = np.stack([(xx - a.vecs[..., 0]).ravel(), (yy - a.vecs[..., 1]).ravel()], 1).astype(np.float64)
                q = np.stack([(xx - b.vecs[..., 0]).ravel(), (yy - b.vecs[..., 1]).ravel()], 1).astype(np.float64)
                amb, _ = nonunique_nodes(pts, shape, queries=q)
            row("mode 2 " + tag, "referen.", r.vecs, g[tag + '/out_vecs'], r.mask, g[tag + '/out_mask'], amb)
    for name in sorted({k.split('/')[1] for k in g.files if k.startswith('apply_u8/')}):
        tag = 'apply_u8/' + name
        vecs, mask = g[tag + '/in_vecs'], g[tag + '/in_mask']
        f = of.Flow(vecs, str(g[tag + '/in_ref']), mask)
        amb, _ = nonunique_nodes(warped_points(vecs, None if mask.all() else mask), vecs.shape[:2])
        got, want = f.apply(g['img_u8']), g[tag + '/out']
        d = np.abs(got.astype(int) - want.astype(int)).max(-1)
        print("{:<34} {:<8} {:>6} {:>9} {:>9} {:>10.4f} {:>12.4f} {:>14.4f}   (uint8: nodes off by one level; none by more: {})".format(
            tag, "referen.", d.size, "-", "-", 100.0 * amb.mean(), 100.0 * ((d > 0) & ~amb).sum() / d.size,
            100.0 * ((d > 0) & amb).sum() / d.size, bool((d <= 1).all() or (d[~amb] <= 1).all())))
    from oracle import np_oracle as O
    for shape, t1, t2 in (((96, 128), [['rotation', 60, 50, -12]], [['scaling', 30, 40, 0.9]]),
                          ((120, 90), [['scaling', 20, 70, 1.08]], [['rotation', 40, 60, 9]])):
        for ref in ("s", "t"):
            f2, f3 = of.Flow.from_transforms(t2, list(shape), ref), of.Flow.from_transforms(t1 + t2, list(shape), ref)
            r = f2.combine_with(f3, 1)
            want = O.OFlow(f2.vecs, ref, f2.mask).combine_with(O.OFlow(f3.vecs, ref, f3.mask), 1)
            amb, _ = nonunique_nodes(warped_points(f2.vecs, None, +1 if ref == 's' else -1), shape)      # first scatter of the chain
            row("mode 1 '{}' {}x{}".format(ref, *shape), "oracle", r.vecs, want.vecs, r.mask, want.mask, amb)
    for ref in ("s", "t"):
        f = of.Flow.from_transforms([['rotation', 256, 256, 30]], [512, 512], ref)
        o = O.OFlow(f.vecs, ref, f.mask)
        amb, _ = nonunique_nodes(warped_points(f.vecs, None, +1 if ref == 's' else -1), (512, 512))
        for nm, r, want in (("invert", f.invert(), o.invert()), ("switch_ref", f.switch_ref(), o.switch_ref())):
            row("{} '{}' 512x512 rotation".format(nm, ref), "oracle", r.vecs, want.vecs, r.mask, want.mask, amb)
    # float64 image (utils.py:253-258 keeps float64): values vs the oracle; the valid area follows the float32 rule
    rng = np.random.default_rng(9)
    shape = (48, 64)
    yy, xx = np.mgrid[:shape[0], :shape[1]].astype(np.float32)
    vecs = of.Flow.from_transforms([['rotation', 30, 20, 11], ['scaling', 10, 10, 0.93]], shape, 's').vecs + \
        np.stack([0.7 * np.sin(xx / 9) * np.cos(yy / 7), 0.5 * np.cos(xx / 8)], -1).astype(np.float32)
    img = rng.random(shape + (3,)) * 1e3
    got, valid = of.Flow(vecs, 's').apply(img, return_valid_area=True)
    want, wvalid = O.OFlow(vecs, 's').apply(img, return_valid_area=True)
    w32valid = O.OFlow(vecs, 's').apply(img.astype(np.float32), return_valid_area=True)[1]
    both = valid & wvalid
    print("float64 image 48x64: values max rel. error {:.2e} on {} nodes; valid area == the float32 rule's: {}; nodes valid here but not under the "
          "reference's float64 `== 1` (its own 1-ulp rounding noise): {:.2f} % (the other way round: {})".format(
              float(np.abs(got[both] - want[both]).max() / 1e3), int(both.sum()), bool(np.array_equal(valid, w32valid)),
              100.0 * (valid & ~wvalid).mean(), int((wvalid & ~valid).sum())))
    print("\nmask_uniq / mask_nonu = nodes whose validity differs from the reference's, outside / inside non-unique simplices (the latter only with\n"
          "speckled mask VALUES, consider_mask=False); nonuniq_% = share of nodes inside a simplex of SciPy's triangulation")
    print("with a fourth site within 1e-9 of its circumcircle (or a duplicated site); bad_* = share of ALL nodes whose value differs by more than")
    print("rtol 1e-4 / atol 2e-5, inside such simplices (Qhull's choice arbitrary) and outside them (triangulation unique).")


if __name__ == "__main__":
    main()
