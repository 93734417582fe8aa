#!/usr/bin/env python
"""Parity of the scatter path (K3) against OUTPUTS OF THE REAL REFERENCE, case by case (tests/golden/*.npz):
validity-mask mismatches and the share of nodes whose values differ by more than 1e-4 relative, split into nodes
where SciPy's own triangulation is unique and nodes where it is not (co-circular sites / duplicated sites: Qhull's
choice is arbitrary there).  Run on the GPU box:  python tools/scatter_parity_table.py > profiles/r02_scatter_parity.txt
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
    print("\nmask_uniq / mask_nonu = nodes whose validity differs from the reference's, outside / inside non-unique simplices (the latter only with\n"
          "speckled mask VALUES, consider_mask=False); nonuniq_% = share of nodes inside a simplex of SciPy's triangulation")
    print("with a fourth site within 1e-9 of its circumcircle (or a duplicated site); bad_* = share of ALL nodes whose value differs by more than")
    print("rtol 1e-4 / atol 2e-5, inside such simplices (Qhull's choice arbitrary) and outside them (triangulation unique).")


if __name__ == "__main__":
    main()
