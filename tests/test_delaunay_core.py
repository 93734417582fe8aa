"""CPU suite, part 4: the geometry core of the exact scatter path (oflibnumpy_amd/csrc/ofl_delaunay_core.h, the very
header the HIP kernels include) compiled for the host and checked against scipy.spatial.Delaunay -- what
scipy.interpolate.griddata (src/oflibnumpy/utils.py:253) triangulates with -- on the reference fixtures: every
simplex SciPy builds that is UNIQUELY Delaunay (no other site within rounding of its circumcircle) must be found
by the stars, and every triangle the stars emit must have an empty circumcircle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from scipy.spatial import Delaunay, cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core(tmp_path_factory):
    out = tmp_path_factory.mktemp("dlcore") / "libdlcore.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", str(out),
                           os.path.join(ROOT, "tests", "native", "dl_core_cpu.cpp")])
    lib = ctypes.CDLL(str(out))
    lib.dl_stars_cpu.restype = ctypes.c_int
    return lib


def stars(lib, pts, rings=6, near_cap=16):
    pts = np.ascontiguousarray(pts, np.float64)
    n = len(pts)
    cap = 12 * n + 1024
    tri = np.zeros((cap, 3), np.int32)
    nt = ctypes.c_longlong(0)
    info = (ctypes.c_int * 8)()
    rc = lib.dl_stars_cpu(pts.ctypes.data_as(ctypes.c_void_p), n, rings, near_cap, tri.ctypes.data_as(ctypes.c_void_p),
                          ctypes.c_longlong(cap), ctypes.byref(nt), info)
    assert rc == 0 and nt.value <= cap
    return tri[:nt.value], list(info)


def stars_grid(lib, pts_all, kept, h, w, rings=6, near_cap=-12):
    """all H * W warped positions + the kept mask: the mesh-fan shortcut first, then the clip path (as the GPU does)"""
    pts_all = np.ascontiguousarray(pts_all, np.float64)
    kept = np.ascontiguousarray(kept, np.uint8)
    cap = 12 * h * w + 1024
    tri = np.zeros((cap, 3), np.int32)
    nt = ctypes.c_longlong(0)
    info = (ctypes.c_int * 8)()
    lib.dl_stars_grid_cpu.restype = ctypes.c_int
    rc = lib.dl_stars_grid_cpu(pts_all.ctypes.data_as(ctypes.c_void_p), kept.ctypes.data_as(ctypes.c_void_p), h, w, rings, near_cap,
                               tri.ctypes.data_as(ctypes.c_void_p), ctypes.c_longlong(cap), ctypes.byref(nt), info)
    assert rc == 0 and nt.value <= cap
    return tri[:nt.value], list(info)


def circum(pts, tri):
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    bx, by, cx, cy = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1], c[:, 0] - a[:, 0], c[:, 1] - a[:, 1]
    d = 2 * (bx * cy - by * cx)
    with np.errstate(divide='ignore', invalid='ignore'):
        ux = (cy * (bx * bx + by * by) - by * (cx * cx + cy * cy)) / d
        uy = (bx * (cx * cx + cy * cy) - cx * (bx * bx + by * by)) / d
    return np.stack([a[:, 0] + ux, a[:, 1] + uy], 1), np.hypot(ux, uy), d


def unique_simplices(pts, simplices, tol=1e-9):
    """simplices with no FOURTH site within tol (relative) of their circumcircle"""
    cen, r, d = circum(pts, simplices)
    tree = cKDTree(pts)
    ok = np.zeros(len(simplices), bool)
    for i, (c, rad) in enumerate(zip(cen, r)):
        if not np.isfinite(rad) or rad > 1e6:
            continue
        near = tree.query_ball_point(c, rad * (1 + 1e-6) + 1e-9)
        others = [j for j in near if j not in simplices[i]]
        ok[i] = all(abs(np.hypot(*(pts[j] - c)) - rad) > tol * max(rad, 1.0) for j in others)
    return ok


def fixture_points(g, tag):
    vecs, mask = g[tag + '/in_vecs'], g[tag + '/in_mask']
    h, w = vecs.shape[:2]
    yy, xx = np.mgrid[:h, :w]
    p = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)
    return p[mask.ravel()]


def fixture_grid(g, tag):
    vecs, mask = g[tag + '/in_vecs'], g[tag + '/in_mask']
    h, w = vecs.shape[:2]
    yy, xx = np.mgrid[:h, :w]
    p = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)
    return p, mask.ravel(), h, w


@pytest.mark.parametrize("tag", ["curved", "curved_in", "shear", "wobble3", "speckle_img", "affine_generic",
                                 "affine_generic_hole", "block_generic", "hole_img"])
@pytest.mark.parametrize("near_cap", [16, -12, "fan"])   # float64 cells of 16 vertices / the GPU's float32 cells of 12 / mesh fans first
def test_stars_equal_scipy_delaunay(core, golden2, tag, near_cap):
    pts = fixture_points(golden2, tag)
    if near_cap == "fan":
        pall, kept, h, w = fixture_grid(golden2, tag)
        tri, info = stars_grid(core, pall, kept, h, w)
        # the shortcuts are taken where the mesh is intact (sheared cells: nowhere): whole neighbourhoods by the cell pass, the rest by fans
        assert info[5] + info[7] > (-1 if tag == "shear" else 0.3 * kept.sum()), (tag, info)
        assert info[7] > (-1 if tag in ("shear", "speckle_img") else 0.25 * kept.sum()), (tag, info)
        if tag in ("affine_generic_hole", "hole_img", "block_generic"):
            assert info[6] > 20, (tag, info)                    # rims of the hole / of the tear close in the second per-thread pass
        remap = np.cumsum(kept) - 1                             # grid index -> index among the kept points
        tri = remap[tri].astype(np.int32)
    else:
        tri, info = stars(core, pts, near_cap=near_cap)
    assert info[1] == 0 or info[0] > 0                      # overflowing stars went to the far pass
    ours = {tuple(sorted(t)) for t in tri.tolist()}
    d = Delaunay(pts)
    uniq = unique_simplices(pts, d.simplices)
    assert uniq.mean() > (0.5 if tag == "hole_img" else 0.9)          # hole_img: a similarity, ~13 % co-circular cells
    missing = [tuple(sorted(s)) for s, u in zip(d.simplices.tolist(), uniq) if u and tuple(sorted(s)) not in ours]
    assert not missing, (tag, len(missing), missing[:5])
    # nothing but Delaunay triangles: no site strictly inside a circumcircle (beyond rounding)
    t = np.array(sorted(ours))
    cen, r, det = circum(pts, t)
    tree = cKDTree(pts)
    good = np.isfinite(r) & (r < 1e6)
    dist, _ = tree.query(cen[good], k=4)
    assert (dist[:, 3] >= r[good] * (1 - 1e-7) - 1e-9).all(), tag
    # every star is emitted from each of its sites: a unique simplex appears three times
    cnt = {}
    for tt in tri.tolist():
        cnt[tuple(sorted(tt))] = cnt.get(tuple(sorted(tt)), 0) + 1
    assert all(cnt[tuple(sorted(s))] == 3 for s, u in zip(d.simplices.tolist(), uniq) if u)


@pytest.mark.parametrize("with_fan", [False, True])
def test_stars_on_folds_and_duplicates(core, golden2, with_fan):
    """BASELINE config 5 as loaded: the tiled Sintel field puts every site on the integer lattice, many of them twice.
    The stars still cover SciPy's triangulation wherever it is unique, and their triangles are all Delaunay."""
    pts = fixture_points(golden2, "sintel4x4")
    pall, kept, h, w = fixture_grid(golden2, "sintel4x4")
    assert kept.all()
    if with_fan:
        tri, info = stars_grid(core, pall, kept, h, w)
        assert info[5] >= 0, info          # (u = x * y: every row has its own spacing, no cell of this mesh is Delaunay)
    else:
        tri, info = stars(core, pts)
    ours = {tuple(sorted(t)) for t in tri.tolist()}
    upts, inv = np.unique(pts, axis=0, return_inverse=True)
    d = Delaunay(upts)
    uniq = unique_simplices(upts, d.simplices)
    ours_u = {tuple(sorted(int(inv[i]) for i in t)) for t in ours}
    missing = [tuple(sorted(s)) for s, u in zip(d.simplices.tolist(), uniq) if u and tuple(sorted(s)) not in ours_u]
    assert not missing, (len(missing), missing[:5])


def test_far_pass_handles_unbounded_and_large_cells(core):
    """Random sites with a large empty disc: hull sites (unbounded cells) and the rim of the disc (cells larger than
    the ring search) go through the far pass; the result is SciPy's triangulation."""
    rng = np.random.default_rng(3)
    pts = rng.random((3000, 2)) * [120, 90]
    pts = pts[np.hypot(pts[:, 0] - 60, pts[:, 1] - 45) > 25]
    tri, info = stars(core, pts, rings=3)
    assert info[0] > 50
    ours = {tuple(sorted(t)) for t in tri.tolist()}
    ref = {tuple(sorted(s)) for s in Delaunay(pts).simplices.tolist()}
    assert ref <= ours and len(ours - ref) == 0


def test_tie_on_coincident_vertices_keeps_the_real_run(core):
    """Soak seed 1000020 (tests/scatter_soak_util.py): a similarity-transformed lattice under a random point mask.  The
    bisector of a neighbour passes exactly through two coincident vertices of the growing cell (four co-circular sites); the
    tie rule flags one of them and not the other, and the clip must still remove the run of vertices the half-plane really
    cuts off -- it used to remove the one-vertex run of the tie, the neighbour was never applied, and the star of site 6936
    lacked its south-east neighbour 7020 (the triangle SciPy has there: 6936, 6937, 7020)."""
    from scatter_soak_util import make_case
    h, w, kind, vecs, pm, sign, C, vals, vm = make_case(1000020, 160, 240)
    assert (h, w, kind, sign) == (125, 83, 4, 1) and pm is not None
    yy, xx = np.mgrid[:h, :w]
    pts_all = np.stack([(xx + vecs[..., 0].astype(np.float64)).ravel(), (yy + vecs[..., 1].astype(np.float64)).ravel()], 1)
    tri, info = stars_grid(core, pts_all, pm, h, w)
    star = [tuple(int(v) for v in t[1:]) for t in tri if t[0] == 6936]
    assert (6937, 7020) in star and (7020, 7102) in star, star
    idx = np.flatnonzero(pm.ravel())
    d = Delaunay(pts_all[idx])
    want = {tuple(sorted(int(idx[v]) for v in t)) for t, u in zip(d.simplices, unique_simplices(pts_all[idx], d.simplices)) if u}
    got = {tuple(sorted(int(v) for v in t)) for t in tri}
    assert not (want - got), sorted(want - got)[:5]


def test_exactly_cocircular_sites_are_a_tie_for_every_star(core):
    """Soak seed 3000265: a similarity field whose float32 rounding leaves FIVE sites exactly co-circular (in the rationals) and
    a sixth 1e-5 off.  The in-circle determinant of four of them, evaluated relative to the asking site, is + 3.6e-15 from one
    site and - 0.0 from another; unfiltered, one star inserted the candidate at the wrong end of a run of coincident vertices
    (its neighbours were no longer in angular order) and the triangles of four stars overlapped on a unique simplex.  With the
    filter (ofl_dl::incircle_origin_filtered) everybody calls it a tie and the index rule decides."""
    from scatter_soak_util import make_case
    h, w, kind, vecs, pm, sign, C, vals, vm = make_case(3000265, 320, 420)
    assert (h, w, kind, sign) == (317, 166, 4, -1) and pm is not None
    yy, xx = np.mgrid[:h, :w]
    pts_all = np.stack([(xx + sign * vecs[..., 0].astype(np.float64)).ravel(), (yy + sign * vecs[..., 1].astype(np.float64)).ravel()], 1)
    tri, info = stars_grid(core, pts_all, pm, h, w)
    star = [int(t[1]) for t in tri if t[0] == 20407]
    ang = np.unwrap(np.arctan2(*(pts_all[star] - pts_all[20407]).T[::-1]))
    assert (np.diff(ang) > -1e-9).all() or (np.diff(ang) < 1e-9).all(), star       # neighbours in angular order
    idx = np.flatnonzero(pm.ravel())
    d = Delaunay(pts_all[idx])
    want = {tuple(sorted(int(idx[v]) for v in t)) for t, u in zip(d.simplices, unique_simplices(pts_all[idx], d.simplices, 1e-8)) if u}
    got = {tuple(sorted(int(v) for v in t)) for t in tri}
    assert (20407, 20410, 20574) in want and not (want - got), sorted(want - got)[:5]

