"""Helpers shared by the GPU scatter tests: where is SciPy's own triangulation unique?"""
import numpy as np


def warped_points(vecs, keep=None, sign=1):
    h, w = vecs.shape[:2]
    yy, xx = np.mgrid[:h, :w]
    p = np.stack([(xx + sign * vecs[..., 0].astype(np.float64)).ravel(), (yy + sign * vecs[..., 1].astype(np.float64)).ravel()], 1)
    return p if keep is None else p[np.asarray(keep, bool).ravel()]


def nonunique_nodes(points, shape, queries=None, tol=1e-9):
    """Grid nodes whose covering simplex of SciPy's own triangulation is NOT uniquely Delaunay (a fourth site within
    `tol` (relative) of its circumcircle, or a duplicated site): Qhull's choice among the co-circular alternatives is arbitrary
    there, and non-affine data (image values, speckled masks) can tell the alternatives apart.  Everywhere else the
    Delaunay triangulation -- and with it griddata's result -- is unique.  `queries` (N x 2, (x, y), N = H * W): scattered
    query positions instead of the grid nodes (mode 2 / 't', flow_class.py:1407).  Returns (ambiguous, inside_hull).
    (Qhull merges facets that are coplanar within ITS roundoff, which grows with the coordinates: on a 248 x 411
    similarity field it splits a cell whose fourth corner is 1.1e-9 px OUTSIDE the circle along the other diagonal -- and
    along the right one when x and y are swapped; tools/soak_scatter.py therefore scales `tol` with the field.)"""
    from scipy.spatial import Delaunay
    from test_delaunay_core import unique_simplices
    upts, inv, counts = np.unique(points, axis=0, return_inverse=True, return_counts=True)
    d = Delaunay(upts)
    uniq = unique_simplices(upts, d.simplices, tol)
    dup_vertex = (counts[d.simplices] > 1).any(1)
    yy, xx = np.mgrid[:shape[0], :shape[1]]
    q = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.float64) if queries is None else np.asarray(queries, np.float64)
    s = d.find_simplex(q).reshape(shape)
    amb = np.zeros(shape, bool)
    inside = s >= 0
    amb[inside] = ~uniq[s[inside]] | dup_vertex[s[inside]]
    return amb, inside


def ambiguous_for(vecs, keep=None, sign=1):
    return nonunique_nodes(warped_points(vecs, keep, sign), vecs.shape[:2])[0]


def hull_band(points, shape, queries=None, width=1e-5):
    """Grid nodes (or query positions) within `width` px of the border of the convex hull of `points`: whether such a
    position is inside is decided by Qhull's handling of the sliver facets along a border that is straight only up to
    float rounding (find_simplex fails on positions 4e-7 px INSIDE the hull of float32-rounded points) -- rounding
    noise of the reference, not a rule."""
    from scipy.spatial import ConvexHull
    hull = ConvexHull(np.unique(points, axis=0))
    yy, xx = np.mgrid[:shape[0], :shape[1]]
    q = np.stack([xx.ravel(), yy.ravel()], 1).astype(np.float64) if queries is None else np.asarray(queries, np.float64)
    d = (q @ hull.equations[:, :2].T + hull.equations[:, 2]).max(1)          # signed distance to the nearest facet, < 0 inside
    return (np.abs(d) < width).reshape(shape)
