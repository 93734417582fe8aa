"""np_oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement of the oflibnumpy hot path (Flow.apply / combine_with modes 1-3 /
invert / switch_ref / valid_target / valid_source), used ONLY as the checker by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

Two third-party numerics sit under the reference's hot path:
  * cv2.remap (opencv-python, unpinned in setup.py:52-56, docs pin 4.2.0.34) -- NOT in
    this image; restated in C (oracle/ofl_oracle.c) from OpenCV's published algorithm.
    Sub-1/32-px behaviour is "parity unpinned" (see DESIGN.md).
  * scipy.interpolate.griddata (unpinned, docs pin 1.6.0) -- present in this image and on
    the GPU box, so the scatter restatement below calls the very same SciPy function the
    reference calls (utils.py:253, flow_class.py:1407).

All `file:line` citations are into /root/reference/src/oflibnumpy/.
"""
import ctypes
import os
import subprocess

import numpy as np
from scipy.interpolate import griddata

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libofl_oracle.so")

U8, I16, U16, F32, F64 = 0, 1, 2, 3, 4
QUANT_OPENCV, QUANT_EXACT = 0, 1
ARITH_NATIVE, ARITH_FLOAT_RNE = 0, 1
RULE_EQ1, RULE_GE_HALF, RULE_GT_HALF = 0, 1, 2
DEFAULT_THRESHOLD = 1e-3  # utils.py:22

_DT = {np.dtype('uint8'): U8, np.dtype('int16'): I16, np.dtype('uint16'): U16,
       np.dtype('float32'): F32, np.dtype('float64'): F64}


def build(force=False):
    """Compile oracle/ofl_oracle.c with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "ofl_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libofl_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, ci, cs = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        L.orc_gather_bilinear.argtypes = [vp, ci, ci, ci, ci, vp, ci, ci, ci, ci, ci, vp, vp, vp, ci, ci, ci]
        L.orc_gather_bilinear.restype = ci
        L.orc_compose3.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, ci]
        L.orc_compose3.restype = ci
        L.orc_is_zero.argtypes = [vp, vp, cs, ci, ctypes.c_double]
        L.orc_is_zero.restype = ci
        L.orc_resize_flow.argtypes = [vp, vp, ci, ci, ci, ci, ctypes.c_double, ctypes.c_double,
                                      ctypes.c_float, ctypes.c_float, vp, vp]
        L.orc_resize_flow.restype = ci
        L.orc_set_threads.argtypes = [ci]
        L.orc_set_threads.restype = ci
        _lib = L
    return _lib


def set_threads(n):
    return lib().orc_set_threads(int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _mask_u8(m):
    return None if m is None else np.ascontiguousarray(m).astype(np.uint8)


# --------------------------------------------------------------------------- L0: remap
def gather_bilinear(src, flow, sign, smask=None, want_valid=False, quant=QUANT_OPENCV,
                    arith=ARITH_NATIVE, rule=RULE_EQ1, pad=(0, 0)):
    """dst[y,x] = B(src; (x,y) + sign*flow[y,x]); restates utils.py:231-236 + cv2.remap."""
    src = np.ascontiguousarray(src)
    squeeze = src.ndim == 2
    s3 = src[..., None] if squeeze else src
    s3 = np.ascontiguousarray(s3)
    if s3.dtype not in _DT:
        raise TypeError("oracle remap: unsupported dtype {}".format(s3.dtype))
    H, W, C = s3.shape
    flow = np.ascontiguousarray(flow, dtype=np.float32)
    fH, fW = flow.shape[:2]
    dst = np.empty_like(s3)
    sm = _mask_u8(smask)
    valid = np.empty((H, W), np.uint8) if want_valid else None
    rc = lib().orc_gather_bilinear(_p(s3), _DT[s3.dtype], C, H, W, _p(flow), fH, fW, pad[0], pad[1],
                                   int(sign), _p(sm), _p(dst), _p(valid), quant, arith, rule)
    assert rc == 0
    if squeeze:
        dst = dst[..., 0]
    return (dst, valid.astype(bool)) if want_valid else dst


def compose3_raw(fa, ma, fb, mb, sign, quant=QUANT_OPENCV):
    """out = fb + B(fa; x + sign*fb), mout = mb & [B(ma) == 1]  (flow_class.py:1418,1422)."""
    fa = np.ascontiguousarray(fa, np.float32)
    fb = np.ascontiguousarray(fb, np.float32)
    H, W = fa.shape[:2]
    out = np.empty((H, W, 2), np.float32)
    mout = np.empty((H, W), np.uint8)
    rc = lib().orc_compose3(_p(fa), _p(_mask_u8(ma)), _p(fb), _p(_mask_u8(mb)), int(sign), H, W,
                            _p(out), _p(mout), quant)
    assert rc == 0
    return out, mout.astype(bool)


def is_zero_raw(vecs, mask=None, thresholded=True):
    vecs = np.ascontiguousarray(vecs, np.float32)
    m = _mask_u8(mask)
    return bool(lib().orc_is_zero(_p(vecs), _p(m), vecs.size // 2, int(bool(thresholded)),
                                  DEFAULT_THRESHOLD))


def resize_flow(flow, scale, mask=None):
    """utils.py:493-525 (+ the mask of flow_class.py:504-506 when `mask` is given): cv2.resize(INTER_LINEAR)."""
    scale = [scale, scale] if isinstance(scale, (int, float)) else list(scale)
    flow = np.ascontiguousarray(flow, np.float32)
    H, W = flow.shape[:2]
    Ho, Wo = int(np.rint(H * scale[0])), int(np.rint(W * scale[1]))        # saturate_cast<int>(double) = cvRound
    out = np.empty((Ho, Wo, 2), np.float32)
    m = _mask_u8(mask)
    mout = np.empty((Ho, Wo), np.uint8) if mask is not None else None
    rc = lib().orc_resize_flow(_p(flow), _p(m), H, W, Ho, Wo, 1.0 / scale[0], 1.0 / scale[1],
                               float(np.float32(scale[1])), float(np.float32(scale[0])), _p(out), _p(mout))
    assert rc == 0
    return (out, mout.astype(bool)) if mask is not None else out


# ----------------------------------------------------------------- L1: apply_flow restated
def threshold_vectors(vecs, threshold=None):
    """utils.py:298-316 (use_mag=False branch)."""
    threshold = DEFAULT_THRESHOLD if threshold is None else threshold
    f = vecs.copy()
    f[(vecs < threshold) & (vecs > -threshold)] = 0
    return f


def is_zero_flow(flow, thresholded=True):
    """utils.py:527-544."""
    f = threshold_vectors(flow) if thresholded else flow
    return bool(np.all(f == 0))


def scatter_griddata(flow, target, mask=None):
    """'s' branch of apply_flow, utils.py:237-258: scattered points (row, col) + flow[::-1],
    optional mask filtering, SciPy Delaunay-linear interpolation, NaN -> 0, integer rounding."""
    field = flow.astype('float32')
    h, w = field.shape[:2]
    r, c = np.mgrid[:h, :w]
    pos = np.stack([r.ravel(), c.ravel()], axis=1) + field[..., ::-1].reshape(-1, 2)
    vals = target.reshape(-1, target.shape[-1]) if target.ndim == 3 else target.ravel()
    if mask is not None:
        keep = mask.ravel()
        pos, vals = pos[keep], vals[keep]
    res = np.nan_to_num(griddata(pos, vals, (r, c), method='linear'))
    if np.issubdtype(target.dtype, np.integer):
        res = np.round(res)
    return res.astype(target.dtype)


def apply_flow(flow, target, ref, mask=None, quant=QUANT_OPENCV, arith=ARITH_NATIVE):
    """utils.py:199-261 (validation omitted: the oracle is fed valid inputs)."""
    flow = flow.astype('float32')
    if is_zero_flow(flow, True):
        return target
    if ref == 't':
        res = gather_bilinear(target, flow, -1, quant=quant, arith=arith)
    else:
        res = scatter_griddata(flow, target, mask)
    if res.shape != target.shape:
        res = res[:, :, None]
    return res


# --------------------------------------------------------------- L2: Flow algebra restated
class OFlow:
    """Minimal (vecs, mask, ref) triple with the reference's hot-path algebra."""

    def __init__(self, vecs, ref='t', mask=None):
        self.vecs = np.asarray(vecs).astype('float32')          # flow_class.py:80
        self.ref = 't' if ref is None else ref
        self.mask = np.ones(self.vecs.shape[:2], bool) if mask is None else np.asarray(mask).astype(bool)

    shape = property(lambda self: self.vecs.shape[:2])

    # flow_class.py:310-375, 479-489
    def __add__(self, o):
        return OFlow(self.vecs + o.vecs, self.ref, self.mask & o.mask)

    def __sub__(self, o):
        return OFlow(self.vecs - o.vecs, self.ref, self.mask & o.mask)

    def __neg__(self):
        return OFlow(self.vecs * float(-1), self.ref, self.mask)

    def is_zero(self, thresholded=True, masked=True):
        """flow_class.py:1230-1245."""
        f = self.vecs[self.mask][None] if masked else self.vecs
        return is_zero_flow(f, thresholded)

    def resize(self, scale):
        """flow_class.py:491-506."""
        v, m = resize_flow(self.vecs, scale, self.mask)
        return OFlow(v, self.ref, m)

    def pad(self, padding, mode='constant'):
        """flow_class.py:508-526."""
        p = padding
        vecs = np.pad(self.vecs, ((p[0], p[1]), (p[2], p[3]), (0, 0)), mode=mode)
        mask = np.pad(self.mask, ((p[0], p[1]), (p[2], p[3])))
        return OFlow(vecs, self.ref, mask)

    def apply(self, target, target_mask=None, return_valid_area=False, consider_mask=True,
              quant=QUANT_OPENCV, padding=None, cut=True):
        """flow_class.py:528-695."""
        sh = self.shape
        if isinstance(target, OFlow):
            return_flow, t, mask = True, target.vecs, target.mask
        else:
            return_flow = False
            t = target if target.ndim == 3 else target[..., None]
            mask = np.ones(t.shape[:2], 'b') if target_mask is None else target_mask.copy()   # int8! (:615)
        with_mask = return_flow or return_valid_area
        if with_mask:
            if self.ref == 's':
                if mask.shape != sh:                                                     # :636-641
                    tmp = mask[padding[0]:padding[0] + sh[0], padding[2]:padding[2] + sh[1]].copy()
                    mask = mask.copy()
                    mask[...] = False
                    mask[padding[0]:padding[0] + sh[0], padding[2]:padding[2] + sh[1]] = tmp & self.mask
                else:
                    mask = mask & self.mask                                              # :643
            t = np.concatenate((t, mask[..., None]), axis=-1)                            # :644
        if padding is None:
            warped = apply_flow(self.vecs, t, self.ref, self.mask if consider_mask else None, quant)
        else:
            flow = self.pad(padding, 'constant' if self.ref == 't' else 'edge')          # :652-660
            warped = apply_flow(flow.vecs, t, flow.ref, flow.mask if consider_mask else None, quant)
        if padding is not None and cut:                                                  # :663-664
            warped = warped[padding[0]:padding[0] + sh[0], padding[2]:padding[2] + sh[1]]
        if with_mask:
            valid = warped[..., -1] == 1                                                 # :668
            if self.ref == 't':
                if valid.shape != self.mask.shape:                                       # :673-678
                    tmp = valid[padding[0]:padding[0] + sh[0], padding[2]:padding[2] + sh[1]].copy()
                    valid[...] = False
                    valid[padding[0]:padding[0] + sh[0], padding[2]:padding[2] + sh[1]] = tmp & self.mask
                else:
                    valid = valid & self.mask                                            # :680
        if return_flow:
            return OFlow(warped[:, :, :2], target.ref, valid)                            # :684
        if return_valid_area:
            warped = warped[:, :, :-1]
        if np.issubdtype(target.dtype, np.integer):
            warped = np.round(warped)
        if target.ndim == 2:
            warped = warped[:, :, 0]
        warped = warped.astype(target.dtype)
        return (warped, valid) if return_valid_area else warped

    def switch_ref(self):
        """flow_class.py:697-733, mode 'valid'."""
        other = 't' if self.ref == 's' else 's'
        if self.is_zero(thresholded=False):
            return OFlow(self.vecs, other, self.mask)
        if self.ref == 's':
            out = self.apply(self)
            out.ref = 't'
            return out
        as_s = OFlow(self.vecs, 's', self.mask)
        return (-as_s).apply(as_s)

    def invert(self, ref=None):
        """flow_class.py:735-753."""
        ref = self.ref if ref is None else ref
        if self.ref == 's':
            return self.apply(-self) if ref == 's' else OFlow(-self.vecs, 't', self.mask)
        if ref == 's':
            return OFlow(-self.vecs, 's', self.mask)
        return self.invert('s').switch_ref()

    def valid_target(self, consider_mask=True):
        """flow_class.py:1113-1151."""
        if self.ref == 's':
            area = apply_flow(self.vecs, self.mask.astype('f'), 's', self.mask if consider_mask else None)
            return area == 1
        area = apply_flow(self.vecs, np.ones(self.shape), 't') == 1
        return area & self.mask

    def valid_source(self, consider_mask=True):
        """flow_class.py:1153-1195."""
        if self.ref == 's':
            area = apply_flow(-self.vecs, np.ones(self.shape), 't') == 1
            return area & self.mask
        area = apply_flow(-self.vecs, self.mask.astype('f'), 's', self.mask if consider_mask else None)
        return area == 1

    def combine_with(self, flow, mode, thresholded=False):
        """flow_class.py:1247-1424."""
        if self.is_zero(thresholded=thresholded):
            return flow
        if flow.is_zero(thresholded=thresholded):
            return self if mode == 3 else self.invert()
        s = self.ref == 's'
        if mode == 1:
            if s:                                                                        # :1369-1370
                g = flow.invert('t')
                return flow - (g + g.apply(self.switch_ref())).apply(self)
            a = self.switch_ref()                                                        # :1383-1385
            res = flow.switch_ref() - (a + a.invert(ref='t').apply(flow.invert('s'))).apply(a)
            return res.switch_ref()
        if mode == 2:
            if s:
                return self.apply(flow - self)                                           # :1390
            return flow - _mode2_t_resample(self, flow)                                  # :1398-1410
        if s:
            return self + self.invert(ref='t').apply(flow)                               # :1418
        return flow + flow.apply(self)                                                   # :1422


def _mode2_t_resample(f1, f3):
    """flow_class.py:1398-1410: f1 resampled (Delaunay-linear) from points x - f1 to points x - f3."""
    h, w = f1.shape
    c1 = np.copy(-f1.vecs)
    c1[:, :, 0] += np.arange(w)
    c1[:, :, 1] += np.arange(h)[:, None]
    vals = np.concatenate((f1.vecs, f1.mask[..., None]), axis=-1).reshape(-1, 3)
    c3 = np.copy(-f3.vecs)
    c3[:, :, 0] += np.arange(w)
    c3[:, :, 1] += np.arange(h)[:, None]
    r = griddata(c1.reshape(-1, 2), vals, (c3[..., 0], c3[..., 1]), method='linear', fill_value=0)
    return OFlow(r[..., :-1], 't', r[..., -1] > .99)


# ------------------------------------------------- input generators (analytic affine oracle)
def matrix_from_transforms(transform_list):
    """utils.py:114-158: 3x3 matrix of a list of translation / rotation / scaling transforms."""
    m = np.identity(3)
    for name, *v in reversed(transform_list):
        t = np.identity(3)
        if name == 'translation':
            t[0:2, 2] = v[0], v[1]
        else:
            pre, post = np.identity(3), np.identity(3)
            pre[0:2, 2] = -v[0], -v[1]
            post[0:2, 2] = v[0], v[1]
            if name == 'scaling':
                t[0, 0] = t[1, 1] = v[2]
            elif name == 'rotation':
                a = np.radians(v[2])
                t[0:2, 0:2] = [[np.cos(a), np.sin(a)], [-np.sin(a), np.cos(a)]]
            else:
                raise ValueError(name)
            t = post @ t @ pre
        m = m @ t
    return m


def flow_from_matrix(matrix, shape, ref):
    """utils.py:91-111, 319-344."""
    h, w = shape
    if ref == 't':
        matrix = np.linalg.pinv(matrix)
    hom = np.zeros((h, w, 3), 'f')
    hom[..., 0] += np.arange(w)
    hom[..., 1] += np.arange(h)[:, None]
    hom[..., 2] = 1
    tr = np.squeeze(np.matmul(matrix, hom[..., None]))
    vec = np.array(tr[..., 0:2] / tr[..., 2, None] - hom[..., 0:2], 'float32')
    return -vec if ref == 't' else vec


def from_transforms(transform_list, shape, ref, mask=None):
    return OFlow(flow_from_matrix(matrix_from_transforms(transform_list), shape, ref), ref, mask)


# ------------------------------------------------------------------ sparse point tracking (next-tier row)
def bilinear_interpolation(data, pts):
    """utils.py:161-196, restated INCLUDING its pairing of the (ver1, hor0) sample with the (ver0, hor1)
    weight and vice versa (b / c below) -- results must match the reference, not the textbook."""
    ver, hor = pts[:, 0], pts[:, 1]
    h, w = data.shape[:2]
    if any(~((0 <= ver) & (ver <= h - 1)) | ~((0 <= hor) & (hor <= w - 1))):
        raise IndexError("Some points are outside of the data area.")
    v0, h0 = np.floor(ver).astype(int), np.floor(hor).astype(int)
    v0c, h0c = np.clip(v0, 0, h - 1), np.clip(h0, 0, w - 1)
    v1c, h1c = np.clip(v0 + 1, 0, h - 1), np.clip(h0 + 1, 0, w - 1)
    w_a = (v1c - ver) * (h1c - hor)
    w_b = (v1c - ver) * (hor - h0c)
    w_c = (ver - v0c) * (h1c - hor)
    w_d = (ver - v0c) * (hor - h0c)
    return (w_a[:, None] * data[v0c, h0c] + w_b[:, None] * data[v1c, h0c] +
            w_c[:, None] * data[v0c, h1c] + w_d[:, None] * data[v1c, h1c])


def track_pts(flow, ref, pts, int_out=False, s_exact_mode=False):
    """utils.py:547-622 (validation omitted)."""
    flow = flow.astype('float32')
    if is_zero_flow(flow, True):
        warped = pts
    else:
        h, w = flow.shape[:2]
        r, c = np.mgrid[:h, :w]
        grid = np.stack([r.ravel(), c.ravel()], axis=1)
        flat = flow[..., ::-1].reshape(-1, 2)
        if ref == 's':
            if np.issubdtype(pts.dtype, np.integer):
                vecs = flow[pts[:, 0], pts[:, 1], ::-1]
            elif s_exact_mode:
                vecs = griddata(grid, flat, (pts[:, 0], pts[:, 1]), method='linear')
            else:
                vecs = bilinear_interpolation(flow[..., ::-1], pts)
        else:
            vecs = griddata(grid - flat, flat, (pts[:, 0], pts[:, 1]), method='linear')
        warped = pts + vecs
        nan = np.isnan(warped)
        warped[nan[:, 0] | nan[:, 1]] = 0
    if int_out:
        warped = np.round(warped).astype('i')
    return warped
