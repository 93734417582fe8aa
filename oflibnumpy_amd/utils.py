"""Array-level functions of the engine: the drop-in seam `apply_flow` plus the input validators and
flow generators the `Flow` class builds on.

Mirrors the public names, argument meaning and exception types of the reference's
src/oflibnumpy/utils.py; the numerics of the warp itself run in hand-written HIP kernels
(csrc/ofl_gather.hip, csrc/ofl_scatter.hip) reached through the C ABI of include/ofl.h.
"""
import math
from typing import Any, Union

import numpy as np

from . import _native as nat
from . import device as dev

nd = np.ndarray
DEFAULT_THRESHOLD = 1e-3           # reference utils.py:22

_REMAP_DTYPES = (np.uint8, np.int16, np.uint16, np.float32, np.float64)


# --------------------------------------------------------------------------- validators
def get_valid_ref(ref: Any) -> str:
    """None -> 't'; otherwise must be the string 's' or 't' (reference utils.py:25-39)."""
    if ref is None:
        return 't'
    if not isinstance(ref, str):
        raise TypeError("Error setting flow reference: Input is not a string")
    if ref not in ('s', 't'):
        raise ValueError("Error setting flow reference: Input is not 's' or 't', but {}".format(ref))
    return ref


def get_valid_padding(padding: Any, error_string: str = None) -> list:
    """[top, bot, left, right] of non-negative ints (reference utils.py:42-59)."""
    pre = error_string or ''
    if not isinstance(padding, (list, tuple)):
        raise TypeError(pre + "Padding needs to be a tuple or a list list of values [top, bot, left, right]")
    if len(padding) != 4:
        raise ValueError(pre + "Padding list needs to be a list or tuple of length 4 [top, bot, left, right]")
    if not all(isinstance(p, int) for p in padding):
        raise ValueError(pre + "Padding list [top, bot, left, right] items need to be integers")
    if not all(p >= 0 for p in padding):
        raise ValueError(pre + "Padding list [top, bot, left, right] items need to be 0 or larger")
    return padding


def validate_shape(shape: Any):
    """(H, W) of positive ints (reference utils.py:62-68)."""
    if not isinstance(shape, (list, tuple)):
        raise TypeError("Error creating flow from matrix: Dims need to be a list or a tuple")
    if len(shape) != 2:
        raise ValueError("Error creating flow from matrix: Dims need to be a list or a tuple of length 2")
    if any((not isinstance(s, int) or s <= 0) for s in shape):
        raise ValueError("Error creating flow from matrix: Dims need to be a list or a tuple of integers above zero")


def validate_flow_array(flow: Any, error_string: str = None) -> nd:
    """ndarray (H, W, 2), all finite -> float32 copy (reference utils.py:71-88)."""
    pre = error_string or ''
    if not isinstance(flow, np.ndarray):
        raise TypeError(pre + "Flow is not a numpy array")
    if flow.ndim != 3:
        raise ValueError(pre + "Flow array is not 3-dimensional")
    if flow.shape[2] != 2:
        raise ValueError(pre + "Flow array does not have 2 channels")
    if not np.isfinite(flow).all():
        raise ValueError(pre + "Flow array contains NaN or Inf values")
    return flow.astype('float32')


# --------------------------------------------------------------------------- flow generators
def matrix_from_transform(transform: str, values: list) -> nd:
    """3x3 matrix of one 'translation' / 'rotation' / 'scaling' (reference utils.py:131-158).
    Rotation angles are degrees, counter-clockwise, with the y axis pointing down."""
    m = np.identity(3)
    if transform == 'translation':
        m[0, 2], m[1, 2] = values[0], values[1]
        return m
    if transform in ('scaling', 'rotation'):
        to_origin, back = np.identity(3), np.identity(3)
        to_origin[0, 2], to_origin[1, 2] = -values[0], -values[1]
        back[0, 2], back[1, 2] = values[0], values[1]
        if transform == 'scaling':
            m[0, 0] = m[1, 1] = values[2]
        else:
            a = math.radians(values[2])
            m[0, 0], m[0, 1], m[1, 0], m[1, 1] = math.cos(a), math.sin(a), -math.sin(a), math.cos(a)
        return back @ m @ to_origin
    return m


def matrix_from_transforms(transform_list: list) -> nd:
    """Product of the individual transform matrices, first transform applied first (utils.py:114-128)."""
    m = np.identity(3)
    for t in reversed(transform_list):
        m = m @ matrix_from_transform(t[0], t[1:])
    return m


def flow_from_matrix(matrix: nd, shape: Union[list, tuple]) -> nd:
    """'s'-reference flow of a projective transform: warped grid minus grid (reference utils.py:91-111)."""
    h, w = shape
    grid = np.zeros((h, w, 3), 'f')
    grid[..., 0] += np.arange(w)
    grid[..., 1] += np.arange(h)[:, np.newaxis]
    grid[..., 2] = 1
    moved = np.squeeze(np.matmul(matrix, grid[..., np.newaxis]))
    return np.array(moved[..., 0:2] / moved[..., 2, np.newaxis] - grid[..., 0:2], 'float32')


def from_matrix(matrix: nd, shape: Union[list, tuple], ref: str) -> nd:
    """Flow vectors of a 3x3 transformation matrix (reference utils.py:319-344)."""
    validate_shape(shape)
    if not isinstance(matrix, np.ndarray):
        raise TypeError("Error creating flow from matrix: Matrix needs to be a numpy array")
    if matrix.shape != (3, 3):
        raise ValueError("Error creating flow from matrix: Matrix needs to be a numpy array of shape (3, 3)")
    if get_valid_ref(ref) == 's':
        return flow_from_matrix(matrix, shape)
    return -flow_from_matrix(np.linalg.pinv(matrix), shape)


_N_VALUES = {'translation': 2, 'rotation': 3, 'scaling': 3}


def from_transforms(transform_list: list, shape: Union[list, tuple], ref: str) -> nd:
    """Flow vectors of a list of [name, value, ...] transforms (reference utils.py:347-423)."""
    validate_shape(shape)
    if not isinstance(transform_list, list):
        raise TypeError("Error creating flow from transforms: Transform_list needs to be a list")
    if not all(isinstance(t, list) for t in transform_list):
        raise TypeError("Error creating flow from transforms: Transform_list needs to be a list of lists")
    if not all(len(t) > 1 for t in transform_list):
        raise ValueError("Error creating flow from transforms: Invalid transforms passed")
    for t in transform_list:
        if t[0] not in _N_VALUES:
            raise ValueError("Error creating flow from transforms: Transform '{}' not recognised".format(t[0]))
        if len(t) - 1 != _N_VALUES[t[0]]:
            raise ValueError("Error creating flow from transforms: Not enough transform values passed for "
                             "'{}' - expected {}, got {}".format(t[0], _N_VALUES[t[0]], len(t) - 1))
        if not all(isinstance(v, (float, int)) for v in t[1:]):
            raise ValueError("Error creating flow from transforms: "
                             "Transform values for '{}' need to be integers or floats".format(t[0]))
    ref = get_valid_ref(ref)
    return from_matrix(matrix_from_transforms(transform_list), shape, ref)


def load_sintel(path: str) -> nd:
    """Sintel .flo: 'PIEH', int32 w, int32 h, little-endian float32 [h, w, 2] (reference utils.py:447-470)."""
    if not isinstance(path, str):
        raise TypeError("Error loading flow from Sintel data: Path needs to be a string")
    with open(path, 'rb') as f:
        if f.read(4) != b'PIEH':
            raise ValueError("Error loading flow from Sintel data: Path not a valid .flo file")
        w = int.from_bytes(f.read(4), 'little')
        h = int.from_bytes(f.read(4), 'little')
        return np.fromfile(f, dtype=np.dtype('<f4')).reshape(h, w, 2)


def save_sintel(path: str, flow: nd) -> None:
    """Writer for the Sintel .flo layout load_sintel reads (reference utils.py:447-470 is read-only): 'PIEH', int32 w,
    int32 h, little-endian float32 [h, w, 2]."""
    if not isinstance(path, str):
        raise TypeError("Error saving flow as Sintel data: Path needs to be a string")
    flow = validate_flow_array(flow, "Error saving flow as Sintel data: ")
    h, w = flow.shape[:2]
    with open(path, 'wb') as f:
        f.write(b'PIEH')
        f.write(int(w).to_bytes(4, 'little'))
        f.write(int(h).to_bytes(4, 'little'))
        f.write(np.ascontiguousarray(flow, dtype='<f4').tobytes())


def _read_png_or_none(path):
    from ._png import read_png
    try:
        return read_png(path)
    except (OSError, ValueError, TypeError, IndexError, KeyError):
        return None


def load_kitti(path: str) -> nd:
    """KITTI uint16 PNG flow -> float64 (H, W, 3): channels (u, v) = (value - 2^15) / 64 and the valid flag
    (reference utils.py:426-444; PNG decoding by oflibnumpy_amd._png instead of cv2.imread)."""
    inp = _read_png_or_none(path)
    if inp is None:
        raise ValueError("Error loading flow from KITTI data: Flow data could not be loaded")
    if inp.ndim != 3 or inp.shape[-1] != 3:
        raise ValueError("Error loading flow from KITTI data: Loaded flow data has the wrong shape")
    out = inp.astype('float64')
    out[..., :2] = (out[..., :2] - 2 ** 15) / 64
    return out


def load_sintel_mask(path: str) -> nd:
    """Sintel invalid-pixel PNG -> boolean mask of VALID pixels (reference utils.py:473-490)."""
    if not isinstance(path, str):
        raise TypeError("Error loading flow from Sintel data: Path needs to be a string")
    mask = _read_png_or_none(path)
    if mask is None:
        raise ValueError("Error loading flow from Sintel data: Invalid mask could not be loaded from path")
    if mask.ndim == 3:
        mask = mask[..., :3].any(axis=-1)
    return ~(mask.astype('bool'))


# --------------------------------------------------------------------------- zero-flow predicates
def resize_flow(flow: nd, scale: Union[float, int, list, tuple]) -> nd:
    """Resize a flow field array and scale its vectors accordingly (reference utils.py:493-525): bilinear
    cv2.resize semantics, one launch of the resize kernel (ofl_resize_flow)."""
    flow = validate_flow_array(flow, "Error resizing flow: ")
    dev.resize_scales(scale)
    return dev.resize_host(flow, None, scale)[0]


def threshold_vectors(vecs: nd, threshold: Union[float, int] = None, use_mag: bool = None) -> nd:
    """Copy with small vectors zeroed: per component |v| < threshold, or by magnitude (utils.py:298-316)."""
    threshold = DEFAULT_THRESHOLD if threshold is None else threshold
    out = vecs.copy()
    if use_mag:
        out[np.linalg.norm(vecs, axis=-1) < threshold] = 0
    else:
        out[(vecs < threshold) & (vecs > -threshold)] = 0
    return out


def is_zero_flow(flow: nd, thresholded: bool = None) -> bool:
    """True if every (optionally thresholded) vector component is zero (reference utils.py:527-544)."""
    flow = validate_flow_array(flow, "Error checking whether flow is zero: ")
    thresholded = True if thresholded is None else thresholded
    if not isinstance(thresholded, bool):
        raise TypeError("Error checking whether flow is zero: Thresholded needs to be a boolean")
    bits = dev.flow_stats(dev.DeviceBuffer.from_host(flow), None, flow.shape[0] * flow.shape[1])
    return not (bits & (nat.STAT_NONZERO_TH if thresholded else nat.STAT_NONZERO))


# --------------------------------------------------------------------------- THE seam: apply_flow
def _remap_rules(dtype):
    """(arith, rule) the reference's cv2.remap call implies for an array of `dtype`."""
    if dtype == np.uint8:
        return nat.ARITH_NATIVE, nat.RULE_GE_HALF
    if dtype in (np.int16, np.uint16):
        return nat.ARITH_NATIVE, nat.RULE_GT_HALF
    return nat.ARITH_NATIVE, nat.RULE_EQ1


def apply_flow(flow: nd, target: nd, ref: str, mask: nd = None, quant: int = None) -> nd:
    """Warp `target` (H, W) or (H, W, C) with `flow` (H, W, 2); same meaning, validation and return
    conventions as the reference's apply_flow (utils.py:199-261).

    't': bilinear gather at grid - flow, zero outside (replaces cv2.remap, utils.py:231-236).
    's': scattered points grid + flow (only where `mask`) interpolated back onto the grid
         (replaces scipy griddata, utils.py:237-258).
    `quant` (extension): nat.QUANT_OPENCV (default, cv2's 1/32-px coordinate snapping) or QUANT_EXACT.
    """
    ref = get_valid_ref(ref)
    flow = validate_flow_array(flow, "Error applying flow to a target: ")
    if not isinstance(target, np.ndarray):
        raise TypeError("Error applying flow to a target: Target needs to be a numpy array")
    if target.ndim < 2 or target.ndim > 3:
        raise ValueError("Error applying flow to a target: Target array needs to have shape H-W or H-W-C")
    if target.shape[:2] != flow.shape[:2]:
        raise ValueError("Error applying flow to a target: Target height and width needs to match flow field array")
    if mask is not None:
        if not isinstance(mask, np.ndarray):
            raise TypeError("Error applying flow to a target: Mask needs to be a numpy array")
        if mask.shape != flow.shape[:2]:
            raise ValueError("Error applying flow to a target: Mask height and width needs to match flow field array")
        if mask.dtype != bool:
            raise TypeError("Error applying flow to a target: Mask needs to be boolean")
    quant = nat.QUANT_OPENCV if quant is None else quant
    if ref == 't' and target.dtype.type not in _REMAP_DTYPES:
        raise TypeError("Error applying flow to a target: dtype {} is not supported by the bilinear "
                        "remap (uint8, int16, uint16, float32, float64)".format(target.dtype))
    # the reference tests for a zero flow before it looks at the target (utils.py:214-216); here the arguments are
    # validated first because the zero test is a device reduction over the uploaded field
    fbuf = dev.DeviceBuffer.from_host(flow)
    if not (dev.flow_stats(fbuf, None, flow.shape[0] * flow.shape[1]) & nat.STAT_NONZERO_TH):
        return target
    if ref == 't':
        src = dev.DeviceImage.from_host(target)
        arith, _ = _remap_rules(target.dtype)
        dst, _ = dev.gather_bilinear(src, fbuf, flow.shape[:2], -1, quant=quant, arith=arith)
        result = dst.to_host()
        if target.ndim == 2:
            result = result[:, :, 0]
    else:
        result, _ = dev.scatter_host(fbuf, target if target.ndim == 3 else target[..., np.newaxis], mask)
        if target.ndim == 2:
            result = result[:, :, 0]
    if result.shape != target.shape:
        result = result[:, :, np.newaxis]
    return result


# --------------------------------------------------------------------------- sparse point tracking
def track_pts(flow: nd, ref: str, pts: nd, int_out: bool = None, s_exact_mode: bool = None) -> nd:
    """Warp points (N, 2) given as (row, col) with a flow field; arguments, defaults, return conventions and
    exception types of the reference's track_pts (utils.py:547-622).

    's' + integer points: the flow vectors at those pixels (host indexing, nothing to compute);
    's' + float points:   bilinear samples of the flow (device), or with `s_exact_mode` the Delaunay-linear
                          interpolation on the regular grid (device scatter kernel, query mode);
    't':                  Delaunay-linear interpolation of the flow carried by the points grid - flow,
                          evaluated at `pts` (device scatter kernel, query mode); points outside the hull -> 0.
    """
    flow = validate_flow_array(flow, "Error tracking points: ")
    if not isinstance(pts, np.ndarray):
        raise TypeError("Error tracking points: Pts needs to be a numpy array")
    if pts.ndim != 2 or pts.shape[1] != 2:
        raise ValueError("Error tracking points: Pts needs to have shape N-2")
    int_out = False if int_out is None else int_out
    s_exact_mode = False if s_exact_mode is None else s_exact_mode
    if not isinstance(int_out, bool):
        raise TypeError("Error tracking points: Int_out needs to be a boolean")
    if not isinstance(s_exact_mode, bool):
        raise TypeError("Error tracking points: S_exact_mode needs to be a boolean")

    if is_zero_flow(flow, thresholded=True):
        warped = pts
    else:
        h, w = flow.shape[:2]
        if ref == 's' and np.issubdtype(pts.dtype, np.integer):
            warped = pts + flow[pts[:, 0], pts[:, 1], ::-1]
        elif ref == 's' and not np.issubdtype(pts.dtype, np.floating):
            raise TypeError("Error tracking points: Pts numpy array needs to have a float or int dtype")
        else:
            fbuf = dev.DeviceBuffer.from_host(flow)
            if ref == 's' and not s_exact_mode:
                ver, hor = pts[:, 0], pts[:, 1]
                if np.any(~((0 <= ver) & (ver <= h - 1)) | ~((0 <= hor) & (hor <= w - 1))):
                    raise IndexError("Some points are outside of the data area.")
                vecs = dev.sample_points(fbuf, h, w, pts)
                found = np.ones(len(pts), bool)
            else:
                if ref == 's':      # exact mode: values live on the undisplaced regular grid
                    pos, sign = dev.DeviceBuffer.zeros(h * w * 8), 1
                else:               # 't': values live at grid - flow
                    pos, sign = fbuf, -1
                vals, found = dev.scatter_query(pos, sign, fbuf, 2, h, w, pts[:, ::-1].astype(np.float64))
                vecs = vals[:, ::-1]            # (u, v) -> (row, col) order
            warped = pts + vecs
            warped[~found] = 0                  # NaN -> 0 (reference utils.py:616-618)
    if int_out:
        warped = np.round(warped).astype('i')
    return warped
