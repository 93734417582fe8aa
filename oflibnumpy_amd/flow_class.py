"""The `Flow` class: same public surface, defaults, validation and exception types as the reference's
hot path (src/oflibnumpy/flow_class.py), with every warp executed by the MI355X engine.

A `Flow` owns host NumPy arrays (vecs float32 (H, W, 2), mask bool (H, W), ref 's'/'t') exactly like
the reference.  Its hot-path methods upload once, run the device-side algebra of `device.DeviceFlow`
(all intermediates stay in HBM) and download the result.  Pipelines that must not cross PCIe at all
use `DeviceFlow` directly.

Next-tier row already widened into: `track` / `track_pts`.
Dataset loaders (KITTI, Sintel + mask) are host I/O with a small built-in PNG reader.
Out of scope (not on the hot path): matrix fitting, visualisation.
"""
from __future__ import annotations

import warnings
from typing import Tuple, Union

import numpy as np

from . import _native as nat
from . import device as dev
from .utils import (get_valid_ref, get_valid_padding, validate_shape, from_matrix, from_transforms,
                    load_sintel, load_sintel_mask, load_kitti, is_zero_flow, threshold_vectors, track_pts,
                    _REMAP_DTYPES)

FlowAlias = 'Flow'


class Flow(object):
    _vecs: np.ndarray
    _mask: np.ndarray
    _ref: str

    def __init__(self, flow_vectors: np.ndarray, ref: str = None, mask: np.ndarray = None):
        """flow_vectors (H, W, 2) in OpenCV convention (channel 0 horizontal, 1 vertical), `ref` 't'
        (default) or 's', `mask` bool (H, W) of valid vectors (default all True).
        Reference: flow_class.py:38-51."""
        self.vecs = flow_vectors
        self.ref = ref
        self.mask = mask

    # ------------------------------------------------------------------ attributes
    @property
    def vecs(self) -> np.ndarray:
        return self._vecs

    @vecs.setter
    def vecs(self, input_vecs: np.ndarray):
        # reference flow_class.py:65-81: always a fresh float32 copy
        if not isinstance(input_vecs, np.ndarray):
            raise TypeError("Error setting flow vectors: Input is not a numpy array")
        if input_vecs.ndim != 3:
            raise ValueError("Error setting flow vectors: Input not 3-dimensional")
        if input_vecs.shape[2] != 2:
            raise ValueError("Error setting flow vectors: Input does not have 2 channels")
        if not np.isfinite(input_vecs).all():
            raise ValueError("Error setting flow vectors: Input contains NaN, Inf or -Inf values")
        self._vecs = input_vecs.astype('float32')

    @property
    def ref(self) -> str:
        return self._ref

    @ref.setter
    def ref(self, input_ref: str = None):
        self._ref = get_valid_ref(input_ref)

    @property
    def mask(self) -> np.ndarray:
        return self._mask

    @mask.setter
    def mask(self, input_mask: np.ndarray = None):
        # reference flow_class.py:142-161
        if input_mask is None:
            self._mask = np.ones(self.shape, 'bool')
            return
        if not isinstance(input_mask, np.ndarray):
            raise TypeError("Error setting flow mask: Input is not a numpy array")
        if input_mask.ndim != 2:
            raise ValueError("Error setting flow mask: Input not 2-dimensional")
        if input_mask.shape != self.shape:
            raise ValueError("Error setting flow mask: Input has a different shape than the flow vectors")
        if ((input_mask != 0) & (input_mask != 1)).any():
            raise ValueError("Error setting flow mask: Values must be 0 or 1")
        self._mask = input_mask.astype('bool')

    @property
    def shape(self) -> tuple:
        return self._vecs.shape[:2]

    # ------------------------------------------------------------------ constructors
    @classmethod
    def zero(cls, shape: Union[list, tuple], ref: str = None, mask: np.ndarray = None) -> FlowAlias:
        validate_shape(shape)
        return cls(np.zeros((shape[0], shape[1], 2)), ref, mask)

    @classmethod
    def from_matrix(cls, matrix: np.ndarray, shape: Union[list, tuple], ref: str = None,
                    mask: np.ndarray = None) -> FlowAlias:
        return cls(from_matrix(matrix, shape, ref), ref, mask)

    @classmethod
    def from_transforms(cls, transform_list: list, shape: Union[list, tuple], ref: str = None,
                        mask: np.ndarray = None) -> FlowAlias:
        return cls(from_transforms(transform_list, shape, ref), ref, mask)

    @classmethod
    def from_kitti(cls, path: str, load_valid: bool = None) -> FlowAlias:
        """KITTI uint16 PNG -> flow with reference 's', optionally with the valid pixels as mask
        (reference flow_class.py:237-259)."""
        load_valid = True if load_valid is None else load_valid
        if not isinstance(load_valid, bool):
            raise TypeError("Error loading flow from KITTI data: Load_valid needs to be boolean")
        data = load_kitti(path)
        return cls(data[..., :2], 's', data[..., 2].astype('bool')) if load_valid else cls(data[..., :2], 's')

    @classmethod
    def from_sintel(cls, path: str, inv_path: str = None) -> FlowAlias:
        """Sintel .flo (+ optional invalid-pixel PNG) -> flow with reference 's' (reference flow_class.py:262-275)."""
        mask = None if inv_path is None else load_sintel_mask(inv_path)
        return cls(load_sintel(path), 's', mask)

    def copy(self) -> FlowAlias:
        return Flow(self._vecs, self._ref, self._mask)

    def __str__(self) -> str:
        return "Flow object, reference {}, shape {}*{}; ".format(self._ref, *self.shape) + self.__repr__()

    def __getitem__(self, item) -> FlowAlias:
        return Flow(self._vecs.__getitem__(item), self._ref, self._mask.__getitem__(item))

    # ------------------------------------------------------------------ element-wise operators
    def _check_other(self, other, verb, noun):
        if not isinstance(other, (np.ndarray, Flow)):
            raise TypeError("Error {} flow: {} is not a flow object or a numpy array".format(verb, noun))
        if isinstance(other, Flow):
            if self.shape != other.shape:
                raise ValueError("Error {} flow: flow objects are not the same shape".format(verb))
        elif self.shape != other.shape[:2] or other.ndim != 3 or other.shape[2] != 2:
            raise ValueError("Error {} flow: numpy array needs to have the same shape as the flow object, "
                             "3 dimensions overall, and a channel length of 2".format(verb))

    def __add__(self, other: Union[np.ndarray, FlowAlias]) -> FlowAlias:
        # reference flow_class.py:310-341: keeps the left operand's ref, masks are ANDed
        self._check_other(other, "adding to", "Addend")
        if isinstance(other, Flow):
            return Flow(self._vecs + other._vecs, self._ref, np.logical_and(self._mask, other._mask))
        return Flow(self._vecs + other, self._ref, self._mask)

    def __sub__(self, other: Union[np.ndarray, FlowAlias]) -> FlowAlias:
        self._check_other(other, "subtracting from", "Subtrahend")
        if isinstance(other, Flow):
            return Flow(self._vecs - other._vecs, self._ref, np.logical_and(self._mask, other._mask))
        return Flow(self._vecs - other, self._ref, self._mask)

    def _broadcast_operand(self, other, verb, noun):
        """Scalar / list of 2 / array (2,), (H, W) or (H, W, 2) -> something NumPy can broadcast
        (reference flow_class.py:377-477)."""
        try:
            return float(other)
        except TypeError:
            pass
        if isinstance(other, list):
            if len(other) != 2:
                raise ValueError("Error {} flow: {} list not length 2".format(verb, noun))
            return np.array(other)[np.newaxis, np.newaxis, :]
        if isinstance(other, np.ndarray):
            if other.ndim == 1 and other.size == 2:
                return other[np.newaxis, np.newaxis, :]
            if other.ndim == 2 and other.shape == self.shape[:2]:
                return other[:, :, np.newaxis]
            if other.shape == self.shape + (2,):
                return other
            raise ValueError("Error {} flow: {} array is not one of the following: size 2, shape of the "
                             "flow object, shape of the flow vectors".format(verb, noun))
        raise TypeError("Error {} flow: {} cannot be converted to float, or isn't a list or numpy array"
                        .format(verb, noun))

    def __mul__(self, other) -> FlowAlias:
        return Flow(self._vecs * self._broadcast_operand(other, "multiplying", "Multiplier"), self._ref, self._mask)

    def __truediv__(self, other) -> FlowAlias:
        return Flow(self._vecs / self._broadcast_operand(other, "dividing", "Divisor"), self._ref, self._mask)

    def __pow__(self, other) -> FlowAlias:
        return Flow(self._vecs ** self._broadcast_operand(other, "exponentiating", "Exponent"), self._ref, self._mask)

    def __neg__(self) -> FlowAlias:
        return self * -1

    def resize(self, scale: Union[float, int, list, tuple]) -> FlowAlias:
        """Resize the flow field, scaling the vectors accordingly; the mask is resized bilinearly and rounded
        (flow_class.py:491-506)."""
        dev.resize_scales(scale)
        vecs, mask = dev.resize_host(self._vecs, self._mask, scale)
        return Flow(vecs, self._ref, mask)

    def pad(self, padding: Union[list, tuple] = None, mode: str = None) -> FlowAlias:
        """Pad vecs ('constant' zeros, 'edge' or 'symmetric') and the mask with False (flow_class.py:508-526)."""
        mode = 'constant' if mode is None else mode
        if mode not in ('constant', 'edge', 'symmetric'):
            raise ValueError("Error padding flow: Mode should be one of "
                             "'constant', 'edge', 'symmetric', but instead got '{}'".format(mode))
        p = get_valid_padding(padding, "Error padding flow: ")
        vecs = np.pad(self._vecs, ((p[0], p[1]), (p[2], p[3]), (0, 0)), mode=mode)
        mask = np.pad(self._mask, ((p[0], p[1]), (p[2], p[3])))
        return Flow(vecs, self._ref, mask)

    # ------------------------------------------------------------------ device plumbing
    def to_device(self) -> dev.DeviceFlow:
        return dev.DeviceFlow.from_host(self._vecs, self._ref, self._mask)

    @classmethod
    def from_device(cls, dflow: dev.DeviceFlow) -> FlowAlias:
        """Download a device result.  The arrays come out of the kernels as float32 / 0-1 bytes computed from
        already validated inputs, so the constructor's copy-and-validate passes (3 x the PCIe time at 4K) are
        skipped."""
        vecs, mask = dflow.to_host()
        f = cls.__new__(cls)
        f._vecs, f._mask, f._ref = vecs, mask, get_valid_ref(dflow.ref)
        return f

    # ------------------------------------------------------------------ hot path
    def apply(self, target: Union[np.ndarray, FlowAlias], target_mask: np.ndarray = None,
              return_valid_area: bool = None, consider_mask: bool = None,
              padding: Union[list, tuple] = None, cut: bool = None
              ) -> Union[Union[np.ndarray, FlowAlias], Tuple[Union[np.ndarray, FlowAlias], np.ndarray]]:
        """Warp `target` (ndarray (H, W[, C]) or Flow) with this flow.  Arguments, defaults, return
        conventions and raised exception types follow the reference (flow_class.py:528-695)."""
        return_valid_area = False if return_valid_area is None else return_valid_area
        if not isinstance(return_valid_area, bool):
            raise TypeError("Error applying flow: Return_valid_area needs to be a boolean")
        consider_mask = True if consider_mask is None else consider_mask
        if not isinstance(consider_mask, bool):
            raise TypeError("Error applying flow: Consider_mask needs to be a boolean")
        cut = True if cut is None else cut
        if not isinstance(cut, bool):
            raise TypeError("Error applying flow: Cut needs to be a boolean")
        if padding is None:
            if self.shape[0] != target.shape[0] or self.shape[1] != target.shape[1]:
                raise ValueError("Error applying flow: Flow shape does not match target shape")
        else:
            padding = get_valid_padding(padding, "Error applying flow: ")
            if self.shape[0] + padding[0] + padding[1] != target.shape[0] or \
                    self.shape[1] + padding[2] + padding[3] != target.shape[1]:
                raise ValueError("Error applying flow: Padding values do not match flow and target shape difference")

        return_2d = False
        if isinstance(target, Flow):
            return_flow, t, tmask, default_mask = True, target._vecs, target._mask, False
        elif isinstance(target, np.ndarray):
            return_flow = False
            if target.ndim == 3:
                t = target
            elif target.ndim == 2:
                return_2d, t = True, target[..., np.newaxis]
            else:
                raise ValueError("Error applying flow: Target needs to have the shape H-W (2 dimensions) "
                                 "or H-W-C (3 dimensions)")
            default_mask = target_mask is None
            if default_mask:
                tmask = np.ones(t.shape[:2], bool)
            else:
                if not isinstance(target_mask, np.ndarray):
                    raise TypeError("Error applying flow: Target_mask needs to be a numpy ndarray")
                if target_mask.shape != target.shape[:2]:
                    raise ValueError("Error applying flow: Target_mask needs to match the target shape")
                if target_mask.dtype != bool:
                    raise TypeError("Error applying flow: Target_mask needs to have dtype 'bool'")
                if not return_valid_area:
                    warnings.warn("Warning applying flow: a mask is passed, but return_valid_area is False - so the "
                                  "mask passed will not affect the output, but possibly make the function slower.")
                tmask = target_mask
        else:
            raise ValueError("Error applying flow: Target needs to be either a flow object, or a numpy ndarray")

        with_mask = return_flow or return_valid_area
        pad_tl = (0, 0) if padding is None else (padding[0], padding[2])
        if self._ref == 't':
            warped, valid = self._apply_t(t, tmask, default_mask, with_mask, pad_tl)
        else:
            warped, valid = self._apply_s(t, tmask, with_mask, consider_mask, padding)

        if padding is not None and cut:
            sl = (slice(padding[0], padding[0] + self.shape[0]), slice(padding[2], padding[2] + self.shape[1]))
            warped = warped[sl]
            valid = valid[sl] if valid is not None else None

        if return_flow:
            return Flow(warped, target._ref, valid)
        if return_2d:
            warped = warped[:, :, 0]
        warped = np.ascontiguousarray(warped)
        return (warped, valid) if return_valid_area else warped

    def _apply_t(self, t, tmask, default_mask, with_mask, pad_tl):
        """'t' reference: one launch of the gather kernel K1 (replaces flow_class.py:644-650 +
        cv2.remap).  The reference appends the mask as an extra channel, which fixes the dtype the
        remap runs in; the same arithmetic is selected here without materialising the concat."""
        if t.dtype.type not in _REMAP_DTYPES:
            raise TypeError("Error applying flow: target dtype {} is not supported by the bilinear remap "
                            "(uint8, int16, uint16, float32, float64)".format(t.dtype))
        arith, rule = nat.ARITH_NATIVE, nat.RULE_EQ1
        if with_mask:
            # dtype of np.concatenate((t, mask[..., None])) with mask int8 (default, :615) or bool (:626)
            concat = np.result_type(t.dtype, np.int8 if default_mask else np.bool_)
            if concat.type not in _REMAP_DTYPES:
                raise TypeError("Error applying flow: target dtype {} with a validity mask needs a {} remap, "
                                "which cv2.remap does not provide".format(t.dtype, concat))
            if concat == np.uint8:
                rule = nat.RULE_GE_HALF
            elif concat in (np.int16, np.uint16):
                rule = nat.RULE_GT_HALF
                if t.dtype == np.uint8:
                    arith = nat.ARITH_FLOAT_RNE        # uint8 image inside an int16 concat
        fbuf = dev.DeviceBuffer.from_host(self._vecs)
        if not (dev.flow_stats(fbuf, None, self._vecs.shape[0] * self._vecs.shape[1]) & nat.STAT_NONZERO_TH):
            warped = t                                      # identity short cut, reference utils.py:215-216
            valid = None
            if with_mask:
                valid = np.zeros(t.shape[:2], bool)
                valid[pad_tl[0]:pad_tl[0] + self.shape[0], pad_tl[1]:pad_tl[1] + self.shape[1]] = self._mask
                valid &= tmask.astype(bool)
            return warped, valid
        src = dev.DeviceImage.from_host(t)
        smask = dev.DeviceBuffer.from_host(tmask.astype(np.uint8)) if (with_mask and not default_mask) else None
        fmask = dev.DeviceBuffer.from_host(self._mask.view(np.uint8)) if with_mask else None
        dst, valid = dev.gather_bilinear(src, fbuf, self.shape, -1, smask=smask, fmask=fmask,
                                         want_valid=with_mask, pad=pad_tl, arith=arith, rule=rule)
        warped = dst.to_host()
        valid = valid.to_host(t.shape[:2], np.uint8).view(np.bool_) if with_mask else None
        return warped, valid

    def _apply_s(self, t, tmask, with_mask, consider_mask, padding):
        """'s' reference: scattered -> grid interpolation kernel K3 (replaces flow_class.py:634-660 +
        scipy griddata).  With padding the flow is edge-padded first (flow_class.py:652-659)."""
        flow = self if padding is None else self.pad(padding, mode='edge')
        if with_mask:
            tmask = tmask.astype(bool) & flow._mask                  # flow_class.py:636-643
        fbuf = dev.DeviceBuffer.from_host(flow._vecs)           # uploaded once: zero test and scatter share it
        if not (dev.flow_stats(fbuf, None, flow.shape[0] * flow.shape[1]) & nat.STAT_NONZERO_TH):
            return t, (tmask.copy() if with_mask else None)        # identity short cut, utils.py:215-216
        return dev.scatter_host(fbuf, t, flow._mask if consider_mask else None,
                                vmask=tmask if with_mask else None)

    def switch_ref(self, mode: str = None) -> FlowAlias:
        """Switch between 's' and 't' reference (reference flow_class.py:697-733)."""
        mode = 'valid' if mode is None else mode
        if mode == 'invalid':
            return Flow(self._vecs, 't' if self._ref == 's' else 's', self._mask)
        if mode != 'valid':
            raise ValueError("Error switching flow reference: Mode not recognised, should be 'valid' or 'invalid'")
        d = self.to_device()
        if d.is_zero(thresholded=False):
            return self.switch_ref(mode='invalid')
        return Flow.from_device(d.switch_ref())

    def invert(self, ref: str = None) -> FlowAlias:
        """Inverse flow in the requested reference (reference flow_class.py:735-753)."""
        ref = self._ref if ref is None else get_valid_ref(ref)
        if ref != self._ref:
            return Flow(-self._vecs, ref, self._mask)          # s->t and t->s are a negation only
        return Flow.from_device(self.to_device().invert(ref))

    def track(self, pts: np.ndarray, int_out: bool = None, get_valid_status: bool = None,
              s_exact_mode: bool = None) -> np.ndarray:
        """Warp points (N, 2) in (row, col) order with the flow; optionally also return which points are
        moved by valid vectors to a position inside the flow area (reference flow_class.py:755-795)."""
        get_valid_status = False if get_valid_status is None else get_valid_status
        if not isinstance(get_valid_status, bool):
            raise TypeError("Error tracking points: Get_tracked needs to be a boolean")
        warped = track_pts(flow=self._vecs, ref=self._ref, pts=pts, int_out=int_out, s_exact_mode=s_exact_mode)
        if get_valid_status:
            status = self.valid_source()[np.round(pts[..., 0]).astype('i'), np.round(pts[..., 1]).astype('i')]
            return warped, status
        return warped

    def valid_target(self, consider_mask: bool = None) -> np.ndarray:
        """Area of the target domain reached by valid vectors (reference flow_class.py:1113-1151)."""
        consider_mask = True if consider_mask is None else consider_mask
        if not isinstance(consider_mask, bool):
            raise TypeError("Error applying flow: Consider_mask needs to be a boolean")
        buf = self.to_device().valid_target(consider_mask)
        return buf.to_host(self.shape, np.uint8).view(np.bool_)

    def valid_source(self, consider_mask: bool = None) -> np.ndarray:
        """Area of the source domain that ends up valid in the target (reference flow_class.py:1153-1195)."""
        consider_mask = True if consider_mask is None else consider_mask
        if not isinstance(consider_mask, bool):
            raise TypeError("Error applying flow: Consider_mask needs to be a boolean")
        buf = self.to_device().valid_source(consider_mask)
        return buf.to_host(self.shape, np.uint8).view(np.bool_)

    def get_padding(self) -> list:
        """[top, bottom, left, right] padding needed so that no valid vector leaves the padded area
        (reference flow_class.py:1197-1228)."""
        return self.to_device().get_padding()

    def is_zero(self, thresholded: bool = None, masked: bool = None) -> bool:
        """All (masked) vectors zero?  (reference flow_class.py:1230-1245)"""
        masked = True if masked is None else masked
        if not isinstance(masked, bool):
            raise TypeError("Error checking whether flow is zero: Masked needs to be a boolean")
        thresholded = True if thresholded is None else thresholded
        if not isinstance(thresholded, bool):
            raise TypeError("Error checking whether flow is zero: Thresholded needs to be a boolean")
        return self.to_device().is_zero(thresholded, masked)

    def combine_with(self, flow: FlowAlias, mode: int, thresholded: bool = None) -> FlowAlias:
        """flow_1 (+) flow_2 = flow_3: mode k returns flow_k from the other two (`self` comes first in
        that formula among the two given).  Reference flow_class.py:1247-1424."""
        if not isinstance(flow, Flow):
            raise TypeError("Error combining flows: Flow need to be of type 'Flow'")
        if self.shape != flow.shape:
            raise ValueError("Error combining flows: Flow fields need to have the same shape")
        if self.ref != flow.ref:
            raise ValueError("Error combining flows: Flow fields need to have the same reference")
        if mode not in [1, 2, 3]:
            raise ValueError("Error combining flows: Mode needs to be 1, 2 or 3")
        thresholded = False if thresholded is None else thresholded
        if not isinstance(thresholded, bool):
            raise TypeError("Error combining flows: Thresholded needs to be a boolean")

        d_self, d_flow = self.to_device(), flow.to_device()
        res = d_self.combine_with(d_flow, mode, thresholded)
        if res is d_flow:                      # zero-flow early exits hand back the operand itself
            return flow
        if res is d_self:
            return self
        return Flow.from_device(res)
