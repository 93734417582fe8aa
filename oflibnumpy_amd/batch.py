"""Batches of independent flow-field pairs (BASELINE.json config 4): one device launch per stack.

`DeviceFlowBatch` keeps B fields of one shape back to back in HBM ([B][H][W][2] float32 + [B][H][W] uint8),
the layout `ofl_compose3_dev` takes with `batch > 1`.  `combine_flows_batch` is the array-level convenience:
it composes the pairs this rank owns (`sharding.shard`) with mode 3 in one launch and honours the
reference's per-pair zero-flow early exits (flow_class.py:1339-1354) from the flag words of that launch.
"""
import numpy as np

from . import _native as nat
from . import device as dev
from . import sharding
from .flow_class import Flow


class DeviceFlowBatch:
    def __init__(self, n, shape, ref, packed=False):
        """packed: the masks live as bit planes (`self.bits`, include/ofl.h: ofl_compose3_bits_dev) instead of bytes -- for chains
        of compositions that stay on the device; `mask` is then produced on demand (unpack)."""
        self.n, self.shape, self.ref = int(n), (int(shape[0]), int(shape[1])), ref
        px = self.shape[0] * self.shape[1]
        self.vecs = dev.DeviceBuffer(self.n * px * 8)
        self.packed = bool(packed)
        if self.packed:
            self.bits = dev.DeviceBuffer(dev.mask_bits_bytes(self.shape[0], self.shape[1], self.n))
            self._mask = None
        else:
            self.bits = None
            self._mask = dev.DeviceBuffer(self.n * px)

    @property
    def mask(self):
        if self._mask is None:
            self._mask = dev.mask_unpack(self.bits, self.shape[0], self.shape[1], self.n)
        return self._mask

    def pack(self):
        """the same fields with their masks as bit planes (a copy of the masks only; the vectors are shared)"""
        if self.packed:
            return self
        b = DeviceFlowBatch.__new__(DeviceFlowBatch)
        b.n, b.shape, b.ref, b.vecs, b.packed = self.n, self.shape, self.ref, self.vecs, True
        b.bits = dev.mask_pack(self._mask, self.shape[0], self.shape[1], self.n)
        b._mask = self._mask
        return b

    @classmethod
    def from_flows(cls, flows, packed=False):
        flows = list(flows)
        if not flows:
            raise ValueError("empty batch")
        shape, ref = flows[0].shape, flows[0].ref
        if any(f.shape != shape or f.ref != ref for f in flows):
            raise ValueError("all flows of a batch need the same shape and reference")
        b = cls(len(flows), shape, ref)
        px = shape[0] * shape[1]
        lib = nat.load()
        for i, f in enumerate(flows):
            m = np.ascontiguousarray(f.mask).view(np.uint8)
            v = np.ascontiguousarray(f.vecs, np.float32)       # Flow.vecs keeps the layout of its input (astype order 'K')
            nat.check(lib.ofl_upload(b.vecs.ptr + i * px * 8, v.ctypes.data, px * 8, None))
            nat.check(lib.ofl_upload(b.mask.ptr + i * px, m.ctypes.data, px, None))
            nat.check(lib.ofl_stream_sync(None))
        return b.pack() if packed else b

    def to_flows(self):
        h, w = self.shape
        v = self.vecs.to_host((self.n, h, w, 2), np.float32)
        m = self.mask.to_host((self.n, h, w), np.uint8).view(np.bool_)
        return [Flow(v[i], self.ref, m[i]) for i in range(self.n)]

    def compose3(self, other, quant=nat.QUANT_OPENCV):
        """self[i].combine_with(other[i], mode=3) for every i in ONE launch.
        Returns (DeviceFlowBatch out, flag words uint32 [n][8]) -- see ofl_compose3_dev for the words.  Two PACKED batches
        compose on their bit planes (ofl_compose3_bits_dev) and give a packed result: the same fields, bit for bit."""
        if (self.n, self.shape, self.ref) != (other.n, other.shape, other.ref):
            raise ValueError("batches need the same length, shape and reference")
        fa, fb, sign = (other, self, +1) if self.ref == 's' else (self, other, -1)
        words = dev.DeviceBuffer.zeros(32 * self.n)
        if self.packed and other.packed and quant == nat.QUANT_OPENCV and self.shape[1] % 2 == 0:
            out = DeviceFlowBatch(self.n, self.shape, self.ref, packed=True)
            dev.compose3_bits_launch(fa.vecs, fa.bits, fb.vecs, fb.bits, sign, self.shape, out.vecs, out.bits, words, batch=self.n)
        else:
            out = DeviceFlowBatch(self.n, self.shape, self.ref)
            dev.compose3_launch(fa, fb, sign, out, words, batch=self.n, quant=quant)
        return out, words.to_host((self.n, 8), np.uint32), (fa, fb)


    def apply_images(self, images, dtype, channels, shared=False, target_masks=None, shared_masks=False, quant=nat.QUANT_OPENCV):
        """self[i].apply(image_i, target_mask_i, return_valid_area=True) for every i in ONE launch of the gather kernel
        (ref 't' batches; Flow.apply, flow_class.py:604-695 without padding).  `images`: a DeviceBuffer holding [n][H][W][C] of
        `dtype` back to back -- or ONE [H][W][C] image warped by every field when `shared`; `target_masks` uint8 [n][H][W] (one
        [H][W] when `shared_masks`) or None.  Returns (warped [n][H][W][C] DeviceBuffer, valid [n][H][W] DeviceBuffer), field
        for field what DeviceFlow.apply_image gives (the dtype rules of the reference's concatenated array, device.apply_image).
        A field whose vectors are all below the 1e-3 threshold is warped like any other: under cv2's 1/32-px coordinate
        snapping that IS the identity of utils.py:215-216."""
        if self.ref != 't':
            raise ValueError("apply_images batches the gather ('t') warp; 's' fields go through DeviceFlow.apply_image one by one")
        dtype = np.dtype(dtype)
        h, w = self.shape
        arith, rule = nat.ARITH_NATIVE, nat.RULE_EQ1
        if dtype == np.uint8:      # concat dtype of the reference: bool mask -> uint8, default int8 -> int16
            arith, rule = (nat.ARITH_NATIVE, nat.RULE_GE_HALF) if target_masks is not None else (nat.ARITH_FLOAT_RNE, nat.RULE_GT_HALF)
        elif dtype == np.int16 or (dtype == np.uint16 and target_masks is not None):
            rule = nat.RULE_GT_HALF
        elif dtype == np.uint16:
            raise TypeError("uint16 image with the default int8 mask needs an int32 remap, which cv2.remap does not provide")
        return dev.gather_bilinear_batch(images, dtype, channels, h, w, self.n, self.vecs, -1, smask=target_masks, fmask=self.mask,
                                         valid=True, shared_src=shared, shared_smask=shared_masks, quant=quant, arith=arith, rule=rule)


def combine_flows_batch(flows_1, flows_2, ref=None, rank=0, world=1, thresholded=False):
    """Mode-3 composition of many independent pairs.  `flows_1[i] (+) flows_2[i]`; inputs are lists of
    `Flow` objects or of (H, W, 2) arrays with reference `ref`.  With world > 1 only the contiguous block of
    pairs owned by `rank` is processed and returned, as a list of (index, Flow)."""
    if len(flows_1) != len(flows_2):
        raise ValueError("need as many first as second flows")
    as_flow = lambda f: f if isinstance(f, Flow) else Flow(f, ref)
    mine = sharding.shard(len(flows_1), rank, world)
    if len(mine) == 0:
        return []
    f1 = [as_flow(flows_1[i]) for i in mine]
    f2 = [as_flow(flows_2[i]) for i in mine]
    b1, b2 = DeviceFlowBatch.from_flows(f1), DeviceFlowBatch.from_flows(f2)
    out, words, (fa, fb) = b1.compose3(b2)
    res = out.to_flows()
    bit = 1 if thresholded else 0                 # OFL_STAT_NONZERO_TH_MASKED / OFL_STAT_NONZERO_MASKED word index
    results = []
    for k, i in enumerate(mine):
        self_f, other_f = f1[k], f2[k]
        a_cert, b_nonzero = words[k][bit], words[k][4 + bit]
        # fa/fb roles: 't' -> fa = self, fb = flow; 's' -> fa = flow, fb = self
        fa_zero = (not a_cert) and (other_f if b1.ref == 's' else self_f).is_zero(thresholded=thresholded)
        fb_zero = not b_nonzero
        self_zero, flow_zero = (fb_zero, fa_zero) if b1.ref == 's' else (fa_zero, fb_zero)
        if self_zero:
            r = other_f
        elif flow_zero:
            r = self_f
        elif not words[k][7]:                      # sampling field thresholded-zero: plain sum (utils.py:215-216)
            r = other_f + self_f if b1.ref == 't' else self_f + other_f
        else:
            r = res[k]
        results.append((i, r))
    return results
