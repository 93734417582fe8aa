"""HBM-resident flow fields and the device-side flow algebra.

`DeviceFlow` mirrors the hot-path methods of the reference's `Flow` (apply / switch_ref / invert /
combine_with / valid_target / valid_source / + - neg / is_zero; src/oflibnumpy/flow_class.py) but
keeps vectors and mask in GPU memory between operations, so chains such as combine_with(mode=1)
(1 scatter + 2 gathers, flow_class.py:1369-1370) never cross PCIe.  The host `Flow` class is a thin
upload -> DeviceFlow op -> download wrapper around this module.

This layer is SINGLE-STREAM: recycled buffers (`_Pool`) and the one scatter workspace per shape are handed out as soon
as Python drops them, which is only safe while all work is queued on one stream (the library's default).  The `stream`
arguments exist for callers that manage their own buffers at the C level.

Data layout in HBM: vecs float32 [H][W][2] interleaved (x, y) -- the same layout as the reference,
so a pixel's vector is one 8-byte element and two horizontally adjacent bilinear taps are one
16-byte load; mask uint8 [H][W] (0/1).
"""
import ctypes
import weakref

import numpy as np

from . import _native as nat

DEFAULT_THRESHOLD = 1e-3          # src/oflibnumpy/utils.py:22
_DT_CODE = {np.dtype('uint8'): nat.U8, np.dtype('int16'): nat.I16, np.dtype('uint16'): nat.U16,
            np.dtype('float32'): nat.F32, np.dtype('float64'): nat.F64}


def _lib():
    nat.ensure_device()
    return nat.load()


# ------------------------------------------------------------------------------ memory
class _Pool:
    """Size-bucketed free lists: hipFree synchronises the device, so chained operations recycle
    their intermediates instead of returning them to the driver."""

    def __init__(self):
        self.free = {}
        self.cached_bytes = 0
        self.limit = 64 << 30

    def take(self, nbytes):
        lst = self.free.get(nbytes)
        if lst:
            self.cached_bytes -= nbytes
            return lst.pop()
        p = ctypes.c_void_p()
        try:
            nat.check(_lib().ofl_malloc(ctypes.byref(p), nbytes))
        except nat.NativeError:
            self.trim()
            nat.check(_lib().ofl_malloc(ctypes.byref(p), nbytes))
        return p.value

    def give(self, ptr, nbytes):
        if self.cached_bytes + nbytes > self.limit:
            nat.load().ofl_free(ptr)
            return
        self.free.setdefault(nbytes, []).append(ptr)
        self.cached_bytes += nbytes

    def trim(self):
        lib = nat.load()
        for lst in self.free.values():
            for p in lst:
                lib.ofl_free(p)
        self.free.clear()
        self.cached_bytes = 0


_pool = _Pool()


def empty_cache():
    _pool.trim()


def _release(ptr, nbytes):
    try:
        _pool.give(ptr, nbytes)
    except Exception:       # interpreter shutdown
        pass


class DeviceBuffer:
    """A block of HBM owned by this process (ofl_malloc / pooled)."""

    __slots__ = ("ptr", "nbytes", "_fin", "__weakref__")

    def __init__(self, nbytes):
        self.nbytes = max(int(nbytes), 16)
        self.ptr = _pool.take(self.nbytes)
        self._fin = weakref.finalize(self, _release, self.ptr, self.nbytes)

    @classmethod
    def from_host(cls, arr, stream=None):
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        if arr.nbytes:
            nat.check(_lib().ofl_upload(buf.ptr, arr.ctypes.data, arr.nbytes, stream))
            # the source array may be a temporary: make the (possibly staged) copy complete now
            nat.check(_lib().ofl_stream_sync(stream))
        return buf

    @classmethod
    def zeros(cls, nbytes, stream=None):
        buf = cls(nbytes)
        nat.check(_lib().ofl_memset(buf.ptr, 0, buf.nbytes, stream))
        return buf

    def to_host(self, shape, dtype, stream=None):
        """Download into a fresh array.  Large results land in page-locked memory from a recycling pool (a DMA at link
        speed; a pageable destination of fresh pages costs 2-3 x as long in page faults) -- the array owns its block and
        returns it to the pool when it is garbage-collected."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        if nbytes >= _PINNED_MIN:
            out = _pinned_pool.array(shape, dtype)
            if out is not None:
                nat.check(_lib().ofl_download_async(out.ctypes.data, self.ptr, nbytes, stream))
                nat.check(_lib().ofl_stream_sync(stream))
                return out
        out = np.empty(shape, dtype)
        if out.nbytes:
            nat.check(_lib().ofl_download(out.ctypes.data, self.ptr, out.nbytes, stream))
        return out


class _BufferView:
    """Part of a DeviceBuffer (an address and a length; the buffer itself stays with its owner)."""

    __slots__ = ("ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, int(nbytes)


class PinnedArray:
    """A NumPy view of page-locked host memory (ofl_host_alloc): transfers from / to it are asynchronous DMA."""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(int(v) for v in shape), np.dtype(dtype)
        self.nbytes = max(int(np.prod(self.shape)) * self.dtype.itemsize, 16)
        p = ctypes.c_void_p()
        nat.check(_lib().ofl_host_alloc(ctypes.byref(p), self.nbytes))
        self.ptr = p.value
        self._fin = weakref.finalize(self, nat.load().ofl_host_free, self.ptr)
        buf = (ctypes.c_char * self.nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=int(np.prod(self.shape))).reshape(self.shape)


def load_sintel_device(path, ref='s', stream=None):
    """Sintel .flo (utils.py:447-470) straight onto the device: the payload is read into a pinned buffer and uploaded
    asynchronously on `stream`; returns (DeviceFlow with an all-valid mask, the PinnedArray that must stay alive until
    the stream has passed the upload).  Flow.from_sintel labels such fields 's' (flow_class.py:262-275)."""
    if not isinstance(path, str):
        raise TypeError("Error loading flow from Sintel data: Path needs to be a string")
    with open(path, 'rb') as f:
        if f.read(4) != b'PIEH':
            raise ValueError("Error loading flow from Sintel data: Path not a valid .flo file")
        w = int.from_bytes(f.read(4), 'little')
        h = int.from_bytes(f.read(4), 'little')
        pin = PinnedArray((h, w, 2), '<f4')
        got = f.readinto(memoryview(pin.array).cast('B'))
        if got != h * w * 8:
            raise ValueError("Error loading flow from Sintel data: file is truncated")
    vecs, mask = DeviceBuffer(h * w * 8), DeviceBuffer(h * w)
    nat.check(_lib().ofl_upload(vecs.ptr, pin.ptr, h * w * 8, stream))
    nat.check(_lib().ofl_memset(mask.ptr, 1, h * w, stream))
    return DeviceFlow(vecs, mask, (h, w), ref), pin


def save_sintel_device(path, dflow, stream=None):
    """DeviceFlow -> .flo through a pinned buffer (asynchronous download, one synchronisation before the write)."""
    h, w = dflow.shape
    pin = PinnedArray((h, w, 2), '<f4')
    nat.check(_lib().ofl_download_async(pin.ptr, dflow.vecs.ptr, h * w * 8, stream))
    nat.check(_lib().ofl_stream_sync(stream))
    with open(path, 'wb') as f:
        f.write(b'PIEH')
        f.write(int(w).to_bytes(4, 'little'))
        f.write(int(h).to_bytes(4, 'little'))
        f.write(memoryview(pin.array).cast('B'))


_PINNED_MIN = 1 << 20        # results below 1 MiB stay pageable


class _PinnedBlock:
    __slots__ = ("ptr", "nbytes", "__weakref__")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes


class _PinnedPool:
    """Page-locked host blocks for downloads, recycled by size (hipHostMalloc of 66 MB takes milliseconds)."""

    def __init__(self, limit=8 << 30):
        self.free, self.in_use, self.limit = {}, 0, limit

    def _give(self, ptr, nbytes):
        try:
            self.in_use -= nbytes
            self.free.setdefault(nbytes, []).append(ptr)
        except Exception:       # interpreter shutdown
            pass

    def array(self, shape, dtype):
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dtype.itemsize
        if self.in_use + nbytes > self.limit:
            return None
        lst = self.free.get(nbytes)
        if lst:
            ptr = lst.pop()
        else:
            p = ctypes.c_void_p()
            try:
                nat.check(_lib().ofl_host_alloc(ctypes.byref(p), nbytes))
            except nat.NativeError:
                return None
            ptr = p.value
        self.in_use += nbytes
        block = _PinnedBlock(ptr, nbytes)
        weakref.finalize(block, self._give, ptr, nbytes)
        buf = (ctypes.c_char * nbytes).from_address(ptr)
        buf._block = block                      # the array's base keeps the block alive
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


_pinned_pool = _PinnedPool()


def sync(stream=None):
    nat.check(_lib().ofl_stream_sync(stream))


class DeviceImage:
    """A warp target in HBM: [H][W][C] of uint8 / int16 / uint16 / float32 / float64."""

    def __init__(self, buf, shape, dtype):
        self.buf, self.shape, self.dtype = buf, tuple(shape), np.dtype(dtype)

    @classmethod
    def from_host(cls, arr):
        arr = np.ascontiguousarray(arr)
        if arr.ndim == 2:
            arr = arr[..., None]
        if arr.dtype not in _DT_CODE:
            raise TypeError("warp targets must be uint8, int16, uint16, float32 or float64 "
                            "(what cv2.remap accepts), got {}".format(arr.dtype))
        return cls(DeviceBuffer.from_host(arr), arr.shape, arr.dtype)

    def to_host(self):
        return self.buf.to_host(self.shape, self.dtype)


# ------------------------------------------------------------------------------ kernels
def gather_bilinear(src, flow_buf, flow_shape, sign, smask=None, fmask=None, want_valid=False,
                    pad=(0, 0), quant=nat.QUANT_OPENCV, arith=nat.ARITH_NATIVE, rule=nat.RULE_EQ1,
                    stream=None):
    """K1: dst = B(src; x + sign*flow) (+ validity).  Replaces apply_flow 't', utils.py:231-236."""
    H, W, C = src.shape
    dst = DeviceImage(DeviceBuffer(H * W * C * src.dtype.itemsize), src.shape, src.dtype)
    valid = DeviceBuffer(H * W) if want_valid else None
    nat.check(_lib().ofl_gather_bilinear_dev(
        src.buf.ptr, _DT_CODE[src.dtype], C, H, W, flow_buf.ptr, flow_shape[0], flow_shape[1],
        pad[0], pad[1], sign, smask.ptr if smask is not None else None,
        fmask.ptr if fmask is not None else None, dst.buf.ptr,
        valid.ptr if valid is not None else None, quant, arith, rule, stream))
    return dst, valid


def gather_bilinear_batch(src, dtype, C, H, W, batch, flow, sign, smask=None, fmask=None, dst=None, valid=None,
                          shared_src=False, shared_smask=False, flow_shape=None, pad=(0, 0),
                          quant=nat.QUANT_OPENCV, arith=nat.ARITH_NATIVE, rule=nat.RULE_EQ1, stream=None):
    """K1 over `batch` fields in ONE launch (ofl_gather_bilinear_batch_dev): buffers hold the fields back to back -- flow
    [B][fH][fW][2], fmask [B][fH][fW], src [B][H][W][C] (or one [H][W][C] image for all with shared_src), smask likewise,
    dst [B][H][W][C], valid [B][H][W].  dst / valid are allocated when not given (valid only if `valid is True`).
    Returns (dst buffer, valid buffer or None)."""
    dtype = np.dtype(dtype)
    fH, fW = flow_shape if flow_shape is not None else (H, W)
    if dst is None:
        dst = DeviceBuffer(batch * H * W * C * dtype.itemsize)
    if valid is True:
        valid = DeviceBuffer(batch * H * W)
    nat.check(_lib().ofl_gather_bilinear_batch_dev(
        src.ptr if src is not None else None, 1 if shared_src else 0, _DT_CODE[dtype], C, H, W, batch, flow.ptr, fH, fW,
        pad[0], pad[1], sign, smask.ptr if smask is not None else None, 1 if shared_smask else 0,
        fmask.ptr if fmask is not None else None, dst.ptr if C else None, valid.ptr if valid is not None else None,
        quant, arith, rule, stream))
    return dst, valid


def gather_rows(src, row0, rows, flow_rows, sign, smask=None, fmask_rows=None, want_valid=False,
                quant=nat.QUANT_OPENCV, arith=nat.ARITH_NATIVE, rule=nat.RULE_EQ1, stream=None):
    """K1 on one row band of a field split over several GPUs: `src` is the replicated
    H x W image, `flow_rows` / `fmask_rows` hold rows [row0, row0 + rows) only; returns the same rows of the
    warped image (and of the valid area)."""
    H, W, C = src.shape
    dst = DeviceImage(DeviceBuffer(rows * W * C * src.dtype.itemsize), (rows, W, C), src.dtype)
    valid = DeviceBuffer(rows * W) if want_valid else None
    nat.check(_lib().ofl_gather_rows_dev(
        src.buf.ptr, _DT_CODE[src.dtype], C, H, W, row0, rows, flow_rows.ptr, sign,
        smask.ptr if smask is not None else None, fmask_rows.ptr if fmask_rows is not None else None,
        dst.buf.ptr, valid.ptr if valid is not None else None, quant, arith, rule, stream))
    return dst, valid


def gather_valid_only(H, W, flow_buf, flow_shape, sign, smask=None, fmask=None, pad=(0, 0),
                      quant=nat.QUANT_OPENCV, rule=nat.RULE_EQ1, stream=None):
    """K1 without image channels: where does a warped all-ones (or smask) image stay == 1?
    (valid_target 't' flow_class.py:1148-1150, valid_source 's' :1179-1183)."""
    valid = DeviceBuffer(H * W)
    nat.check(_lib().ofl_gather_bilinear_dev(
        None, nat.U8, 0, H, W, flow_buf.ptr, flow_shape[0], flow_shape[1], pad[0], pad[1], sign,
        smask.ptr if smask is not None else None, fmask.ptr if fmask is not None else None,
        None, valid.ptr, quant, nat.ARITH_NATIVE, rule, stream))
    return valid


def flow_stats(vecs_buf, mask_buf, n_px, stream=None):
    """K4: OFL_STAT_* bits of one field (utils.py:527-544, flow_class.py:1230-1245)."""
    out = DeviceBuffer(16)
    nat.check(_lib().ofl_flow_stats_dev(vecs_buf.ptr, mask_buf.ptr if mask_buf is not None else None,
                                        n_px, np.float32(DEFAULT_THRESHOLD), out.ptr, stream))
    return int(out.to_host((1,), np.uint32, stream)[0])


def compose3_launch(fa, fb, sign, out, stats_buf=None, stats_offset=0, batch=1, quant=nat.QUANT_OPENCV,
                    stream=None):
    """K2 launch on raw DeviceFlow-like triples; asynchronous.  stats_buf: uint32[batch][8] words."""
    H, W = fa.shape
    sp = None if stats_buf is None else stats_buf.ptr + stats_offset
    nat.check(_lib().ofl_compose3_dev(fa.vecs.ptr, fa.mask.ptr, fb.vecs.ptr, fb.mask.ptr, sign, H, W, batch,
                                      out.vecs.ptr, out.mask.ptr, sp, quant, stream))


def mask_bits_bytes(h, w, batch=1):
    n = ctypes.c_size_t(0)
    nat.check(_lib().ofl_mask_bits_bytes(h, w, batch, ctypes.byref(n)))
    return n.value


def mask_pack(mask_buf, h, w, batch=1, stream=None):
    """uint8 masks [batch][H][W] -> packed bit planes [batch][H][(W + 31) / 32] uint32 (ofl_mask_pack_dev)"""
    bits = DeviceBuffer(mask_bits_bytes(h, w, batch))
    nat.check(_lib().ofl_mask_pack_dev(mask_buf.ptr, h, w, batch, bits.ptr, stream))
    return bits


def mask_unpack(bits_buf, h, w, batch=1, stream=None):
    mask = DeviceBuffer(batch * h * w)
    nat.check(_lib().ofl_mask_unpack_dev(bits_buf.ptr, h, w, batch, mask.ptr, stream))
    return mask


def compose3_bits_launch(fa_vecs, fa_bits, fb_vecs, fb_bits, sign, shape, out_vecs, out_bits, stats_buf=None, stats_offset=0, batch=1, stream=None):
    """K2 on packed mask planes (ofl_compose3_bits_dev); asynchronous."""
    sp = None if stats_buf is None else stats_buf.ptr + stats_offset
    nat.check(_lib().ofl_compose3_bits_dev(fa_vecs.ptr, fa_bits.ptr, fb_vecs.ptr, fb_bits.ptr, sign, shape[0], shape[1], batch,
                                           out_vecs.ptr, out_bits.ptr, sp, stream))


_STATS_KNOW_MASK = 1 << 30        # private flag in DeviceFlow._stats: STAT_MASK_HAS_ZERO has been evaluated


# ------------------------------------------------------------------------------ DeviceFlow
class DeviceFlow:
    """(vecs, mask, ref) resident in HBM.  Buffers are immutable once wrapped."""

    def __init__(self, vecs, mask, shape, ref, stats=None):
        self.vecs, self.mask = vecs, mask
        self.shape = (int(shape[0]), int(shape[1]))
        self.ref = ref
        self._stats = stats
        self._certs = {}            # (sign, point_precision) -> MeshCert of the warped grid without a point mask

    # -- construction / transfer
    @classmethod
    def empty(cls, shape, ref):
        n = int(shape[0]) * int(shape[1])
        return cls(DeviceBuffer(n * 8), DeviceBuffer(n), shape, ref)

    @classmethod
    def from_host(cls, vecs, ref='t', mask=None):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32)
        h, w = vecs.shape[:2]
        if mask is None:
            m = np.ones((h, w), np.uint8)
        else:
            m = np.ascontiguousarray(mask)
            m = m.view(np.uint8) if m.dtype == np.bool_ else m.astype(np.uint8)
        return cls(DeviceBuffer.from_host(vecs), DeviceBuffer.from_host(m), (h, w), ref)

    def to_host(self):
        """-> (vecs float32 [H,W,2], mask bool [H,W])"""
        h, w = self.shape
        return self.vecs.to_host((h, w, 2), np.float32), self.mask.to_host((h, w), np.uint8).view(np.bool_)    # bytes are 0 / 1

    def relabel(self, ref):
        out = DeviceFlow(self.vecs, self.mask, self.shape, ref, self._stats)
        out._certs = self._certs        # same vectors: same warped grid
        return out

    def mesh_cert(self, sign, point_precision=0):
        """Certificate of the grid warped by sign * vecs with every point kept (ofl_scatter_certify_dev): evaluated
        once per field and sign, then the certified scatter entry runs without any synchronisation."""
        key = (sign, point_precision)
        c = self._certs.get(key)
        if c is None:
            h, w = self.shape
            c = nat.MeshCert()
            ws = _workspace(h, w, 0)
            nb = ctypes.c_size_t(0)
            nat.check(_lib().ofl_scatter_diag_bytes(h, w, ctypes.byref(nb)))
            bits = DeviceBuffer(nb.value)          # the cells' Delaunay diagonals, read by the walk kernel; lives with the certificate
            nat.check(_lib().ofl_scatter_certify_dev(self.vecs.ptr, sign, point_precision, None, h, w, ws.ptr, ws.nbytes,
                                                     ctypes.byref(c), bits.ptr, None))
            if c.certified:
                c._diag_buf = bits
            else:
                c.diag_bits = None
            self._certs[key] = c
        return c

    @property
    def n_px(self):
        return self.shape[0] * self.shape[1]

    # -- predicates
    def stats(self):
        if self._stats is None:
            self._stats = flow_stats(self.vecs, self.mask, self.n_px) | _STATS_KNOW_MASK
        return self._stats

    def is_zero(self, thresholded=True, masked=True):
        """Flow.is_zero, flow_class.py:1230-1245."""
        s = self.stats()
        if masked:
            bit = nat.STAT_NONZERO_TH_MASKED if thresholded else nat.STAT_NONZERO_MASKED
        else:
            bit = nat.STAT_NONZERO_TH if thresholded else nat.STAT_NONZERO
        return not (s & bit)

    def get_padding(self):
        """Flow.get_padding (flow_class.py:1197-1228): one min/max reduction over the masked sampling positions."""
        h, w = self.shape
        ext = DeviceBuffer(16)
        nat.check(_lib().ofl_flow_extent_dev(self.vecs.ptr, self.mask.ptr, h, w, -1 if self.ref == 't' else 1,
                                             np.float32(DEFAULT_THRESHOLD), ext.ptr, None))
        min_y, max_y, min_x, max_x = (float(v) for v in ext.to_host((4,), np.float32))
        if not np.isfinite(min_y):
            raise ValueError("zero-size array to reduction operation minimum which has no identity")   # NumPy's error in the reference
        pads = [max(-min_y, 0), max(max_y - (h - 1), 0), max(-min_x, 0), max(max_x - (w - 1), 0)]
        return [int(np.ceil(p)) for p in pads]

    # -- element-wise algebra (Flow.__add__/__sub__/__neg__, flow_class.py:310-375, 479-489)
    def _axpy(self, other, alpha):
        out = DeviceFlow.empty(self.shape, self.ref)
        nat.check(_lib().ofl_axpy_dev(self.vecs.ptr, self.mask.ptr,
                                      other.vecs.ptr if other is not None else None,
                                      other.mask.ptr if other is not None else None,
                                      np.float32(alpha), self.n_px, out.vecs.ptr, out.mask.ptr, None))
        return out

    def __add__(self, other):
        return self._axpy(other, 1.0)

    def __sub__(self, other):
        return self._axpy(other, -1.0)

    def __neg__(self):
        out = self._axpy(None, -1.0)
        out._stats = self._stats            # the zero-flow predicates do not change under negation
        out._certs = {(-sg, pp): c for (sg, pp), c in self._certs.items()}      # x - (-f) is x + f: same warped grid
        return out

    def _compose(self, sampled, sign, quant=nat.QUANT_OPENCV):
        """self + B(sampled; x + sign * self) with the masks of a warp followed by an addition: ONE launch of the
        fused compose kernel.  This is `g + g.apply(sampled)` for a 't'-reference g (sign -1) and
        `a + (-a as 't').apply(sampled)` for an 's'-reference a (sign +1) -- the expressions inside mode 1
        (flow_class.py:1369-1370, 1383-1385) -- bit for bit."""
        out = DeviceFlow.empty(self.shape, self.ref)
        compose3_launch(sampled, self, sign, out, quant=quant)
        return out

    # -- warping
    def apply(self, target, consider_mask=True, quant=nat.QUANT_OPENCV, target_mask=None):
        """Flow.apply with a Flow target (flow_class.py:600-603, 632-684): the target's vectors and mask
        are warped together; the result keeps the TARGET's reference.  A `DeviceImage` target returns
        (warped DeviceImage, valid-area DeviceBuffer) -- see apply_image."""
        if isinstance(target, DeviceImage):
            return self.apply_image(target, target_mask, consider_mask, quant)
        if self.ref == 't':
            # utils.py:215-216: a (thresholded-)zero flow returns the target itself.  Under cv2's 1/32-px coordinate
            # snapping the gather of such a flow IS the identity (|v| < 1e-3 snaps to 0), so the test -- a
            # reduction and a host synchronisation -- is only needed for the un-snapped extension mode
            if quant != nat.QUANT_OPENCV and self.is_zero(thresholded=True, masked=False):
                return target._and_mask(self)
            src = DeviceImage(target.vecs, self.shape + (2,), np.float32)
            dst, valid = gather_bilinear(src, self.vecs, self.shape, -1, smask=target.mask, fmask=self.mask,
                                         want_valid=True, quant=quant)
            return DeviceFlow(dst.buf, valid, self.shape, target.ref)
        return self._scatter_flow(target, consider_mask)

    def apply_image(self, image, target_mask=None, consider_mask=True, quant=nat.QUANT_OPENCV):
        """Warp an HBM-resident image [H][W][C] and propagate validity (Flow.apply with an ndarray target and
        return_valid_area=True, flow_class.py:604-695, without padding).  't': any remap dtype, one launch of
        the gather kernel; 's': float32 images through the scatter kernel.  `target_mask`: uint8 DeviceBuffer
        or None (all valid)."""
        h, w = self.shape
        if image.shape[:2] != (h, w):
            raise ValueError("image and flow need the same height and width")
        if self.ref == 't':
            if self.is_zero(thresholded=True, masked=False):        # identity short cut, utils.py:215-216
                valid = DeviceBuffer(self.n_px)
                if target_mask is None:
                    nat.check(_lib().ofl_copy_dev(valid.ptr, self.mask.ptr, self.n_px, None))
                else:
                    _mask_and(self.mask, target_mask, valid, self.n_px)
                return image, valid
            arith, rule = nat.ARITH_NATIVE, nat.RULE_EQ1
            if image.dtype == np.uint8:      # concat dtype of the reference: bool mask -> uint8, default int8 -> int16
                arith, rule = (nat.ARITH_NATIVE, nat.RULE_GE_HALF) if target_mask is not None else (nat.ARITH_FLOAT_RNE, nat.RULE_GT_HALF)
            elif image.dtype == np.int16 or (image.dtype == np.uint16 and target_mask is not None):
                rule = nat.RULE_GT_HALF
            elif image.dtype == np.uint16:
                raise TypeError("uint16 image with the default int8 mask needs an int32 remap, which cv2.remap does not provide")
            return gather_bilinear(image, self.vecs, self.shape, -1, smask=target_mask, fmask=self.mask,
                                   want_valid=True, quant=quant, arith=arith, rule=rule)
        if image.dtype != np.float32:
            raise TypeError("'s'-reference warps of device images need float32 (got {})".format(image.dtype))
        C = image.shape[2]
        vmask = self.mask
        if target_mask is not None:
            vmask = DeviceBuffer(self.n_px)
            _mask_and(target_mask, self.mask, vmask, self.n_px)                  # flow_class.py:643
        if self.is_zero(thresholded=True, masked=False):
            return image, vmask
        out = DeviceImage(DeviceBuffer(self.n_px * C * 4), image.shape, np.float32)
        valid = DeviceBuffer(self.n_px)
        pm = self._point_mask(consider_mask)
        scatter_linear(self.vecs, +1, pm, image.buf, C, vmask, h, w, None, out.buf, valid, 0,
                       cert=self.mesh_cert(+1) if pm is None else None, drops_points=pm is not None)
        return out, valid

    def apply_image_rows(self, image, rank, world, target_mask=None, consider_mask=True, quant=nat.QUANT_OPENCV,
                         gather=None, align=8):
        """This rank's ROW BAND of apply_image when ONE field is split over `world` GPUs (BASELINE config 5):
        flow, masks and image are the replicated H x W arrays, the result holds rows sharding.row_band(H, rank, world, align)
        only -- (DeviceImage rows x W x C, valid rows x W, (row0, row1)); the bands of all ranks concatenate to
        apply_image's result bit for bit.  't': one launch of the gather kernel on the band (nothing is exchanged).  's':
        a field whose mesh certifies resolves its rows in one kernel (nothing is exchanged); any other field takes the
        slab-wise Delaunay path with its one all-gather (`gather`: see scatter_slab; default RCCL)."""
        from .sharding import row_band
        h, w = self.shape
        if image.shape[:2] != (h, w):
            raise ValueError("image and flow need the same height and width")
        r0, r1 = row_band(h, rank, world, align)
        rows = r1 - r0
        C = image.shape[2]
        at = lambda buf, off, n: _BufferView(buf.ptr + off, n)
        if self.ref == 't':
            if rows <= 0:                       # more ranks than 8-row tiles: nothing to compute, and 't' exchanges nothing
                return None, None, (r0, r1)
            arith, rule = nat.ARITH_NATIVE, nat.RULE_EQ1
            if image.dtype == np.uint8:
                arith, rule = (nat.ARITH_NATIVE, nat.RULE_GE_HALF) if target_mask is not None else (nat.ARITH_FLOAT_RNE, nat.RULE_GT_HALF)
            elif image.dtype == np.int16 or (image.dtype == np.uint16 and target_mask is not None):
                rule = nat.RULE_GT_HALF
            elif image.dtype == np.uint16:
                raise TypeError("uint16 image with the default int8 mask needs an int32 remap, which cv2.remap does not provide")
            if self.is_zero(thresholded=True, masked=False):        # identity short cut, utils.py:215-216
                nb = rows * w * C * image.dtype.itemsize
                dst = DeviceImage(DeviceBuffer(nb), (rows, w, C), image.dtype)
                nat.check(_lib().ofl_copy_dev(dst.buf.ptr, image.buf.ptr + r0 * w * C * image.dtype.itemsize, nb, None))
                valid = DeviceBuffer(rows * w)
                if target_mask is None:
                    nat.check(_lib().ofl_copy_dev(valid.ptr, self.mask.ptr + r0 * w, rows * w, None))
                else:
                    _mask_and(at(self.mask, r0 * w, rows * w), at(target_mask, r0 * w, rows * w), valid, rows * w)
                return dst, valid, (r0, r1)
            dst, valid = gather_rows(image, r0, rows, at(self.vecs, r0 * w * 8, rows * w * 8), -1, smask=target_mask,
                                     fmask_rows=at(self.mask, r0 * w, rows * w), want_valid=True, quant=quant, arith=arith, rule=rule)
            return dst, valid, (r0, r1)
        if image.dtype != np.float32:
            raise TypeError("'s'-reference warps of device images need float32 (got {})".format(image.dtype))
        vmask = self.mask
        if target_mask is not None:
            vmask = DeviceBuffer(self.n_px)
            _mask_and(target_mask, self.mask, vmask, self.n_px)                  # flow_class.py:643
        if self.is_zero(thresholded=True, masked=False):
            if rows <= 0:
                return None, None, (r0, r1)
            nb = rows * w * C * 4
            dst = DeviceImage(DeviceBuffer(nb), (rows, w, C), np.float32)
            nat.check(_lib().ofl_copy_dev(dst.buf.ptr, image.buf.ptr + r0 * w * C * 4, nb, None))
            valid = DeviceBuffer(rows * w)
            nat.check(_lib().ofl_copy_dev(valid.ptr, vmask.ptr + r0 * w, rows * w, None))
            return dst, valid, (r0, r1)
        # (a rank with an EMPTY band -- more ranks than 8-row tiles -- still walks through the path decision below: the
        # slab-wise path has two all-gathers that every rank of `world` must join, scatter_slab with rows == 0)
        out = DeviceImage(DeviceBuffer(max(rows, 0) * w * C * 4), (max(rows, 0), w, C), np.float32)
        valid = DeviceBuffer(max(rows, 0) * w)
        pm = self._point_mask(consider_mask)
        cert = self.mesh_cert(+1) if pm is None else None
        if cert is not None and cert.certified and not getattr(cert, "_walk_checked", False):
            # Does the walk kernel find every node of this certified mesh (scatter_linear)?  The answer must be the SAME on
            # every rank -- a rank that went on alone to the slab-wise path would wait for the others in its all-gather --
            # so each rank asks for the whole field once per certificate (validity only: 0.1 ms at 4K), not for its band.
            cnt, scratch = DeviceBuffer.zeros(16), DeviceBuffer(self.n_px)
            nat.check(_lib().ofl_scatter_certified_dev(self.vecs.ptr, +1, 0, None, 0, vmask.ptr, h, w, 0, h,
                                                       None, scratch.ptr, 0, ctypes.byref(cert), cnt.ptr, None))
            if int(cnt.to_host((1,), np.uint32)[0]) == 0:
                cert._walk_checked = True
            else:
                cert.certified = 0
        if cert is not None and cert.certified:
            if rows <= 0:
                return None, None, (r0, r1)
            nat.check(_lib().ofl_scatter_certified_dev(self.vecs.ptr, +1, 0, image.buf.ptr, C, vmask.ptr, h, w, r0, rows,
                                                       out.buf.ptr, valid.ptr, 0, ctypes.byref(cert), None, None))
            return out, valid, (r0, r1)
        if world <= 1:
            scatter_linear(self.vecs, +1, pm, image.buf, C, vmask, h, w, None, out.buf, valid, nat.SCATTER_UNCERTIFIED)
        else:
            scatter_slab(self.vecs, +1, pm, image.buf, C, vmask, h, w, r0, rows, out.buf, valid, rank, world,
                         gather=gather if gather is not None else comm_allgather)
            if rows <= 0:
                return None, None, (r0, r1)
        return out, valid, (r0, r1)

    def resize(self, scale):
        """Flow.resize (flow_class.py:491-506) on HBM-resident data: one launch of the resize kernel."""
        fy, fx = resize_scales(scale)
        h, w = self.shape
        ho, wo = resized_shape(h, w, fy, fx)
        out = DeviceFlow.empty((ho, wo), self.ref)
        nat.check(_lib().ofl_resize_flow_dev(self.vecs.ptr, self.mask.ptr, h, w, ho, wo, 1.0 / fy, 1.0 / fx,
                                             float(np.float32(fx)), float(np.float32(fy)), out.vecs.ptr, out.mask.ptr, None))
        return out

    def _and_mask(self, other):
        """vecs unchanged, mask = self.mask & other.mask (zero-flow identity warp of a Flow target)."""
        out_mask = DeviceBuffer(self.n_px)
        scratch = DeviceBuffer(self.n_px * 8)
        nat.check(_lib().ofl_axpy_dev(self.vecs.ptr, self.mask.ptr, None, other.mask.ptr, np.float32(1.0),
                                      self.n_px, scratch.ptr, out_mask.ptr, None))
        return DeviceFlow(scratch, out_mask, self.shape, self.ref)

    def _point_mask(self, consider_mask=True):
        """The point mask of utils.py:249-251 -- None when every point is kept (same result, and the scatter kernel then
        skips its per-pixel search for dropped neighbours).  Only statistics that come from ofl_flow_stats know the
        mask; predicates taken over from the compose kernel do not."""
        if not consider_mask:
            return None
        if self._stats is None or not (self._stats & _STATS_KNOW_MASK):
            self._stats = flow_stats(self.vecs, self.mask, self.n_px) | _STATS_KNOW_MASK
        return self.mask if (self._stats & nat.STAT_MASK_HAS_ZERO) else None

    def _scatter_flow(self, target, consider_mask=True, sign=1, negate=False):
        """'s'-reference apply of a Flow target: utils.py:237-258 + flow_class.py:634-643, 668.  negate: the target is
        -`target` (its vectors are negated inside the kernel, OFL_SCATTER_NEGATE)."""
        if self.is_zero(thresholded=True, masked=False):
            return (-target if negate else target)._and_mask(self)
        h, w = self.shape
        out = DeviceFlow.empty(self.shape, target.ref)
        if target.mask is self.mask:
            vmask = self.mask                                     # m & m
        else:
            vmask = DeviceBuffer(self.n_px)
            _mask_and(target.mask, self.mask, vmask, self.n_px)  # mask channel, flow_class.py:643
        pm = self._point_mask(consider_mask)
        scatter_linear(self.vecs, sign, pm, target.vecs, 2, vmask, h, w, None, out.vecs, out.mask,
                       nat.SCATTER_NEGATE if negate else 0, cert=self.mesh_cert(sign) if pm is None else None, drops_points=pm is not None)
        return out

    def switch_ref(self):
        """Flow.switch_ref(mode='valid'), flow_class.py:716-726."""
        other = 't' if self.ref == 's' else 's'
        if self.is_zero(thresholded=False):
            return self.relabel(other)
        if self.ref == 's':
            return self.apply(self).relabel('t')
        as_s = self.relabel('s')
        return (-as_s).apply(as_s)

    def invert(self, ref=None):
        """Flow.invert, flow_class.py:735-753."""
        ref = self.ref if ref is None else ref
        if self.ref == 's':
            # s -> s: self.apply(-self) (flow_class.py:746) -- the negation happens inside the scatter kernel
            return self._scatter_flow(self, negate=True) if ref == 's' else (-self).relabel('t')
        if ref == 's':
            return (-self).relabel('s')
        return self.invert('s').switch_ref()

    def combine_with(self, flow, mode, thresholded=False, quant=nat.QUANT_OPENCV):
        """Flow.combine_with, flow_class.py:1338-1424.  Returns `self` / `flow` themselves on the
        reference's zero-flow early exits."""
        if mode == 3:
            return self._combine3(flow, thresholded, quant)
        if self.is_zero(thresholded=thresholded):
            return flow
        if flow.is_zero(thresholded=thresholded):
            return self.invert()
        s = self.ref == 's'
        if mode == 1:
            if s:                                                            # flow_class.py:1369-1370
                g = flow.invert('t')
                return flow - g._compose(self.switch_ref(), -1, quant).apply(self)       # g + g.apply(..), fused
            a = self.switch_ref()                                            # flow_class.py:1383-1385
            res = flow.switch_ref() - a._compose(flow.invert('s'), +1, quant).apply(a)  # a + (-a).apply(..), fused
            return res.switch_ref()
        if s:
            return self.apply(flow - self)                                   # flow_class.py:1390
        return flow - self._resample_to(flow)                                # flow_class.py:1398-1410

    def _combine3(self, flow, thresholded, quant):
        """Mode 3 through the fused kernel K2.  The early-exit predicates of flow_class.py:1339-1354 and
        utils.py:215 are evaluated by the same launch (stat words) and honoured afterwards."""
        if self.ref == 's':
            fa, fb, sign = flow, self, +1          # self + self.invert('t').apply(flow)   (:1418)
        else:
            fa, fb, sign = self, flow, -1          # flow + flow.apply(self)               (:1422)
        out = DeviceFlow.empty(self.shape, self.ref)
        need = fa._stats is None or fb._stats is None
        words = DeviceBuffer.zeros(32) if need else None
        compose3_launch(fa, fb, sign, out, words, quant=quant)
        cert_a = 0
        if need:
            w = words.to_host((8,), np.uint32)
            if fb._stats is None:                       # exact predicates of the streamed field
                fb._stats = sum((1 << k) for k in range(4) if w[4 + k])
            cert_a = (1 if w[0] else 0) | (2 if w[1] else 0)   # certificates "fa is not zero"
        bit = nat.STAT_NONZERO_TH_MASKED if thresholded else nat.STAT_NONZERO_MASKED
        if fa._stats is None and not (cert_a & bit):
            fa.stats()                                  # not certified by the gather: exact pass (K4)
        a_zero = fa.is_zero(thresholded=thresholded) if fa._stats is not None else False
        b_zero = fb.is_zero(thresholded=thresholded)
        self_zero, flow_zero = (b_zero, a_zero) if self.ref == 's' else (a_zero, b_zero)
        if self_zero:
            return flow
        if flow_zero:
            return self
        if fb.is_zero(thresholded=True, masked=False):
            # apply_flow returned the target untouched (utils.py:215-216): plain vector sum
            return fb + fa
        return out

    def _resample_to(self, flow3):
        """Mode 2 / ref 't' (flow_class.py:1398-1410): self sampled from the float32 points x - self onto
        the points x - flow3 (no mask filtering); mask = interpolated mask > 0.99."""
        h, w = self.shape
        out = DeviceFlow.empty(self.shape, 't')
        query = DeviceBuffer(self.n_px * 8)
        grid_minus(flow3.vecs, query, h, w)
        scatter_linear(self.vecs, -1, None, self.vecs, 2, self.mask, h, w, query, out.vecs, out.mask, 1,
                       point_precision=1)
        return out

    def valid_target(self, consider_mask=True, quant=nat.QUANT_OPENCV):
        """flow_class.py:1113-1151 -> uint8 mask buffer."""
        h, w = self.shape
        if self.ref == 't':
            if self.is_zero(thresholded=True, masked=False):
                return self.mask
            return gather_valid_only(h, w, self.vecs, self.shape, -1, fmask=self.mask, quant=quant)
        return self._scatter_mask(+1, consider_mask)

    def valid_source(self, consider_mask=True, quant=nat.QUANT_OPENCV):
        """flow_class.py:1153-1195 -> uint8 mask buffer."""
        h, w = self.shape
        if self.ref == 's':
            if self.is_zero(thresholded=True, masked=False):
                return self.mask
            return gather_valid_only(h, w, self.vecs, self.shape, +1, fmask=self.mask, quant=quant)
        return self._scatter_mask(-1, consider_mask)

    def _scatter_mask(self, sign, consider_mask):
        """apply_flow(+-vecs, mask.astype('f'), 's', mask|None) == 1  (flow_class.py:1140-1141, 1190-1193)."""
        if self.is_zero(thresholded=True, masked=False):
            return self.mask
        h, w = self.shape
        valid = DeviceBuffer(self.n_px)
        pm = self._point_mask(consider_mask)
        scatter_linear(self.vecs, sign, pm, None, 0, self.mask, h, w, None, None, valid, 0,
                       cert=self.mesh_cert(sign) if pm is None else None, drops_points=pm is not None)
        return valid


# ------------------------------------------------------------------------------ small helpers
def _mask_and(a, b, out, n):
    """out = a & b for uint8 masks (flow_class.py:643)."""
    nat.check(_lib().ofl_mask_and_dev(a.ptr, b.ptr, out.ptr, n, None))


def resize_scales(scale, error_string="Error resizing flow: "):
    """Validation of resize_flow's `scale` (utils.py:505-518): returns (vertical, horizontal) factors."""
    if isinstance(scale, (float, int)):
        scale = [scale, scale]
    elif isinstance(scale, (tuple, list)):
        if len(scale) != 2:
            raise ValueError(error_string + "Scale {} must have a length of 2".format(type(scale)))
        if not all(isinstance(item, (float, int)) for item in scale):
            raise ValueError(error_string + "Scale {} items must be integers or floats".format(type(scale)))
    else:
        raise TypeError(error_string + "Scale must be an integer, float, or list or tuple of integers or floats")
    if any(s <= 0 for s in scale):
        raise ValueError(error_string + "Scale values must be larger than 0")
    return float(scale[0]), float(scale[1])


def resized_shape(h, w, fy, fx):
    """cv2.resize(dsize=None, fx, fy): dsize = (cvRound(W * fx), cvRound(H * fy)), round half to even."""
    ho, wo = int(np.rint(h * fy)), int(np.rint(w * fx))
    if ho <= 0 or wo <= 0:
        raise ValueError("Error resizing flow: scale {} leaves no pixels of a {}x{} field".format((fy, fx), h, w))
    return ho, wo


def resize_host(vecs, mask, scale):
    """resize_flow / Flow.resize for host arrays through ofl_resize_flow (upload, one launch, download)."""
    fy, fx = resize_scales(scale)
    vecs = np.ascontiguousarray(vecs, np.float32)
    h, w = vecs.shape[:2]
    ho, wo = resized_shape(h, w, fy, fx)
    out = np.empty((ho, wo, 2), np.float32)
    m = None if mask is None else np.ascontiguousarray(mask).astype(np.uint8)
    mout = None if mask is None else np.empty((ho, wo), np.uint8)
    hp = lambda a: None if a is None else a.ctypes.data
    nat.check(_lib().ofl_resize_flow(hp(vecs), hp(m), h, w, ho, wo, 1.0 / fy, 1.0 / fx,
                                     float(np.float32(fx)), float(np.float32(fy)), hp(out), hp(mout)))
    return out, (None if mout is None else mout.astype(bool))


def grid_minus(vecs, out, h, w):
    """out = float32(grid - vecs): the query positions of mode 2 / ref 't' (flow_class.py:1404-1406)."""
    nat.check(_lib().ofl_grid_offset_dev(vecs.ptr, -1, h, w, out.ptr, None))


_ws_cache = {}


def _workspace(h, w, C, stream=None):
    """Scatter workspace for fields of this shape -- one per (shape, stream): calls on different streams may overlap, and
    each then needs bucket lists and an owner map of its own."""
    key = (h, w, getattr(stream, "value", stream) or 0)
    ws = _ws_cache.get(key)
    if ws is None:
        n = ctypes.c_size_t(0)
        nat.check(_lib().ofl_scatter_workspace_bytes(h, w, C, ctypes.byref(n)))
        if len(_ws_cache) > 6:
            _ws_cache.clear()
        ws = _ws_cache[key] = DeviceBuffer(n.value)
    return ws


def scatter_linear(flow, sign, pmask, vals, C, vmask, h, w, query, out, valid, valid_rule, point_precision=0,
                   stream=None, cert=None, drops_points=False):
    """K3: scattered -> regular-grid linear interpolation.  Replaces utils.py:237-258 (and, with `query`,
    flow_class.py:1398-1410).  Raises ValueError("No points given") like qhull when nothing is kept.
    cert: a MeshCert of this very (flow, sign, point_precision) without point mask; when it certifies the mesh the
    asynchronous one-kernel entry is taken (no workspace, no read-back).  drops_points: pmask is KNOWN to hold zeros
    (the flow's statistics say so); like a certificate that says "not certified" this spares the entry its own certificate
    pass (OFL_SCATTER_UNCERTIFIED)."""
    ptr = lambda b: b.ptr if b is not None else None
    if cert is not None and cert.certified and pmask is None and query is None:
        # The certificate says the mesh IS the triangulation; that the walk kernel also FINDS every node in it is checked on
        # the first launch with this certificate (a device counter, one read-back): which nodes it locates depends on the
        # field and the sign only, so later launches with the same cached certificate run without any synchronisation.
        checked = getattr(cert, "_walk_checked", False)
        cnt = None if checked else DeviceBuffer.zeros(16, stream)
        nat.check(_lib().ofl_scatter_certified_dev(flow.ptr, sign, point_precision, ptr(vals), C, ptr(vmask), h, w, 0, h,
                                                   ptr(out), ptr(valid), valid_rule, ctypes.byref(cert), ptr(cnt), stream))
        if checked or int(cnt.to_host((1,), np.uint32, stream)[0]) == 0:
            cert._walk_checked = True
            return (h * w, 0, 0)
        cert.certified = 0                                  # nodes were lost: this field takes the Delaunay path from now on
    if (cert is not None and not cert.certified and pmask is None) or (drops_points and pmask is not None):
        valid_rule |= nat.SCATTER_UNCERTIFIED          # the certificate pass has been run for this field: not again per call
    ws = _workspace(h, w, C, stream)
    info = (ctypes.c_uint64 * 3)()
    nat.check(_lib().ofl_scatter_linear_dev(flow.ptr, sign, point_precision, ptr(pmask), ptr(vals), C, ptr(vmask),
                                            h, w, ptr(query), ptr(out), ptr(valid), valid_rule, ws.ptr, ws.nbytes,
                                            info, stream))
    return tuple(info)


def scatter_linear_f64(flow, sign, pmask, vals, C, vmask, h, w, out, valid, valid_rule, point_precision=0, stream=None):
    """K3 with float64 values at the grid nodes (float64 targets of apply_flow 's', utils.py:253-258)."""
    ws = _workspace(h, w, C, stream)
    info = (ctypes.c_uint64 * 3)()
    ptr = lambda b: b.ptr if b is not None else None
    nat.check(_lib().ofl_scatter_linear_f64_dev(flow.ptr, sign, point_precision, ptr(pmask), ptr(vals), C, ptr(vmask),
                                                h, w, ptr(out), ptr(valid), valid_rule, ws.ptr, ws.nbytes, info, stream))
    return tuple(info)


def scatter_rows(flow, sign, pmask, vals, C, vmask, h, w, row0, rows, out_rows, valid_rows, valid_rule=0,
                 point_precision=0, stream=None):
    """K3 on one row band of a field split over several GPUs (SURVEY 8e, config 5 as loaded): all inputs are the
    replicated H x W arrays; only rows [row0, row0 + rows) of the result are produced."""
    ws = _workspace(h, w, C, stream)
    info = (ctypes.c_uint64 * 3)()
    ptr = lambda b: b.ptr if b is not None else None
    nat.check(_lib().ofl_scatter_rows_dev(flow.ptr, sign, point_precision, ptr(pmask), ptr(vals), C, ptr(vmask),
                                          h, w, row0, rows, ptr(out_rows), ptr(valid_rows), valid_rule, ws.ptr, ws.nbytes,
                                          info, stream))
    return tuple(info)


SLAB_LIST_HEAD = 16          # bytes before the first record of a slab list (entries, error bits, 0, 0)
SLAB_RECORD = 64             # bytes per unfinished site


def slab_list_bytes(entries):
    return SLAB_LIST_HEAD + SLAB_RECORD * int(entries)


def comm_allgather(send_ptr, recv, nbytes, stream=None):
    """ncclAllGather of `nbytes` per rank over the live communicator (send may be the rank's own slot of recv)."""
    nat.check(_lib().ofl_comm_allgather(send_ptr, recv.ptr, nbytes, stream))


def scatter_slab_stars(flow, sign, pmask, h, w, row0, rows, list_ptr, list_bytes, point_precision=0, stream=None, ws=None):
    """Step 1 of the slab-wise scatter (include/ofl.h, ofl_scatter_slab_stars_dev): bins, the stars around rows
    [row0, row0 + rows) and -- at list_ptr (device) -- the unfinished sites of those rows.  The workspace keeps the star
    state for scatter_slab_finish: `ws` (a DeviceBuffer the caller holds on to across both steps, as scatter_slab does) or
    the cached one of this (shape, stream) -- then no other scatter call of that shape on that stream in between, and
    nothing that makes the cache drop it (step 2 refuses a workspace without step 1's stamp)."""
    ws = ws if ws is not None else _workspace(h, w, 0, stream)
    nat.check(_lib().ofl_scatter_slab_stars_dev(flow.ptr, sign, point_precision, pmask.ptr if pmask is not None else None,
                                                h, w, row0, rows, list_ptr, list_bytes, ws.ptr, ws.nbytes, stream))


def scatter_slab_finish(flow, sign, vals, C, vmask, h, w, row0, rows, lists, list_bytes, n_lists, out_rows, valid_rows,
                        valid_rule=0, point_precision=0, stream=None, ws=None):
    """Step 2: the gathered lists of all ranks -> unfinished stars, owner map and result of the band."""
    ws = ws if ws is not None else _workspace(h, w, 0, stream)
    info = (ctypes.c_uint64 * 3)()
    ptr = lambda b: b.ptr if b is not None else None
    nat.check(_lib().ofl_scatter_slab_finish_dev(flow.ptr, sign, point_precision, ptr(vals), C, ptr(vmask), h, w, row0, rows,
                                                 lists.ptr, list_bytes, n_lists, ptr(out_rows), ptr(valid_rows), valid_rule,
                                                 ws.ptr, ws.nbytes, info, stream))
    return tuple(info)


SLAB_ERR_LIST = 32           # error bit of a list head: the rank's unfinished sites did not fit its list (kErrSlabList)


def _slab_timeout():
    import os
    return float(os.environ.get("OFL_SLAB_TIMEOUT", "120"))


def scatter_slab(flow, sign, pmask, vals, C, vmask, h, w, row0, rows, out_rows, valid_rows, rank=0, world=1, valid_rule=0,
                 point_precision=0, stream=None, entries=1 << 17, gather=comm_allgather, timeout=None):
    """One row band of a ref-'s' warp whose mesh does not certify, with the star passes sharded over `world` ranks
    (SURVEY 8e, config 5): step 1 into a list of up to `entries` records, the exchange, step 2.  The exchange is two
    all-gathers -- the 16-byte list heads first, then (one read-back of the counts later) only as many 64-byte records per
    rank as the fullest list holds: config 5 at 8K leaves 75 000 sites unfinished in all, 1 MB per rank instead of the
    8 MiB the buffers are sized for.  `gather(send_ptr, recv_buffer, nbytes, stream)` defaults to RCCL over the live
    communicator (sharding.host_allgather(dist) goes through the host instead: rehearsals with ranks that share a GPU).
    Bands concatenate to scatter_linear's result bit for bit.

    Every rank of `world` MUST call this, whatever its band: a rank with an EMPTY band (rows == 0: more ranks than 8-row
    tiles) skips both steps but takes part in both gathers with an empty list; a rank whose step 1 fails joins them with an
    error head and raises afterwards, so that its peers fail with it instead of waiting for it.  When some rank's list
    overflowed (`entries` too small: large holes, hull sites of an 8K field) every rank sees the same counts in the gathered
    heads and all of them repeat the exchange ONCE with lists sized for the fullest.  Each gather -- and the read-back that
    waits for it -- is bounded by `timeout` seconds (default: OFL_SLAB_TIMEOUT, 120): a rank left alone in the collective
    ends its process with exit code 3 (sharding.bounded_call) instead of hanging for ever."""
    if world <= 1 and (row0 != 0 or rows != h):
        raise ValueError("scatter_slab: a band of a field needs the other ranks' lists")
    from .sharding import slab_payload_entries, bounded_call
    world = max(int(world), 1)
    timeout = _slab_timeout() if timeout is None else timeout
    ws = _workspace(h, w, 0, stream)               # held across both steps: whatever the exchange does to the cache, step 2 finds step 1's state
    for attempt in range(2):
        nb = slab_list_bytes(entries)
        mine = DeviceBuffer(nb)
        failed = None
        if rows > 0:
            try:
                scatter_slab_stars(flow, sign, pmask, h, w, row0, rows, mine.ptr, nb, point_precision, stream, ws)
            except nat.NativeError as e:            # the peers are on their way into the gathers: join them, then raise
                failed = e
        if rows <= 0 or failed is not None:
            head = np.array([0, SLAB_ERR_LIST if failed is not None else 0, 0, 0], np.uint32)
            nat.check(_lib().ofl_upload(mine.ptr, head.ctypes.data, SLAB_LIST_HEAD, stream))
            nat.check(_lib().ofl_stream_sync(stream))
        if world == 1:
            if failed is not None:
                raise failed
            return scatter_slab_finish(flow, sign, vals, C, vmask, h, w, row0, rows, mine, nb, 1, out_rows, valid_rows,
                                       valid_rule, point_precision, stream, ws)
        heads = DeviceBuffer(SLAB_LIST_HEAD * world)

        def exchange_heads():
            gather(mine.ptr, heads, SLAB_LIST_HEAD, stream)
            return heads.to_host((world, SLAB_LIST_HEAD // 4), np.uint32, stream)
        hw = bounded_call(exchange_heads, timeout, "the all-gather of the slab list heads")
        counts, errs = hw[:, 0], hw[:, 1]
        if attempt == 0 and int(counts.max()) > entries and not errs.any():
            # some rank's list overflowed; every rank reads the same heads and takes this branch together
            entries = int(counts.max()) + 1024
            continue
        m = slab_payload_entries(counts, entries)
        nb2 = slab_list_bytes(m)
        lists = DeviceBuffer(nb2 * world)

        def exchange_lists():
            gather(mine.ptr, lists, nb2, stream)
            nat.check(_lib().ofl_stream_sync(stream))
        bounded_call(exchange_lists, timeout, "the all-gather of the slab lists")
        if failed is not None:
            raise failed
        if rows <= 0:
            return (0, 0, 0)
        # (error bits in a peer's head -- a refused point set, a step 1 that failed there -- reach step 2 with the lists: it
        # blanks this band and raises here as well, include/ofl.h)
        return scatter_slab_finish(flow, sign, vals, C, vmask, h, w, row0, rows, lists, nb2, world, out_rows, valid_rows,
                                   valid_rule, point_precision, stream, ws)


def scatter_host(flow, target, pmask, vmask=None):
    """apply_flow(flow, target, 's', mask) for host arrays (utils.py:237-258; `flow` may already be a DeviceBuffer
    holding the float32 vectors): target (H, W, C) of any numeric
    dtype is interpolated in float32/float64 on the device, then rounded / cast back like the reference.
    Returns (warped, valid or None); valid = float32(interpolated vmask) == 1 (flow_class.py:668)."""
    h, w, C = target.shape
    n = h * w * C
    fbuf = flow if isinstance(flow, DeviceBuffer) else DeviceBuffer.from_host(np.ascontiguousarray(flow, np.float32))
    as_mask = lambda m: np.ascontiguousarray(m).view(np.uint8) if m.dtype == np.bool_ else np.ascontiguousarray(m).astype(np.uint8)
    pm = DeviceBuffer.from_host(as_mask(pmask)) if pmask is not None else None
    vm = DeviceBuffer.from_host(as_mask(vmask)) if vmask is not None else None
    valid = DeviceBuffer(h * w) if vmask is not None else None
    if target.dtype == np.float64:                                           # griddata's own precision end to end
        vals = DeviceBuffer.from_host(np.ascontiguousarray(target))
        out = DeviceBuffer(n * 8)
        scatter_linear_f64(fbuf, +1, pm, vals, C, vm, h, w, out, valid, 0)
        v = valid.to_host((h, w), np.uint8).view(np.bool_) if valid is not None else None
        return out.to_host((h, w, C), np.float64), v
    native = target.dtype in _DT_CODE and target.dtype != np.float32         # the casts of utils.py:253 / :258 run on the device
    integer = np.issubdtype(target.dtype, np.integer)
    if native:
        raw = DeviceBuffer.from_host(np.ascontiguousarray(target))
        vals = DeviceBuffer(n * 4)
        nat.check(_lib().ofl_convert_dev(raw.ptr, _DT_CODE[target.dtype], vals.ptr, nat.F32, n, None))
    else:
        vals = DeviceBuffer.from_host(np.ascontiguousarray(target, np.float32))
    out = DeviceBuffer(n * 4)
    # integer targets: values AND the concatenated mask channel are np.round-ed before the cast (utils.py:256-257)
    scatter_linear(fbuf, +1, pm, vals, C, vm, h, w, None, out, valid, (nat.SCATTER_ROUND | 2) if integer else 0)
    if native:
        back = DeviceBuffer(n * target.dtype.itemsize)
        nat.check(_lib().ofl_convert_dev(out.ptr, nat.F32, back.ptr, _DT_CODE[target.dtype], n, None))
        res = back.to_host((h, w, C), target.dtype)
    else:
        res = out.to_host((h, w, C), np.float32).astype(target.dtype)       # already rounded for integer targets
    v = valid.to_host((h, w), np.uint8).view(np.bool_) if valid is not None else None
    return res, v


def sample_points(flow_buf, h, w, pts_rc):
    """Bilinear flow samples (v, u) at float64 points (row, col): utils.py:161-196 / :605."""
    pts = np.ascontiguousarray(pts_rc, np.float64)
    n = pts.shape[0]
    dp = DeviceBuffer.from_host(pts)
    out = DeviceBuffer(max(n, 1) * 16)
    nat.check(_lib().ofl_sample_points_dev(flow_buf.ptr, h, w, dp.ptr, n, out.ptr, None))
    return out.to_host((n, 2), np.float64)


def scatter_query(pos_flow_buf, sign, vals_buf, C, h, w, query_xy, pmask=None):
    """griddata(points, values, query) for sparse float64 query points (utils.py:603, 614).
    Returns (values float64 [n, C], found bool [n])."""
    q = np.ascontiguousarray(query_xy, np.float64)
    n = q.shape[0]
    dq = DeviceBuffer.from_host(q)
    out = DeviceBuffer(max(n, 1) * C * 8)
    found = DeviceBuffer(max(n, 1))
    ws = _workspace(h, w, C)
    nat.check(_lib().ofl_scatter_query_dev(pos_flow_buf.ptr, sign, 0, pmask.ptr if pmask is not None else None,
                                           vals_buf.ptr, C, h, w, dq.ptr, n, out.ptr, found.ptr, ws.ptr, ws.nbytes, None))
    return out.to_host((n, C), np.float64), found.to_host((n,), np.uint8).view(np.bool_)
