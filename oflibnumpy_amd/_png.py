"""Minimal PNG reader for the dataset loaders (KITTI uint16 RGB flow, Sintel invalid-pixel masks).

The reference decodes these files with cv2.imread; the product has no OpenCV and Pillow collapses
16-bit RGB to 8 bits, so the few PNG features those datasets use are decoded here with zlib + NumPy:
non-interlaced images, colour types 0 / 2 / 4 / 6, bit depths 8 and 16 (1 / 2 / 4 for greyscale).
Returns an array (H, W) or (H, W, C) in file channel order (RGB), uint8 or uint16.
"""
import struct
import zlib

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}


def _unfilter(raw, height, stride, bpp):
    # a real KITTI flow PNG is 1242 x 375 x 6 bytes, mostly Sub / Average / Paeth rows: the byte loop below takes
    # seconds on it, the library's host helper milliseconds
    # (pure host code: no GPU involved; the Python loop is the same algorithm and serves when the library cannot be
    # loaded at all -- not built, or libamdhip64 missing on a CPU-only box)
    try:
        from . import _native as nat
        lib = nat.load()
    except (ImportError, OSError):
        return _unfilter_py(raw, height, stride, bpp)
    out = np.empty((height, stride), np.uint8)
    buf = np.frombuffer(raw, np.uint8)
    try:
        nat.check(lib.ofl_png_unfilter(buf.ctypes.data, buf.size, height, stride, bpp, out.ctypes.data))
    except nat.NativeError as e:
        # corrupt data (unknown filter byte, truncated IDAT): the same exception type as the Python loop, so that the loaders
        # report "could not be loaded" like the reference does when cv2.imread returns None (utils.py:437-439)
        raise ValueError("PNG: {}".format(e)) from None
    return out


def _unfilter_py(raw, height, stride, bpp):
    out = np.zeros((height, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    pos = 0
    for y in range(height):
        ftype = raw[pos]
        line = np.frombuffer(raw, np.uint8, stride, pos + 1).astype(np.int32)
        pos += stride + 1
        if ftype == 0:
            cur = line
        elif ftype == 2:
            cur = (line + prev) & 255
        elif ftype in (1, 3, 4):
            cur = line.copy()
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                if ftype == 1:
                    pred = a
                elif ftype == 3:
                    pred = (a + b) >> 1
                else:
                    c = prev[i - bpp] if i >= bpp else 0
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (cur[i] + pred) & 255
        else:
            raise ValueError("PNG: unknown filter type {}".format(ftype))
        out[y] = cur
        prev = cur
    return out


def read_png(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != _SIG:
        raise ValueError("not a PNG file: {}".format(path))
    pos, idat, header = 8, [], None
    while pos + 8 <= len(data):
        length, ctype = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + length]
        pos += 12 + length
        if ctype == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif ctype == b"IDAT":
            idat.append(body)
        elif ctype == b"IEND":
            break
    if header is None:
        raise ValueError("PNG without IHDR: {}".format(path))
    width, height, depth, ctype, _, _, interlace = header
    if interlace or ctype not in _CHANNELS or depth not in (1, 2, 4, 8, 16) or (depth < 8 and ctype != 0):
        raise ValueError("unsupported PNG layout in {} (colour type {}, depth {}, interlace {})".format(path, ctype, depth, interlace))
    ch = _CHANNELS[ctype]
    stride = (width * ch * depth + 7) // 8
    rows = _unfilter(zlib.decompress(b"".join(idat)), height, stride, max(1, ch * depth // 8))
    if depth == 16:
        img = rows.reshape(height, width * ch, 2).astype(np.uint16)
        img = (img[..., 0] << 8) | img[..., 1]
    elif depth == 8:
        img = rows
    else:
        bits = np.unpackbits(rows, axis=1)[:, :width * depth].reshape(height, width, depth)
        img = np.zeros((height, width), np.uint8)
        for k in range(depth):
            img = (img << 1) | bits[..., k]
        return img
    return img.reshape(height, width, ch) if ch > 1 else img.reshape(height, width)
