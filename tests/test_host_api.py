"""CPU suite, part 2: host logic of the product (constructors, validation, operators, error types --
the reference's tests/test_flow_class.py / test_utils.py restated for the hot-path surface), and the
C-ABI library: it loads, exports every symbol include/ofl.h declares, and refuses to compute without
a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import oflibnumpy_amd as of
from oflibnumpy_amd import utils, _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "ofl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ofl_[a-z0-9_]+)\s*\(", text)))


def test_abi_exports_every_declared_symbol():
    syms = header_symbols()
    assert len(syms) >= 30
    lib = ctypes.CDLL(nat.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libofl_hip.so does not export " + s
    assert set(syms) == set(nat.SIGNATURES), "ctypes table and include/ofl.h disagree"
    assert nat.load().ofl_abi_version() == nat.ABI_VERSION == 4


def test_shipped_library_is_lean():
    """The product library reads no environment knobs and carries none of the A/B kernel variants (they live in the
    experiments build, libofl_hip_exp.so, which tools/ and two subprocess tests load through OFL_LIB)."""
    import subprocess
    from oflibnumpy_amd import build_native
    assert os.path.basename(nat.LIB_PATH) == "libofl_hip.so" or os.environ.get("OFL_LIB")
    und = subprocess.run(["nm", "-D", "--undefined-only", build_native.OUT], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    blob = open(build_native.OUT, "rb").read()
    for name in (b"compose3_lds_kernel", b"compose3_oneshot_kernel", b"OFL_C3_VARIANT", b"OFL_C3_ABLATE", b"OFL_DL_NEAR2_MIN"):
        assert name not in blob, name
    assert b"compose3_xpose_kernel" in blob


def test_no_cpu_fallback_without_device():
    if nat.device_count() > 0:
        pytest.skip("a GPU is present")
    f = of.Flow.from_transforms([['rotation', 5, 5, 20]], (16, 20), 't')
    with pytest.raises(nat.NoDeviceError):
        f.combine_with(f, 3)
    with pytest.raises(nat.NoDeviceError):
        of.apply_flow(f.vecs, np.ones((16, 20), np.float32), 't')
    with pytest.raises(nat.NoDeviceError):
        f.valid_target()
    rc = nat.load().ofl_compose3_dev(None, None, None, None, -1, 4, 4, 1, None, None, None, 0, None)
    assert rc == nat.E_NODEVICE


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "oflibnumpy_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("np_oracle", "oracle") or fn == "", fn
    assert "scipy" not in open(os.path.join(pkg, "device.py")).read()


def test_flow_init_and_setters():
    v = np.zeros((10, 12, 2))
    f = of.Flow(v)
    assert f.vecs.dtype == np.float32 and f.ref == 't' and f.mask.dtype == bool and f.mask.all()
    assert f.shape == (10, 12)
    f = of.Flow(v, 's', np.ones((10, 12), 'i'))
    assert f.ref == 's' and f.mask.dtype == bool
    for bad in ('no', v[..., 0], np.zeros((10, 12, 3))):
        with pytest.raises((TypeError, ValueError)):
            of.Flow(bad)
    with pytest.raises(TypeError):
        of.Flow('test')
    with pytest.raises(ValueError):
        of.Flow(np.full((10, 12, 2), np.nan))
    with pytest.raises(ValueError):
        of.Flow(np.full((10, 12, 2), np.inf))
    with pytest.raises(TypeError):
        of.Flow(v, 0)
    with pytest.raises(ValueError):
        of.Flow(v, 'x')
    with pytest.raises(TypeError):
        of.Flow(v, mask='m')
    with pytest.raises(ValueError):
        of.Flow(v, mask=np.ones((10, 12, 1)))
    with pytest.raises(ValueError):
        of.Flow(v, mask=np.ones((11, 12)))
    with pytest.raises(ValueError):
        of.Flow(v, mask=np.full((10, 12), 2))


def test_constructors_match_analytic(oracle):
    shape = (30, 40)
    for ref in ('s', 't'):
        for tr in ([['translation', 3, -2]], [['rotation', 10, 20, 30]], [['scaling', 5, 5, 0.8]],
                   [['rotation', 10, 20, -30], ['scaling', 15, 25, 1.2], ['translation', 1, 2]]):
            a = of.Flow.from_transforms(tr, shape, ref)
            b = oracle.from_transforms(tr, shape, ref)
            np.testing.assert_array_equal(a.vecs, b.vecs)
            m = utils.matrix_from_transforms(tr)
            np.testing.assert_array_equal(of.Flow.from_matrix(m, shape, ref).vecs, a.vecs)
    z = of.Flow.zero((5, 6), 's')
    assert z.ref == 's' and not z.vecs.any()
    # rotation about the origin by 90 deg ccw moves (x, y) -> (y, -x) with y pointing down
    f = of.Flow.from_transforms([['rotation', 0, 0, 90]], (4, 4), 's')
    np.testing.assert_allclose(f.vecs[1, 2], [1 - 2, -2 - 1], atol=1e-5)
    with pytest.raises(TypeError):
        of.from_transforms('t', shape, 't')
    with pytest.raises(TypeError):
        of.from_transforms(['rotation', 1, 2, 3], shape, 't')
    with pytest.raises(ValueError):
        of.from_transforms([['rotation', 1, 2]], shape, 't')
    with pytest.raises(ValueError):
        of.from_transforms([['shear', 1, 2]], shape, 't')
    with pytest.raises(ValueError):
        of.from_transforms([['translation', 1, 'a']], shape, 't')
    with pytest.raises(TypeError):
        of.from_matrix('m', shape, 't')
    with pytest.raises(ValueError):
        of.from_matrix(np.eye(4), shape, 't')
    with pytest.raises(ValueError):
        of.from_matrix(np.eye(3), (0, 3), 't')
    with pytest.raises(TypeError):
        of.from_matrix(np.eye(3), 3, 't')


def test_operators():
    rng = np.random.default_rng(1)
    v1, v2 = rng.random((8, 9, 2), dtype=np.float32), rng.random((8, 9, 2), dtype=np.float32)
    m1, m2 = rng.random((8, 9)) > 0.3, rng.random((8, 9)) > 0.3
    f1, f2 = of.Flow(v1, 's', m1), of.Flow(v2, 't', m2)
    s = f1 + f2
    np.testing.assert_array_equal(s.vecs, v1 + v2)
    np.testing.assert_array_equal(s.mask, m1 & m2)
    assert s.ref == 's'                                   # left operand's reference
    d = f2 - f1
    np.testing.assert_array_equal(d.vecs, v2 - v1)
    assert d.ref == 't'
    np.testing.assert_array_equal((f1 + v2).vecs, v1 + v2)
    np.testing.assert_array_equal((f1 - v2).mask, m1)
    np.testing.assert_array_equal((-f1).vecs, -v1)
    np.testing.assert_array_equal((f1 * 2).vecs, v1 * 2)
    np.testing.assert_array_equal((f1 * [2, 3]).vecs, (v1 * np.array([2, 3])).astype('f'))
    np.testing.assert_array_equal((f1 / 2).vecs, v1 / 2)
    np.testing.assert_array_equal((f1 ** 2).vecs, v1 ** 2)
    np.testing.assert_array_equal((f1 * m1.astype('f')).vecs, v1 * m1[..., None])
    np.testing.assert_array_equal(f1[2:5, 1:4].vecs, v1[2:5, 1:4])
    c = f1.copy()
    assert c is not f1 and np.array_equal(c.vecs, v1) and c.ref == 's'
    assert "reference s" in str(f1)
    for bad in ('a', np.zeros((8, 9)), np.zeros((3, 9, 2))):
        with pytest.raises((TypeError, ValueError)):
            f1 + bad
    with pytest.raises(ValueError):
        f1 + of.Flow.zero((3, 3))
    with pytest.raises(ValueError):
        f1 * [1, 2, 3]
    with pytest.raises(ValueError):      # float('x') raises ValueError in the reference too
        f1 * 'x'
    with pytest.raises(TypeError):
        f1 * {}
    with pytest.raises(ValueError):
        f1 * np.zeros((3, 3))


def test_pad():
    f = of.Flow.from_transforms([['rotation', 0, 0, 45]], (7, 7), 's')      # get_padding values: tests/test_gpu_gather.py
    p = f.pad([1, 2, 3, 4])
    assert p.shape == (10, 14) and not p.mask[0].any() and p.mask[1, 3]
    np.testing.assert_array_equal(f.pad([1, 2, 3, 4], 'edge').vecs[0, 3:10], f.vecs[0])
    with pytest.raises(ValueError):
        f.pad([1, 2, 3, 4], 'wrap')
    with pytest.raises(TypeError):
        f.pad(3)
    with pytest.raises(ValueError):
        f.pad([1, 2, 3])
    with pytest.raises(ValueError):
        f.pad([1., 2, 3, 4])
    with pytest.raises(ValueError):
        f.pad([-1, 2, 3, 4])


def test_is_zero_validation_and_threshold():
    shape = (10, 10)
    mask = np.ones(shape, bool)
    mask[0, 0] = False
    v = np.zeros(shape + (2,))
    v[0, 0] = 10
    f = of.Flow(v, mask=mask)                   # predicate values run on the device: tests/test_gpu_gather.py
    with pytest.raises(TypeError):
        f.is_zero(masked='test')
    with pytest.raises(TypeError):
        f.is_zero(thresholded='test')
    v = np.zeros(shape + (2,), np.float32)
    with pytest.raises(TypeError):
        of.is_zero_flow(v, 'x')
    t = of.threshold_vectors(np.array([[[1e-4, 2e-3], [-5e-4, -1.]]], np.float32))
    np.testing.assert_array_equal(t, np.array([[[0, 2e-3], [0, -1.]]], np.float32))
    t = of.threshold_vectors(np.array([[[3., 4.], [6e-4, 6e-4]]], np.float32), 1e-3, use_mag=True)
    np.testing.assert_array_equal(t, np.array([[[3., 4.], [0, 0]]], np.float32))


def test_apply_argument_validation():
    """Error types of reference tests/test_flow_class.py:469-499 and tests/test_utils.py:285-302 are raised
    on the host BEFORE anything touches the GPU."""
    for ref in ('t', 's'):
        shape = (10, 10)
        flow = of.Flow.from_transforms([['rotation', 0, 0, 30]], shape, ref)
        img = np.ones(shape + (3,), 'uint8')
        with pytest.raises(ValueError):
            flow.apply(img[0, 0])
        with pytest.raises(ValueError):
            flow.apply(img[..., np.newaxis])
        with pytest.raises(TypeError):
            flow.apply(flow, padding=100, cut=True)
        with pytest.raises(ValueError):
            flow.apply(flow, padding=[10, 20, 30, 40, 50], cut=True)
        with pytest.raises(ValueError):
            flow.apply(flow, padding=[10., 20, 30, 40], cut=True)
        with pytest.raises(ValueError):
            flow.apply(flow, padding=[-10, 10, 10, 10], cut=True)
        with pytest.raises(TypeError):
            flow.apply(flow, padding=[10, 20, 30, 40, 50], cut=2)
        with pytest.raises(TypeError):
            flow.apply(flow, return_valid_area='test')
        with pytest.raises(TypeError):
            flow.apply(flow, consider_mask='test')
        with pytest.raises(TypeError):
            flow.apply(img, target_mask='test')
        with pytest.raises(TypeError):
            flow.apply(img, target_mask=np.ones(shape, 'i'))
        with pytest.raises(ValueError):
            flow.apply(img, target_mask=np.ones((5, 5), 'bool'))
        with pytest.raises(ValueError):
            flow.apply(np.ones((11, 10, 3), 'uint8'))
    f = of.from_transforms([['rotation', 0, 0, 30]], (10, 10), 't')
    with pytest.raises(TypeError):
        of.apply_flow(f, 2, 't')
    with pytest.raises(ValueError):
        of.apply_flow(f, np.ones((10, 10, 3, 1)), 't')
    with pytest.raises(ValueError):
        of.apply_flow(f, np.ones((11, 10)), 't')
    with pytest.raises(TypeError):
        of.apply_flow(f, np.ones((10, 10)), 't', mask=0)
    with pytest.raises(ValueError):
        of.apply_flow(f, np.ones((10, 10)), 't', mask=np.ones((11, 10), bool))
    with pytest.raises(TypeError):
        of.apply_flow(f, np.ones((10, 10)), 't', mask=np.ones((10, 10), 'i'))
    with pytest.raises(TypeError):
        of.apply_flow(f, np.ones((10, 10)), 0)
    with pytest.raises(ValueError):
        of.apply_flow(f, np.ones((10, 10)), 'x')


def test_combine_and_switch_argument_validation():
    tr = [['rotation', 5, 5, -30], ['scaling', 3, 3, 0.8]]
    fs = of.Flow.from_transforms(tr[0:1], [20, 20], 's')
    ft = of.Flow.from_transforms(tr[1:2], [20, 20], 't')
    fs2 = of.Flow.from_transforms(tr[0:1], [20, 30], 's')
    with pytest.raises(TypeError):
        fs.combine_with(fs.vecs, 1)
    with pytest.raises(ValueError):
        fs.combine_with(fs2, 1)
    with pytest.raises(ValueError):
        fs.combine_with(ft, 1)
    with pytest.raises(ValueError):
        fs.combine_with(fs, mode=0)
    with pytest.raises(TypeError):
        fs.combine_with(fs, 1, thresholded='test')
    with pytest.raises(ValueError):
        fs.switch_ref('test')
    with pytest.raises(ValueError):
        fs.switch_ref(1)
    assert fs.switch_ref(mode='invalid').ref == 't'
    assert fs.invert('t').ref == 't' and np.array_equal(fs.invert('t').vecs, -fs.vecs)
    with pytest.raises(TypeError):
        fs.valid_target(consider_mask='test')
    with pytest.raises(TypeError):
        fs.valid_source(consider_mask='test')


def test_load_sintel():
    path = os.path.join(ROOT, "tests", "golden", "sintel.flo")
    v = of.load_sintel(path)
    assert v.shape == (10, 20, 2)
    r, c = np.mgrid[:10, :20]
    np.testing.assert_array_equal(v[..., 0], (r * c).astype('f'))
    assert not v[..., 1].any()
    f = of.Flow.from_sintel(path)
    assert f.ref == 's' and f.mask.all()
    with pytest.raises(TypeError):
        of.load_sintel(0)
    bad = os.path.join(ROOT, "tests", "golden", "make_golden.py")
    with pytest.raises(ValueError):
        of.load_sintel(bad)
    with pytest.raises(ValueError):
        of.load_sintel(os.path.join(ROOT, "tests", "golden", "sintel_wrong.flo"))


def test_dataset_loaders():
    """reference tests/test_utils.py:426-470 and tests/test_flow_class.py:115-148 on the reference's own fixtures."""
    g = lambda n: os.path.join(ROOT, "tests", "golden", n)
    want = np.arange(0, 10)[:, np.newaxis] * np.arange(0, 20)[np.newaxis, :]
    out = of.load_kitti(g("kitti.png"))
    assert isinstance(out, np.ndarray) and out.dtype == np.float64
    np.testing.assert_array_equal(out[..., 0], want)
    np.testing.assert_array_equal(out[..., 1], 0)
    np.testing.assert_array_equal(out[:, 0, 2], 1)
    np.testing.assert_array_equal(out[:, 10, 2], 0)
    with pytest.raises(ValueError):
        of.load_kitti("test")
    with pytest.raises(ValueError):
        of.load_kitti(g("kitti_wrong.png"))
    m = of.load_sintel_mask(g("sintel_invalid.png"))
    assert m.dtype == bool and m[:, 0].all() and not m[:, 10].any()
    with pytest.raises(TypeError):
        of.load_sintel_mask(0)
    with pytest.raises(ValueError):
        of.load_sintel_mask("test.png")
    f = of.Flow.from_kitti(g("kitti.png"), load_valid=True)
    np.testing.assert_array_equal(f.vecs[..., 0], want)
    assert f.ref == 's' and f.mask[:, 0].all() and not f.mask[:, 10].any()
    assert of.Flow.from_kitti(g("kitti.png"), load_valid=False).mask.all()
    with pytest.raises(TypeError):
        of.Flow.from_kitti(g("kitti.png"), load_valid='test')
    with pytest.raises(ValueError):
        of.Flow.from_kitti('test')
    with pytest.raises(ValueError):
        of.Flow.from_kitti(g("kitti_wrong.png"))
    f = of.Flow.from_sintel(g("sintel.flo"), g("sintel_invalid.png"))
    assert f.mask[:, 0].all() and not f.mask[:, 10].any()
    with pytest.raises(ValueError):
        of.Flow.from_sintel(g("sintel_wrong.flo"))
    with pytest.raises(ValueError):
        of.Flow.from_sintel(g("sintel.flo"), 'test.png')
    with pytest.raises(ValueError):
        of.Flow.from_sintel(g("sintel.flo"), g("sintel_invalid_wrong.png"))


def test_resize_argument_validation():
    """Error types of reference tests/test_utils.py:506-518 -- raised before anything touches the GPU."""
    flow = of.Flow.from_transforms([['rotation', 30, 50, 30]], [20, 10], 's')
    for fn in (lambda s: of.resize_flow(flow.vecs, s), flow.resize):
        with pytest.raises(TypeError):
            fn('test')
        with pytest.raises(ValueError):
            fn(['test', 0])
        with pytest.raises(ValueError):
            fn([1, 2, 3])
        with pytest.raises(ValueError):
            fn(0)
        with pytest.raises(ValueError):
            fn(-0.1)
    with pytest.raises(TypeError):
        of.resize_flow('test', 1)
    with pytest.raises(ValueError):
        of.resize_flow(np.zeros((5, 5, 3)), 1)


def test_png_reader_all_filter_types_native_equals_python(tmp_path):
    """load_kitti / load_sintel_mask (reference utils.py:426-490, cv2.imread there) decode PNGs with zlib + a row
    un-filter; the library's host helper (ofl_png_unfilter) and the pure-Python loop agree on every filter type, on a
    16-bit RGB image with the row layout of a KITTI flow file."""
    import struct
    import zlib
    from oflibnumpy_amd import _png
    h, w, bpp = 23, 57, 6
    rng = np.random.default_rng(0)
    img = rng.integers(0, 65535, (h, w, 3)).astype('>u2')
    rows = img.reshape(h, -1).view(np.uint8).reshape(h, -1)
    stride = rows.shape[1]
    raw, prev = bytearray(), np.zeros(stride, np.int32)
    for y in range(h):
        ft, cur = y % 5, rows[y].astype(np.int32)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 4:
            p = a + prev - c
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, c))
        else:
            pred = [0 * cur, a, prev, (a + prev) >> 1][ft]
        raw.append(ft)
        raw += bytes(((cur - pred) & 255).astype(np.uint8))
        prev = cur
    chunk = lambda t, b: struct.pack(">I", len(b)) + t + b + struct.pack(">I", zlib.crc32(t + b))
    path = str(tmp_path / "rgb16.png")
    open(path, 'wb').write(_png._SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b""))
    np.testing.assert_array_equal(_png.read_png(path), img.astype(np.uint16))
    np.testing.assert_array_equal(_png._unfilter(bytes(raw), h, stride, bpp), _png._unfilter_py(bytes(raw), h, stride, bpp))
    # corrupt files fail like they do in the reference (cv2.imread returns None -> ValueError "could not be loaded"),
    # through the native helper and through the Python loop alike: an unknown filter byte, a truncated IDAT
    bad = bytearray(raw)
    bad[0] = 7
    for data, what in ((bytes(bad), "filter byte 7"), (bytes(raw[:len(raw) // 2]), "truncated")):
        p2 = str(tmp_path / "bad.png")
        open(p2, 'wb').write(_png._SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)) +
                             chunk(b"IDAT", zlib.compress(data)) + chunk(b"IEND", b""))
        with pytest.raises(ValueError, match="could not be loaded"):
            of.load_kitti(p2)
        with pytest.raises((ValueError, IndexError)):
            _png._unfilter_py(data, h, stride, bpp)
