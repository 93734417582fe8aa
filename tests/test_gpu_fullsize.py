"""BASELINE.json's configurations at FULL size, asserted (round 2 only benchmarked them): config 2 (1080 x 1920 apply 't' of
an RGB float32 image with valid area), config 4 (a GPU's share of 32 pairs of 1080 x 1920 in one launch, and all 256 pairs
in one launch), config 5 (tests/golden/sintel.flo tiled to 4320 x 7680: 't' against the oracle and as row bands; 's' --
the Delaunay path on 33 M points -- through the size-independent properties the domain offers: bands == full field,
valid_target == the valid area of apply, determinism).  Config 3 at full size lives in test_gpu_scatter.py
(::test_config3_mode1_full_size, ::test_scatter_4k_invert) and test_gpu_scatter_exact.py (::test_certified_4k_no_walk_failures),
the headline config (mode 3 at 2160 x 3840) in test_gpu_gather.py::test_full_size_properties.
"""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_config2_apply_1080p_rgb_f32(gpu, oracle):
    """f1 = rotation by -30 degrees about the centre of a 1080 x 1920 frame ('t'), image =
    rng(1).random((1080, 1920, 3), float32); f1.apply(img, return_valid_area=True) -- bit-identical to the oracle, with and
    without a target mask and a flow mask, through the host API and on HBM-resident data."""
    of = gpu
    from oflibnumpy_amd import device as dev
    h, w = 1080, 1920
    img = np.random.default_rng(1).random((h, w, 3), dtype=np.float32)
    f1 = of.Flow.from_transforms([['rotation', 960, 540, -30]], [h, w], 't')
    o1 = oracle.OFlow(f1.vecs, 't', f1.mask)
    got, valid = f1.apply(img, return_valid_area=True)
    want, wvalid = o1.apply(img, return_valid_area=True)
    np.testing.assert_array_equal(valid, wvalid)
    np.testing.assert_array_equal(got, want)
    assert 0.5 < valid.mean() < 0.95                       # the rotated frame leaves its corners
    rng = np.random.default_rng(7)
    fm, tm = rng.random((h, w)) > 0.05, rng.random((h, w)) > 0.1
    fmasked = of.Flow(f1.vecs, 't', fm)
    got, valid = fmasked.apply(img, tm, return_valid_area=True)
    want, wvalid = oracle.OFlow(f1.vecs, 't', fm).apply(img, tm, return_valid_area=True)
    np.testing.assert_array_equal(valid, wvalid)
    np.testing.assert_array_equal(got, want)
    # device-resident form (what the benchmark times)
    d, dimg = fmasked.to_device(), dev.DeviceImage.from_host(img)
    dw, dv = d.apply(dimg, target_mask=dev.DeviceBuffer.from_host(tm.astype(np.uint8)))
    np.testing.assert_array_equal(dw.to_host(), want)
    np.testing.assert_array_equal(dv.to_host((h, w), np.uint8).astype(bool), wvalid)
    # and the combine of the same config: f1.combine_with(f2 = scaling 0.8 about (400, 300), mode 3)
    f2 = of.Flow.from_transforms([['scaling', 400, 300, 0.8]], [h, w], 't')
    r = f1.combine_with(f2, 3)
    o, mo = oracle.compose3_raw(f1.vecs, f1.mask, f2.vecs, f2.mask, -1)
    np.testing.assert_array_equal(r.mask, mo)
    np.testing.assert_array_equal(r.vecs, o)


def test_config4_32_and_256_pairs_in_one_launch(gpu, oracle):
    """256 independent pairs of 1080 x 1920 'tau' flows, pair i = rotation by -30 + 60 i / 255 degrees
    about the centre (+) translation (40 cos i, 40 sin i), mode 3.  One launch of ofl_compose3_dev over all 256 pairs
    (14 GB of HBM-resident stacks), one over a GPU's share of 32 (sharding.shard(256, 3, 8)): sampled pairs equal the
    per-pair oracle bit for bit, the share equals the corresponding slice of the full batch, the fused zero-flow
    predicates are set for every pair."""
    of = gpu
    from oflibnumpy_amd import device as dev, sharding
    nat, lib = of.native, of.native.load()
    H, W, B = 1080, 1920, 256
    n = H * W
    va, ma, vb, mb = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n)
    keep = {}
    sample = (0, 37, 100, 121, 255)                 # 100 .. 127 is rank 3's share; 121 inside it
    ones = np.ones((H, W), np.uint8)
    yy, xx = np.mgrid[:H, :W].astype(np.float32)
    xx -= np.float32(W / 2)
    yy -= np.float32(H / 2)
    for i in range(B):
        if i in sample:          # the reference's own constructor (0.5 s per pair) for the pairs that are compared ...
            f1 = of.from_transforms([['rotation', W / 2, H / 2, -30 + 60 * i / 255]], [H, W], 't')
            f2 = of.from_transforms([['translation', 40 * math.cos(i), 40 * math.sin(i)]], [H, W], 't')
        else:                    # ... and the same fields up to float32 rounding, built directly, for the other 251
            a = math.radians(-30 + 60 * i / 255)
            c1, s1 = np.float32(math.cos(a) - 1), np.float32(math.sin(a))
            f1 = np.stack([-(c1 * xx - s1 * yy), -(s1 * xx + c1 * yy)], -1)
            f2 = np.empty((H, W, 2), np.float32)
            f2[...] = (-40 * math.cos(i), -40 * math.sin(i))
        m1 = (np.random.default_rng(i).random((H, W)) > 0.03).astype(np.uint8) if i in sample else ones
        for dst, a in ((va.ptr + i * n * 8, f1), (vb.ptr + i * n * 8, f2), (ma.ptr + i * n, m1), (mb.ptr + i * n, ones)):
            nat.check(lib.ofl_upload(dst, a.ctypes.data, a.nbytes, None))
        nat.check(lib.ofl_stream_sync(None))
        if i in sample:
            keep[i] = (f1, m1.astype(bool), f2)
    out, mout, stats = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer.zeros(32 * B)
    nat.check(lib.ofl_compose3_dev(va.ptr, ma.ptr, vb.ptr, mb.ptr, -1, H, W, B, out.ptr, mout.ptr, stats.ptr, 0, None))
    words = stats.to_host((B, 8), np.uint32)
    assert words[:, [0, 1, 4, 5, 6, 7]].all()
    one_v, one_m = np.empty((H, W, 2), np.float32), np.empty((H, W), np.uint8)
    for i in sample:
        nat.check(lib.ofl_download(one_v.ctypes.data, out.ptr + i * n * 8, n * 8, None))
        nat.check(lib.ofl_download(one_m.ctypes.data, mout.ptr + i * n, n, None))
        f1, m1, f2 = keep[i]
        o, m = oracle.compose3_raw(f1, m1, f2, np.ones((H, W), bool), -1)
        np.testing.assert_array_equal(one_v, o, err_msg="pair {}".format(i))
        np.testing.assert_array_equal(one_m.astype(bool), m, err_msg="pair {}".format(i))
        # and the analytic composite within the reference's own tolerance (tests/test_flow_class.py:1050-1057)
        f3 = of.from_transforms([['rotation', W / 2, H / 2, -30 + 60 * i / 255], ['translation', 40 * math.cos(i), 40 * math.sin(i)]], [H, W], 't')
        np.testing.assert_allclose(one_v[m], f3[m], atol=5e-2)
    # rank 3 of 8: its 32 pairs as ONE launch on the slices it would own
    r = sharding.shard(B, 3, 8)
    assert len(r) == 32
    sub, msub = dev.DeviceBuffer(32 * n * 8), dev.DeviceBuffer(32 * n)
    nat.check(lib.ofl_compose3_dev(va.ptr + r.start * n * 8, ma.ptr + r.start * n, vb.ptr + r.start * n * 8, mb.ptr + r.start * n,
                                   -1, H, W, 32, sub.ptr, msub.ptr, None, 0, None))
    full = np.empty((32, H, W, 2), np.float32)
    nat.check(lib.ofl_download(full.ctypes.data, out.ptr + r.start * n * 8, 32 * n * 8, None))
    np.testing.assert_array_equal(sub.to_host((32, H, W, 2), np.float32), full)
    fullm = np.empty((32, H, W), np.uint8)
    nat.check(lib.ofl_download(fullm.ctypes.data, mout.ptr + r.start * n, 32 * n, None))
    np.testing.assert_array_equal(msub.to_host((32, H, W), np.uint8), fullm)


@pytest.fixture(scope="module")
def tiled_sintel(gpu):
    flo = gpu.load_sintel(os.path.join(GOLDEN, "sintel.flo"))
    big = np.ascontiguousarray(np.tile(flo, (432, 384, 1)))
    assert big.shape == (4320, 7680, 2)
    img = np.random.default_rng(2).random((4320, 7680, 3), dtype=np.float32)
    return big, img


def test_config5_8k_t_vs_oracle_and_bands(gpu, oracle, tiled_sintel):
    """SURVEY 8(d) config 5 wrapped as 't' at its full 4320 x 7680: warp of the RGB float32 image + valid area equals the
    oracle bit for bit over the whole frame, and the row bands of 8 and of 3 ranks (ofl_gather_rows_dev: replicated image,
    a rank's own rows of the flow) concatenate to the same bits."""
    of = gpu
    from oflibnumpy_amd import device as dev, sharding
    big, img = tiled_sintel
    h, w = big.shape[:2]
    dimg = dev.DeviceImage.from_host(img)
    dflow = dev.DeviceFlow.from_host(big, 't')
    dw, dv = dflow.apply(dimg)
    got, valid = dw.to_host(), dv.to_host((h, w), np.uint8).astype(bool)
    want = oracle.gather_bilinear(img, big, -1)
    np.testing.assert_array_equal(got, want)
    del want
    wvalid = oracle.gather_bilinear(np.ones((h, w), np.float32), big, -1) == 1          # flow_class.py:644, 668 on the mask channel
    np.testing.assert_array_equal(valid, wvalid)
    assert 0.3 < valid.mean() < 1.0
    for world in (8, 3):
        for r in range(world):
            r0, r1 = sharding.row_band(h, r, world)
            fl = dev.DeviceBuffer.from_host(np.ascontiguousarray(big[r0:r1]))
            d, vv = dev.gather_rows(dimg, r0, r1 - r0, fl, -1, want_valid=True)
            np.testing.assert_array_equal(d.to_host(), got[r0:r1], err_msg="band {} of {}".format(r, world))
            np.testing.assert_array_equal(vv.to_host((r1 - r0, w), np.uint8).astype(bool), valid[r0:r1])


def test_config5_8k_s_bands_valid_target_determinism(gpu, tiled_sintel):
    """SURVEY 8(d) config 5 as loaded ('s') at its full 4320 x 7680 -- 33 M points, folded, 42 % duplicated: SciPy would
    need the better part of an hour, so the full size is pinned by properties (the 40 x 80 and 430 x 760 versions are
    compared with outputs of the real reference in test_gpu_scatter_exact.py): the 8 row bands of ofl_scatter_rows_dev
    equal the rows of the full result bit for bit, valid_target() equals the valid area apply() returns, a second call
    returns the same bits, the warp is finite and covers most of the frame."""
    of = gpu
    from oflibnumpy_amd import device as dev, sharding
    big, img = tiled_sintel
    h, w = big.shape[:2]
    dimg = dev.DeviceImage.from_host(img)
    dflow = dev.DeviceFlow.from_host(big, 's')
    dw, dv = dflow.apply(dimg)
    full, fvalid = dw.to_host(), dv.to_host((h, w), np.uint8)
    assert np.isfinite(full).all() and 0.3 < fvalid.mean() <= 1.0
    assert (full[fvalid == 0] == 0).all()                                         # outside the hull: NaN -> 0 (utils.py:254)
    inside = full[fvalid == 1]
    assert inside.min() >= -1e-6 and inside.max() <= 1.0 + 1e-6                   # convex combinations of image values in [0, 1)
    dw2, dv2 = dflow.apply(dimg)
    np.testing.assert_array_equal(dw2.to_host(), full)
    np.testing.assert_array_equal(dv2.to_host((h, w), np.uint8), fvalid)
    del dw2, dv2, inside
    vt = dflow.valid_target().to_host((h, w), np.uint8)
    np.testing.assert_array_equal(vt, fvalid)
    for r in range(8):
        r0, r1 = sharding.row_band(h, r, 8)
        ob, vb = dev.DeviceBuffer((r1 - r0) * w * 12), dev.DeviceBuffer((r1 - r0) * w)
        dev.scatter_rows(dflow.vecs, +1, None, dimg.buf, 3, None, h, w, r0, r1 - r0, ob, vb)
        np.testing.assert_array_equal(ob.to_host((r1 - r0, w, 3), np.float32), full[r0:r1], err_msg="band {}".format(r))
        np.testing.assert_array_equal(vb.to_host((r1 - r0, w), np.uint8), fvalid[r0:r1])
