"""GPU parity tests (run with `-m gpu` on an MI355X): the HIP gather kernels K1 / K2 / K4 / K5 called
through the C ABI, against the CPU oracle on the same seeded inputs.  Bars: boolean masks and integer
images bit-exact; float32 / float64 values bit-exact as well (the kernels mirror the oracle operation
for operation), which is stricter than the 1e-4 relative tolerance BASELINE.json asks for."""
import numpy as np
import pytest

from test_oracle import (k7_flows, K7_VALID_TARGET_T, K7_VALID_TARGET_T_MASKED, K7_VALID_SOURCE_S,
                         K7_VALID_SOURCE_S_MASKED)

pytestmark = pytest.mark.gpu

RTOL = 1e-4   # BASELINE.json: "within 1e-4 relative float32 on the warped pixels"


def wobble_flow(shape, seed, amp=3.0):
    h, w = shape
    y, x = np.mgrid[:h, :w].astype('f')
    rng = np.random.default_rng(seed)
    a, b = rng.uniform(40, 130, 2)
    return np.stack([amp * np.sin(2 * np.pi * x / a) * np.cos(2 * np.pi * y / b),
                     amp * np.cos(2 * np.pi * x / b) * np.sin(2 * np.pi * y / a)], -1).astype('f')


def rand_mask(shape, seed, p=0.05):
    return np.random.default_rng(seed).random(shape) > p


CASES = [  # (shape, transforms f1, transforms f2)
    ((300, 400), [['rotation', 200, 150, -30]], [['translation', 40, 0]]),          # BASELINE config 1 (README)
    ((256, 512), [['rotation', 255.5, 127.5, -30]], [['scaling', 100, 100, 0.8]]),  # reference test_combine_with
    ((64, 68), [['rotation', 10, 10, 75]], [['scaling', 30, 30, 1.7]]),
    ((37, 53), [['rotation', 20, 10, -15]], [['translation', 2.5, -3.25]]),         # W % 4 != 0 -> generic kernel
    ((5, 1), [['translation', 0.5, 1]], [['translation', 0, 0.25]]),                # degenerate width
    ((1, 8), [['translation', 0.5, 0]], [['translation', 1.5, 0]]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("ref", ['t', 's'])
def test_compose3_matches_oracle_bit_exact(gpu, oracle, case, ref):
    of = gpu
    shape, t1, t2 = CASES[case]
    # (np.squeeze inside the reference's flow generator drops a width / height of 1, so the vectors are
    # generated on a padded shape and cut)
    big = [shape[0] + 1, shape[1] + 1]
    v1 = of.from_transforms(t1, big, ref)[:shape[0], :shape[1]] + wobble_flow(shape, case, 0.7)
    v2 = of.from_transforms(t2, big, ref)[:shape[0], :shape[1]]
    f1 = of.Flow(v1, ref, rand_mask(shape, case))
    f2 = of.Flow(v2, ref, rand_mask(shape, case + 100))
    got = f1.combine_with(f2, 3)
    want = oracle.OFlow(f1.vecs, ref, f1.mask).combine_with(oracle.OFlow(f2.vecs, ref, f2.mask), 3)
    assert got.ref == want.ref == ref
    np.testing.assert_array_equal(got.mask, want.mask)
    np.testing.assert_array_equal(got.vecs, want.vecs)
    np.testing.assert_allclose(got.vecs, want.vecs, rtol=RTOL, atol=0)
    # array facade (reference tests/test_flow_operations.py:26-72): same vectors as the method
    np.testing.assert_array_equal(of.combine_flows(f1.vecs, f2.vecs, 3, ref),
                                  of.Flow(f1.vecs, ref).combine_with(of.Flow(f2.vecs, ref), 3).vecs)


@pytest.mark.parametrize("ref", ['t', 's'])
def test_compose3_analytic(gpu, ref):
    """reference tests/test_flow_class.py:1050-1057 at its own size and tolerance (atol 5e-2 inside masks)."""
    of = gpu
    shape = [512, 512]
    tr = [['rotation', 255.5, 255.5, -30], ['scaling', 100, 100, 0.8]]
    f1, f2, f3 = (of.Flow.from_transforms(t, shape, ref) for t in (tr[:1], tr[1:], tr))
    r = f1.combine_with(f2, 3)
    assert isinstance(r, of.Flow) and r.ref == ref
    m = r.mask & f3.mask
    assert m.sum() > 50000
    np.testing.assert_allclose(r.vecs[m], f3.vecs[m], atol=5e-2)


def test_compose3_raw_abi_batch_and_stats(gpu, oracle):
    """ofl_compose3 (host-pointer entry of the C ABI) with a batch of fields, both quantisation modes,
    and the fused zero-flow statistics."""
    import ctypes
    of = gpu
    nat, lib = of.native, of.native.load()
    B, H, W = 3, 48, 64
    rng = np.random.default_rng(5)
    fa = (rng.standard_normal((B, H, W, 2)) * 4).astype('f')
    fb = (rng.standard_normal((B, H, W, 2)) * 6).astype('f')
    ma = (rng.random((B, H, W)) > 0.1).astype(np.uint8)
    mb = (rng.random((B, H, W)) > 0.1).astype(np.uint8)
    fa[1] = 0                      # sampled field exactly zero
    fb[2] = 5e-4                   # position field below the 1e-3 threshold but non-zero
    fb[0][~mb[0].astype(bool)] = 0
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for quant in (nat.QUANT_OPENCV, nat.QUANT_EXACT):
        for sign in (-1, 1):
            out = np.empty_like(fa)
            mout = np.empty_like(ma)
            stats = np.zeros(2 * B, np.uint32)
            nat.check(lib.ofl_compose3(p(fa), p(ma), p(fb), p(mb), sign, H, W, B, p(out), p(mout), p(stats), quant))
            for b in range(B):
                o, m = oracle.compose3_raw(fa[b], ma[b], fb[b], mb[b], sign, quant)
                np.testing.assert_array_equal(out[b], o)
                np.testing.assert_array_equal(mout[b].astype(bool), m)
            S = nat
            assert stats[2] == 0                                               # fa[1] all zero
            assert stats[0] == (S.STAT_NONZERO_MASKED | S.STAT_NONZERO_TH_MASKED | S.STAT_NONZERO | S.STAT_NONZERO_TH)
            assert stats[5] == (S.STAT_NONZERO_MASKED | S.STAT_NONZERO)          # fb[2]: non-zero but thresholded zero
            for b in range(B):                                                  # vs the oracle predicates
                for k, (f, m) in enumerate(((fa[b], ma[b]), (fb[b], mb[b]))):
                    s = int(stats[2 * b + k])
                    assert bool(s & S.STAT_NONZERO_MASKED) == (not oracle.is_zero_raw(f, m, False))
                    assert bool(s & S.STAT_NONZERO_TH_MASKED) == (not oracle.is_zero_raw(f, m, True))
                    assert bool(s & S.STAT_NONZERO) == (not oracle.is_zero_raw(f, None, False))
                    assert bool(s & S.STAT_NONZERO_TH) == (not oracle.is_zero_raw(f, None, True))


def test_compose3_early_exits(gpu):
    """flow_class.py:1339-1354: zero operands hand back the other operand object."""
    of = gpu
    shape = [40, 48]
    f = of.Flow.from_transforms([['rotation', 10, 10, 20]], shape, 't')
    z = of.Flow.zero(shape, 't')
    assert z.combine_with(f, 3) is f
    assert f.combine_with(z, 3) is f
    m = np.zeros(shape, bool)            # nothing valid => "zero" wherever the mask is True
    zm = of.Flow(f.vecs, 't', m)
    assert zm.combine_with(f, 3) is f
    tiny = of.Flow(np.full(shape + [2], 5e-4, 'f'), 't')
    assert tiny.combine_with(f, 3, thresholded=True) is f
    r = tiny.combine_with(f, 3)          # not thresholded: computed normally
    assert r is not f and r is not tiny


DTYPES = [np.uint8, np.int16, np.uint16, np.float32, np.float64]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [1, 2, 3, 4, 6])
def test_gather_raw_matches_oracle(gpu, oracle, dtype, C):
    """ofl_gather_bilinear (host entry) vs the oracle for every dtype / channel count / arithmetic / rule,
    with a flow placed inside a larger target (padding offsets)."""
    import ctypes
    of = gpu
    nat, lib = of.native, of.native.load()
    H, W, fH, fW, top, left = 50, 70, 40, 58, 6, 9
    rng = np.random.default_rng(C * 10 + np.dtype(dtype).itemsize)
    if np.issubdtype(dtype, np.integer):
        info = np.iinfo(dtype)
        src = rng.integers(info.min, info.max, (H, W, C), endpoint=True).astype(dtype)
    else:
        src = (rng.standard_normal((H, W, C)) * 100).astype(dtype)
    flow = (of.from_transforms([['rotation', 30, 20, 25], ['scaling', 10, 10, 1.3]], [fH, fW], 't')
            + wobble_flow((fH, fW), C, 1.5)).astype('f')
    flow[0, 0] = [1e9, -1e9]                   # far outside: saturates like cv2's int16 coordinates
    flow[1, 1] = [0.5, 0.5]                    # exact half-pixel: fixed-point / round-half-even ties
    smask = (rng.random((H, W)) > 0.2).astype(np.uint8)
    fmask = (rng.random((fH, fW)) > 0.2).astype(np.uint8)
    code = {np.uint8: nat.U8, np.int16: nat.I16, np.uint16: nat.U16, np.float32: nat.F32, np.float64: nat.F64}[dtype]
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    ariths = (nat.ARITH_NATIVE, nat.ARITH_FLOAT_RNE) if dtype == np.uint8 else (nat.ARITH_NATIVE,)
    for quant in (nat.QUANT_OPENCV, nat.QUANT_EXACT):
        for arith in ariths:
            for rule in (nat.RULE_EQ1, nat.RULE_GE_HALF, nat.RULE_GT_HALF):
                for sign in (-1, 1):
                    dst = np.empty_like(src)
                    valid = np.empty((H, W), np.uint8)
                    nat.check(lib.ofl_gather_bilinear(p(src), code, C, H, W, p(flow), fH, fW, top, left, sign,
                                                      p(smask), p(fmask), p(dst), p(valid), quant, arith, rule))
                    want, wv = oracle.gather_bilinear(src, flow, sign, smask=smask, want_valid=True, quant=quant,
                                                      arith=arith, rule=rule, pad=(top, left))
                    np.testing.assert_array_equal(dst, want)
                    inside = np.zeros((H, W), bool)
                    inside[top:top + fH, left:left + fW] = fmask.astype(bool)
                    np.testing.assert_array_equal(valid.astype(bool), wv & inside)


def test_apply_flow_t_known_answers(gpu):
    """reference tests/test_utils.py:277-283 (integer translation == ndimage.shift, exact) and the 7x7 masks
    of tests/test_flow_class.py:872-880, 899-912, 928-954, 973-975 through the product API."""
    from scipy import ndimage
    of = gpu
    img = (np.random.default_rng(0).random((512, 512, 3)) * 255).astype(np.uint8)
    f = of.from_transforms([['translation', 10, 20]], [512, 512], 't')
    np.testing.assert_array_equal(of.apply_flow(f, img, 't'), ndimage.shift(img, [20, 10, 0]))
    f_s, f_sm, f_t, f_tm = k7_flows(lambda t, s, r, m=None: of.Flow.from_transforms(t, list(s), r, m))
    np.testing.assert_array_equal(f_t.valid_target(), K7_VALID_TARGET_T)
    np.testing.assert_array_equal(f_tm.valid_target(), K7_VALID_TARGET_T_MASKED)
    np.testing.assert_array_equal(f_s.valid_source(), K7_VALID_SOURCE_S)
    np.testing.assert_array_equal(f_sm.valid_source(), K7_VALID_SOURCE_S_MASKED)
    np.testing.assert_array_equal(of.valid_target(f_t.vecs, 't'), K7_VALID_TARGET_T)
    np.testing.assert_array_equal(of.valid_source(f_s.vecs, 's'), K7_VALID_SOURCE_S)


@pytest.mark.parametrize("dtype", DTYPES)
def test_flow_apply_t_matches_oracle(gpu, oracle, dtype):
    """Flow.apply ('t') for every target kind of reference tests/test_flow_class.py:418-467: 3-D / 2-D arrays,
    with and without valid area and target mask, Flow targets, padding with and without cut."""
    of = gpu
    shape = (90, 110)
    rng = np.random.default_rng(3)
    img = (rng.random(shape + (3,)) * 200).astype(dtype)
    fmask = rand_mask(shape, 11, 0.1)
    tmask = rand_mask(shape, 12, 0.1)
    flow = of.Flow.from_transforms([['rotation', 30, 50, 30]], list(shape), 't', fmask)
    oflow = oracle.OFlow(flow.vecs, 't', fmask)
    np.testing.assert_array_equal(flow.apply(img), oflow.apply(img))
    np.testing.assert_array_equal(flow.apply(img), of.apply_flow(flow.vecs, img, 't'))
    np.testing.assert_array_equal(flow.apply(img[..., 0]), oflow.apply(img[..., 0]))
    for tm in (None, tmask):
        if dtype == np.uint16 and tm is None:
            # uint16 image + the default int8 mask concatenates to int32, which cv2.remap cannot
            # interpolate (cv2.error in the reference): refused on the host
            with pytest.raises(TypeError):
                flow.apply(img, tm, return_valid_area=True)
            continue
        w, v = flow.apply(img, tm, return_valid_area=True)
        ow, ov = oflow.apply(img, tm, return_valid_area=True)
        np.testing.assert_array_equal(w, ow)
        np.testing.assert_array_equal(v, ov)
        assert w.dtype == dtype and v.dtype == bool
        w, v = flow.apply(img[..., 1], tm, return_valid_area=True)
        ow, ov = oflow.apply(img[..., 1], tm, return_valid_area=True)
        np.testing.assert_array_equal(w, ow)
        np.testing.assert_array_equal(v, ov)


def test_flow_apply_t_flow_target_and_padding(gpu, oracle):
    of = gpu
    shape = (90, 110)
    m1, m2 = rand_mask(shape, 1, 0.1), rand_mask(shape, 2, 0.1)
    flow = of.Flow.from_transforms([['rotation', 30, 50, 30]], list(shape), 't', m1)
    tgt = of.Flow.from_transforms([['scaling', 20, 20, 1.2]], list(shape), 's', m2)
    r = flow.apply(tgt)
    o = oracle.OFlow(flow.vecs, 't', m1).apply(oracle.OFlow(tgt.vecs, 's', m2))
    assert r.ref == 's'
    np.testing.assert_array_equal(r.vecs, o.vecs)
    np.testing.assert_array_equal(r.mask, o.mask)
    # padding: a smaller flow on a larger target == the full flow cut out (reference test_apply :441-467)
    img = (np.random.default_rng(4).random((120, 160, 3)) * 255).astype(np.uint8)
    full = of.Flow.from_transforms([['rotation', 30, 50, 30]], [120, 160], 't')
    desired = of.apply_flow(full.vecs, img, 't')
    padding = [50, 40, 30, 80]
    cut_flow = of.Flow.from_transforms([['rotation', 0, 0, 30]], [120 - 90, 160 - 110], 't')
    got = cut_flow.apply(img, padding=padding, cut=False)
    np.testing.assert_array_equal(got[50:-40, 30:-80], desired[50:-40, 30:-80])
    np.testing.assert_array_equal(got[:50], img[:50])               # zero flow outside: identity
    got = cut_flow.apply(img, padding=padding, cut=True)
    np.testing.assert_array_equal(got, desired[50:-40, 30:-80])
    w, v = cut_flow.apply(img, return_valid_area=True, padding=padding, cut=False)
    assert v.shape == (120, 160) and not v[:50].any() and not v[:, :30].any()
    tf = of.Flow.from_transforms([['rotation', 30, 50, 30]], [120, 160], 't')
    wf = cut_flow.apply(tf, padding=padding, cut=True)
    np.testing.assert_array_equal(wf.vecs, of.apply_flow(full.vecs, tf.vecs, 't')[50:-40, 30:-80])


def test_flow_stats_and_axpy(gpu, oracle):
    import ctypes
    of = gpu
    nat, lib = of.native, of.native.load()
    rng = np.random.default_rng(8)
    for n in (1, 2, 7, 1000, 4097):
        v = np.zeros((n, 2), np.float32)
        m = (rng.random(n) > 0.3).astype(np.uint8)
        p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        for variant in range(5):
            if variant == 1:
                v[n // 2] = [0, 7e-4]
            elif variant == 2:
                v[n - 1] = [-3, 0]
                m[n - 1] = 0
            elif variant == 3:
                m[n - 1] = 1
            elif variant == 4:
                v[0, 0] = np.inf
            for mask in (None, m):
                s = np.zeros(1, np.uint32)
                nat.check(lib.ofl_flow_stats(p(v), p(mask), n, np.float32(1e-3), p(s)))
                s = int(s[0])
                fin = v[np.isfinite(v).all(-1)] if variant == 4 else v
                assert bool(s & nat.STAT_NONFINITE) == (variant == 4)
                if variant != 4:
                    mm = mask if mask is not None else np.ones(n, np.uint8)
                    assert bool(s & nat.STAT_NONZERO) == (not oracle.is_zero_raw(v[None], None, False))
                    assert bool(s & nat.STAT_NONZERO_TH) == (not oracle.is_zero_raw(v[None], None, True))
                    assert bool(s & nat.STAT_NONZERO_MASKED) == (not oracle.is_zero_raw(v[None], mm[None], False))
                    assert bool(s & nat.STAT_NONZERO_TH_MASKED) == (not oracle.is_zero_raw(v[None], mm[None], True))
    a = of.DeviceFlow.from_host(rng.standard_normal((33, 35, 2)).astype('f'), 't', rng.random((33, 35)) > 0.2)
    b = of.DeviceFlow.from_host(rng.standard_normal((33, 35, 2)).astype('f'), 's', rng.random((33, 35)) > 0.2)
    (av, am), (bv, bm) = a.to_host(), b.to_host()
    for got, want in (((a + b).to_host(), (av + bv, am & bm)), ((a - b).to_host(), (av - bv, am & bm)),
                      ((-a).to_host(), (-av, am))):
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
    assert (a + b).ref == 't'


def test_full_size_properties(gpu, oracle):
    """BASELINE.json sizes (2160 x 3840), through size-independent properties:
      * translation (+) translation composes to the exact sum with a full-rectangle mask,
      * row bands of the big result equal the oracle run on those bands' dependencies,
      * a batch launch equals per-field launches."""
    of = gpu
    H, W = 2160, 3840
    t1 = of.Flow.from_transforms([['translation', 16, -8]], [H, W], 't')
    t2 = of.Flow.from_transforms([['translation', -5, 24]], [H, W], 't')
    r = t1.combine_with(t2, 3)
    np.testing.assert_array_equal(r.vecs[r.mask], np.broadcast_to(np.float32([11, 16]), (int(r.mask.sum()), 2)))
    want_mask = np.zeros((H, W), bool)
    want_mask[24:H, 0:W - 5] = True         # samples (x + 5, y - 24) must land inside the source
    np.testing.assert_array_equal(r.mask, want_mask)
    # general flows: compare three 16-row bands with the oracle evaluated on the full inputs
    f1 = of.Flow.from_transforms([['rotation', 1920, 1080, -30]], [H, W], 't', rand_mask((H, W), 21))
    f2 = of.Flow.from_transforms([['scaling', 800, 600, 0.8]], [H, W], 't', rand_mask((H, W), 22))
    r = f1.combine_with(f2, 3)
    o, mo = oracle.compose3_raw(f1.vecs, f1.mask, f2.vecs, f2.mask, -1)
    np.testing.assert_array_equal(r.mask, mo)
    np.testing.assert_array_equal(r.vecs, o)
    f3 = of.Flow.from_transforms([['rotation', 1920, 1080, -30], ['scaling', 800, 600, 0.8]], [H, W], 't')
    m = r.mask & f3.mask
    np.testing.assert_allclose(r.vecs[m], f3.vecs[m], atol=5e-2)
    # the other order samples along a ROTATED grid: the workgroups take the transposed-gather path
    r = f2.combine_with(f1, 3)
    o, mo = oracle.compose3_raw(f2.vecs, f2.mask, f1.vecs, f1.mask, -1)
    np.testing.assert_array_equal(r.mask, mo)
    np.testing.assert_array_equal(r.vecs, o)
    for ang in (2.0, 4.0, 77.0):           # around the switch-over between the two paths, and a steep one
        g = of.Flow.from_transforms([['rotation', 1700, 900, ang]], [H, W], 's', rand_mask((H, W), 23))
        r = g.combine_with(f2.switch_ref('invalid'), 3)
        o, mo = oracle.compose3_raw(f2.vecs, f2.mask, g.vecs, g.mask, +1)
        np.testing.assert_array_equal(r.mask, mo)
        np.testing.assert_array_equal(r.vecs, o)


def test_config4_batch_of_pairs(gpu, oracle):
    """BASELINE config 4 in miniature: a batch of independent 1080 x 1920 pairs through ONE device launch
    (ofl_compose3_dev with batch > 1 on HBM-resident stacks) equals the per-pair oracle, and the shard of
    pairs a rank would own (sharding.shard) composes to the same results as the full batch."""
    import math
    of = gpu
    from oflibnumpy_amd import device as dev, sharding
    nat, lib = of.native, of.native.load()
    H, W, B = 1080, 1920, 6
    n = H * W
    pairs = []
    for i in range(B):
        ang = -30 + 60 * i / 255
        f1 = of.Flow.from_transforms([['rotation', W / 2, H / 2, ang]], [H, W], 't', rand_mask((H, W), i, 0.03))
        f2 = of.Flow.from_transforms([['translation', 40 * math.cos(i), 40 * math.sin(i)]], [H, W], 't')
        pairs.append((f1, f2))
    va, ma, vb, mb = (dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n))
    for i, (f1, f2) in enumerate(pairs):
        nat.check(lib.ofl_upload(va.ptr + i * n * 8, f1.vecs.ctypes.data, n * 8, None))
        nat.check(lib.ofl_upload(vb.ptr + i * n * 8, f2.vecs.ctypes.data, n * 8, None))
        m1, m2 = f1.mask.astype(np.uint8), f2.mask.astype(np.uint8)
        nat.check(lib.ofl_upload(ma.ptr + i * n, m1.ctypes.data, n, None))
        nat.check(lib.ofl_upload(mb.ptr + i * n, m2.ctypes.data, n, None))
        nat.check(lib.ofl_stream_sync(None))
    out, mout = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n)
    stats = dev.DeviceBuffer.zeros(32 * B)
    nat.check(lib.ofl_compose3_dev(va.ptr, ma.ptr, vb.ptr, mb.ptr, -1, H, W, B, out.ptr, mout.ptr, stats.ptr, 0, None))
    res = out.to_host((B, H, W, 2), np.float32)
    msk = mout.to_host((B, H, W), np.uint8).astype(bool)
    words = stats.to_host((B, 8), np.uint32)
    assert words[:, [0, 1, 4, 5, 6, 7]].all()
    for i, (f1, f2) in enumerate(pairs):
        o, m = oracle.compose3_raw(f1.vecs, f1.mask, f2.vecs, f2.mask, -1)
        np.testing.assert_array_equal(res[i], o)
        np.testing.assert_array_equal(msk[i], m)
    # a rank's shard launched on its own gives the same fields
    r = sharding.shard(B, 1, 2)
    sub, msub = dev.DeviceBuffer(len(r) * n * 8), dev.DeviceBuffer(len(r) * n)
    nat.check(lib.ofl_compose3_dev(va.ptr + r.start * n * 8, ma.ptr + r.start * n, vb.ptr + r.start * n * 8,
                                   mb.ptr + r.start * n, -1, H, W, len(r), sub.ptr, msub.ptr, None, 0, None))
    np.testing.assert_array_equal(sub.to_host((len(r), H, W, 2), np.float32), res[r.start:r.stop])


def test_config5_tiled_sintel_apply(gpu, oracle):
    """BASELINE config 5 geometry (tests/golden/sintel.flo: 10 x 20, u = r*c up to 171 px, v = 0) tiled to
    430 x 760 and wrapped as a 't' flow: image warp + mask propagation equal the oracle bit for bit, and the
    8-row bands a rank would own (sharding.row_band) tile the result without overlap."""
    import os
    of = gpu
    from oflibnumpy_amd import sharding
    flo = of.load_sintel(os.path.join(os.path.dirname(__file__), "golden", "sintel.flo"))
    big = np.tile(flo, (43, 38, 1))
    H, W = big.shape[:2]
    rng = np.random.default_rng(2)
    img = rng.random((H, W, 3), dtype=np.float32)
    tmask = rng.random((H, W)) > 0.1
    f = of.Flow(big, 't', rng.random((H, W)) > 0.05)
    w, v = f.apply(img, tmask, return_valid_area=True)
    ow, ov = oracle.OFlow(big, 't', f.mask).apply(img, tmask, return_valid_area=True)
    np.testing.assert_array_equal(w, ow)
    np.testing.assert_array_equal(v, ov)
    u8 = (img * 255).astype(np.uint8)
    np.testing.assert_array_equal(f.apply(u8), oracle.OFlow(big, 't', f.mask).apply(u8))
    bands = [sharding.row_band(H, r, 8) for r in range(8)]
    assert bands[0][0] == 0 and bands[-1][1] == H and all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
    # each rank's launch (ofl_gather_rows_dev): replicated image and target mask, its own rows of the flow
    from oflibnumpy_amd import device as dev
    dimg, dtm = dev.DeviceImage.from_host(img), dev.DeviceBuffer.from_host(tmask.astype(np.uint8))
    for world in (8, 3):
        parts, vparts = [], []
        for r in range(world):
            r0, r1 = sharding.row_band(H, r, world)
            fl = dev.DeviceBuffer.from_host(np.ascontiguousarray(big[r0:r1]))
            fm = dev.DeviceBuffer.from_host(np.ascontiguousarray(f.mask[r0:r1]).astype(np.uint8))
            d, vv = dev.gather_rows(dimg, r0, r1 - r0, fl, -1, smask=dtm, fmask_rows=fm, want_valid=True)
            parts.append(d.to_host())
            vparts.append(vv.to_host((r1 - r0, W), np.uint8).astype(bool))
        np.testing.assert_array_equal(np.concatenate(parts), ow)
        np.testing.assert_array_equal(np.concatenate(vparts), ov)
    odd = dev.DeviceImage.from_host(np.ascontiguousarray(img[:, :W - 1]))          # odd width: generic kernel
    ow_odd = oracle.OFlow(big[:, :W - 1], 't').apply(img[:, :W - 1])
    r0, r1 = 16, 203
    d, _ = dev.gather_rows(odd, r0, r1 - r0, dev.DeviceBuffer.from_host(np.ascontiguousarray(big[r0:r1, :W - 1])), -1)
    np.testing.assert_array_equal(d.to_host(), ow_odd[r0:r1])
    # as loaded ('s', Flow.from_sintel) the field goes through the Delaunay path: outputs of the REAL reference for the 40 x 80
    # version are compared in test_gpu_scatter_exact.py (::test_exact_path_matches_reference_outputs[sintel4x4]), the full
    # 4320 x 7680 field in test_gpu_fullsize.py; here: the valid area is valid_target(), values are convex combinations
    fs = of.Flow(big, 's')
    ws, vs = fs.apply(img, return_valid_area=True)
    np.testing.assert_array_equal(vs, fs.valid_target())
    assert np.isfinite(ws).all() and ws[vs].min() >= -1e-6 and ws[vs].max() <= 1 + 1e-6 and (ws[~vs] == 0).all()
    ws2, vs2 = fs.apply(img, return_valid_area=True)
    np.testing.assert_array_equal(ws2, ws)
    np.testing.assert_array_equal(vs2, vs)


def test_randomised_sweep_compose_and_gather(gpu, oracle):
    """Seeded sweep over ragged shapes (odd widths take the generic kernels, widths < 128 leave partial tiles),
    flows mixing sub-threshold, half-pixel, ordinary and far-out-of-range vectors, and random masks: compose
    (both refs, through the Flow API incl. its early exits) and the image gather stay bit-identical to the oracle."""
    of = gpu
    rng = np.random.default_rng(1234)
    for case in range(40):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 300))
        kind = rng.integers(0, 4, size=(h, w, 1))
        base = rng.standard_normal((h, w, 2)) * rng.choice([0.3, 3.0, 40.0])
        v = np.where(kind == 0, np.round(base * 2) / 2,                      # exact half / whole pixels
            np.where(kind == 1, base,
            np.where(kind == 2, rng.uniform(-9e-4, 9e-4, (h, w, 2)), base * 1e3))).astype(np.float32)
        if case % 7 == 0:
            v[...] = rng.uniform(-9e-4, 9e-4, (h, w, 2))                    # thresholded-zero sampling field
        if case % 11 == 0:
            v[...] = 0
        u = (rng.standard_normal((h, w, 2)) * 5).astype(np.float32)
        m1, m2 = rng.random((h, w)) > 0.2, rng.random((h, w)) > 0.2
        for ref in ('t', 's'):
            a, b = (u, v) if ref == 't' else (v, u)          # v always plays the sampling field
            fa, fb = of.Flow(a, ref, m1), of.Flow(b, ref, m2)
            got = fa.combine_with(fb, 3)
            want = oracle.OFlow(a, ref, m1).combine_with(oracle.OFlow(b, ref, m2), 3)
            np.testing.assert_array_equal(got.vecs, want.vecs, err_msg="case {} {}".format(case, ref))
            np.testing.assert_array_equal(got.mask, want.mask, err_msg="case {} {}".format(case, ref))
        c = int(rng.integers(1, 6))
        img = (rng.random((h, w, c)) * 255).astype(rng.choice([np.uint8, np.float32]))
        f = of.Flow(v, 't', m1)
        ow, ov = oracle.OFlow(v, 't', m1).apply(img, m2, return_valid_area=True)
        gw, gv = f.apply(img, m2, return_valid_area=True)
        np.testing.assert_array_equal(gw, ow, err_msg="case {} apply".format(case))
        np.testing.assert_array_equal(gv, ov, err_msg="case {} apply".format(case))


def test_maximum_dimensions(gpu, oracle):
    """cv2.remap's own limit (dims < SHRT_MAX): 32766 rows / columns are accepted and correct, 32767 is refused
    on both the device and the host entry points."""
    import ctypes
    of = gpu
    nat, lib = of.native, of.native.load()
    rng = np.random.default_rng(9)
    for shape in ((32766, 4), (2, 32766)):
        a = (rng.standard_normal(shape + (2,)) * 3).astype('f')
        b = (rng.standard_normal(shape + (2,)) * 3).astype('f')
        m1, m2 = rng.random(shape) > 0.1, rng.random(shape) > 0.1
        got = of.Flow(a, 't', m1).combine_with(of.Flow(b, 't', m2), 3)
        o, m = oracle.compose3_raw(a, m1, b, m2, -1)
        np.testing.assert_array_equal(got.vecs, o)
        np.testing.assert_array_equal(got.mask, m)
        img = rng.random(shape + (3,), dtype=np.float32)
        np.testing.assert_array_equal(of.apply_flow(b, img, 't'), oracle.apply_flow(b, img, 't'))
    rc = lib.ofl_compose3_dev(1, 1, 1, 1, -1, 32767, 4, 1, 1, 1, None, 0, None)
    assert rc == nat.E_INVALID and b"32766" in lib.ofl_last_error()


def test_rccl_broadcast_single_rank(gpu):
    """The RCCL binding of the sharded workload's one exchange step (ofl_comm_*), exercised with a
    communicator of one rank: library loading, id creation, init, broadcast on the engine's stream, destroy."""
    import ctypes
    of = gpu
    from oflibnumpy_amd import device as dev
    nat, lib = of.native, of.native.load()
    uid = np.zeros(128, np.uint8)
    nat.check(lib.ofl_comm_unique_id(uid.ctypes.data))
    assert uid.any()
    nat.check(lib.ofl_comm_init(uid.ctypes.data, 0, 1))
    try:
        payload = np.arange(1 << 16, dtype=np.uint8)
        buf = dev.DeviceBuffer.from_host(payload)
        nat.check(lib.ofl_comm_broadcast(buf.ptr, payload.nbytes, 0, None))
        np.testing.assert_array_equal(buf.to_host(payload.shape, np.uint8), payload)
        assert lib.ofl_comm_broadcast(buf.ptr, payload.nbytes, 3, None) == nat.E_INVALID     # root out of range
    finally:
        nat.check(lib.ofl_comm_destroy())
    assert lib.ofl_comm_broadcast(1, 16, 0, None) == nat.E_INVALID                            # no communicator


def test_combine_flows_batch_api(gpu, oracle):
    """Batch convenience API: one launch for all pairs a rank owns, per-pair early exits honoured, shards of two
    ranks together reproduce the single-rank result."""
    of = gpu
    shape = [96, 128]
    rng = np.random.default_rng(3)
    f1s, f2s = [], []
    for i in range(7):
        f1s.append(of.Flow.from_transforms([['rotation', 60, 50, -20 + 7 * i]], shape, 't', rand_mask(shape, i, 0.05)))
        f2s.append(of.Flow.from_transforms([['translation', 3 * i - 9, 2.5 * i]], shape, 't', rand_mask(shape, 50 + i, 0.05)))
    f1s[2] = of.Flow.zero(shape, 't')                      # self zero  -> the other operand comes back
    f2s[4] = of.Flow.zero(shape, 't')                      # flow zero  -> self comes back
    f2s[5] = of.Flow(np.full(shape + [2], 4e-4, 'f'), 't')  # sampling field below the threshold -> plain sum
    full = of.combine_flows_batch(f1s, f2s)
    assert [i for i, _ in full] == list(range(7))
    for i, r in full:
        want = f1s[i].combine_with(f2s[i], 3)
        np.testing.assert_array_equal(r.vecs, want.vecs)
        np.testing.assert_array_equal(r.mask, want.mask)
    assert full[2][1] is f2s[2] and full[4][1] is f1s[4]
    parts = of.combine_flows_batch(f1s, f2s, rank=0, world=2) + of.combine_flows_batch(f1s, f2s, rank=1, world=2)
    assert [i for i, _ in parts] == list(range(7))
    for (i, a), (_, b) in zip(parts, full):
        np.testing.assert_array_equal(a.vecs, b.vecs)
        np.testing.assert_array_equal(a.mask, b.mask)
    arr = of.combine_flows_batch([f.vecs for f in f1s[:2]], [f.vecs for f in f2s[:2]], ref='s')
    np.testing.assert_array_equal(arr[1][1].vecs, of.combine_flows(f1s[1].vecs, f2s[1].vecs, 3, 's'))


def test_lds_staged_variant_in_subprocess(gpu):
    """The A/B compose kernel variants of the EXPERIMENTS build (libofl_hip_exp.so, OFL_C3_VARIANT read once per process;
    the shipped library holds only the default kernel and reads no environment) produce the same bits as the oracle:
    small and large footprints (the latter exceed the LDS budget and take the in-kernel direct path), both references,
    odd tile remainders."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import oflibnumpy_amd as of
from oracle import np_oracle as O
rng = np.random.default_rng(5)
for shape, t1, t2 in (((300, 400), [['rotation', 200, 150, -30]], [['scaling', 100, 80, 0.8]]),
                      ((257, 130), [['scaling', 60, 60, 0.9]], [['rotation', 65, 128, 40]]),
                      ((96, 256), [['rotation', 10, 10, 5]], [['scaling', 128, 48, 3.5]]),       # footprint > LDS budget
                      ((33, 66), [['translation', 1.5, -2.25]], [['translation', 70.5, 0]])):
    for ref in 'ts':
        f1 = of.Flow.from_transforms(t1, list(shape), ref, rng.random(shape) > 0.1)
        f2 = of.Flow.from_transforms(t2, list(shape), ref, rng.random(shape) > 0.1)
        got = f1.combine_with(f2, 3)
        want = O.OFlow(f1.vecs, ref, f1.mask).combine_with(O.OFlow(f2.vecs, ref, f2.mask), 3)
        assert np.array_equal(got.vecs, want.vecs) and np.array_equal(got.mask, want.mask), (shape, ref)
print("lds-variant-ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # variant 1 = source tile staged in LDS, 2 = one-shot without the transposed gather, 0 = persistent grid;
    # the default (3, transposed gather for rotated sampling grids) is what every other test runs
    from oflibnumpy_amd import build_native
    if not os.path.exists(build_native.EXP_OUT):
        build_native.build_experiments()
    for variant in ("1", "2", "0"):
        env = dict(os.environ, OFL_C3_VARIANT=variant, OFL_LIB=build_native.EXP_OUT)
        p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "lds-variant-ok" in p.stdout, (variant, p.stderr[-2000:])


def test_general_gather_kernel_and_walk_tilings_in_subprocess(gpu):
    """What the product library runs by default has a twin in the EXPERIMENTS build: the paired gather kernel WITHOUT the folded
    (quantisation, arithmetic, rule) instantiations (OFL_G2_SPEC=0: one kernel for every combination, as before round 4), and the
    certified walk with XCD-banded tile orders / 64 x 4 tiles (OFL_WALK_TILING).  Same bits as the oracle ('t') and as the default
    order ('s', whose values the oracle pins to 1e-4 only): 8-bit, 16-bit and float images, 1 - 4 channels, with and without a mask."""
    import os
    import subprocess
    import sys
    code = r"""
import os, sys, numpy as np
sys.path.insert(0, %r)
import oflibnumpy_amd as of
from oracle import np_oracle as O
rng = np.random.default_rng(8)
shape = (131, 258)
tm = rng.random(shape) > 0.1
f = of.Flow.from_transforms([['rotation', 100, 60, 17], ['scaling', 100, 60, 0.9]], list(shape), 't', rng.random(shape) > 0.05)
o = O.OFlow(f.vecs, 't', f.mask)
for dtype in (np.uint8, np.int16, np.float32):
    for C in (1, 2, 3, 4):
        img = (rng.random(shape + (C,)) * 255).astype(dtype)
        for mask in (None, tm):
            got, gv = f.apply(img, mask, return_valid_area=True)
            want, wv = o.apply(img, mask, return_valid_area=True)
            assert np.array_equal(got, want) and np.array_equal(gv, wv), (dtype, C, mask is None)
fs = of.Flow.from_transforms([['rotation', 100, 60, 17], ['scaling', 100, 60, 0.9]], [200, 330], 's')
inv = fs.invert()
sw = fs.switch_ref()
np.save(os.environ['OFL_TEST_OUT'], np.concatenate([inv.vecs.ravel(), inv.mask.ravel().astype(np.float32), sw.vecs.ravel(), sw.mask.ravel().astype(np.float32)]))
print("twin-ok")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from oflibnumpy_amd import build_native
    import tempfile
    if not os.path.exists(build_native.EXP_OUT):
        build_native.build_experiments()
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, (spec, tiling) in enumerate((("1", "0"), ("0", "5"), ("1", "13"), ("0", "1"), ("1", "7"))):
            out = os.path.join(tmp, "o%d.npy" % k)
            env = dict(os.environ, OFL_G2_SPEC=spec, OFL_WALK_TILING=tiling, OFL_LIB=build_native.EXP_OUT, OFL_TEST_OUT=out)
            p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
            assert p.returncode == 0 and "twin-ok" in p.stdout, (spec, tiling, p.stderr[-2000:])
            outs.append(np.load(out))
    for o2 in outs[1:]:
        np.testing.assert_array_equal(outs[0], o2)


def test_device_resident_image_warp(gpu, oracle):
    """DeviceFlow.apply with an HBM-resident image (no PCIe between operations): 't' for uint8 / float32 with and
    without a target mask, 's' for float32, against the host API (which is pinned against the oracle above)."""
    of = gpu
    from oflibnumpy_amd import device as dev
    rng = np.random.default_rng(12)
    shape = (120, 168)
    tmask = rng.random(shape) > 0.1
    for ref in ('t', 's'):
        f = of.Flow.from_transforms([['rotation', 60, 80, 14], ['scaling', 50, 50, 0.92]], list(shape), ref, rng.random(shape) > 0.05)
        d = f.to_device()
        for dtype in ((np.uint8, np.float32) if ref == 't' else (np.float32,)):
            img = (rng.random(shape + (3,)) * 255).astype(dtype)
            dimg = dev.DeviceImage.from_host(img)
            for tm in (None, tmask):
                dtm = dev.DeviceBuffer.from_host(tm.astype(np.uint8)) if tm is not None else None
                warped, valid = d.apply(dimg, target_mask=dtm)
                hw, hv = f.apply(img, tm, return_valid_area=True)
                np.testing.assert_array_equal(warped.to_host(), hw)
                np.testing.assert_array_equal(valid.to_host(shape, np.uint8).astype(bool), hv)
    z = of.Flow.zero(list(shape), 't').to_device()
    w, v = z.apply(dimg)
    assert w is dimg and v.to_host(shape, np.uint8).all()


def test_resize_matches_oracle(gpu, oracle):
    """Flow.resize / resize_flow / DeviceFlow.resize: bit-exact against the oracle (which reproduces the
    reference's known answers, tests/test_oracle.py) for up-, down-, anisotropic and non-integer scales, odd
    sizes, speckled masks; plus the reference's own mask known answer (tests/test_flow_class.py:380-389)."""
    of, O = gpu, oracle
    rng = np.random.default_rng(21)
    for shape in ((20, 10), (37, 53), (64, 129), (1, 7), (9, 1), (33, 300), (12, 128)):
        vecs = (rng.standard_normal(shape + (2,)) * 5).astype(np.float32)
        mask = rng.random(shape) > 0.3
        f = of.Flow(vecs, 's', mask)
        for scale in (.2, .5, 1, 1.5, 2, 10, 1 / 3, (0.5, 2), (2, 0.5), [1.7, 0.9], 3):
            fy, fx = (scale, scale) if isinstance(scale, (int, float)) else scale
            if int(np.rint(shape[0] * fy)) <= 0 or int(np.rint(shape[1] * fx)) <= 0:
                with pytest.raises(ValueError):
                    f.resize(scale)
                continue
            ev, em = O.resize_flow(vecs, scale, mask)
            r = f.resize(scale)
            assert r.ref == 's'
            np.testing.assert_array_equal(r.vecs, ev)
            np.testing.assert_array_equal(r.mask, em)
            np.testing.assert_array_equal(of.resize_flow(vecs, scale), ev)
            dv, dm = f.to_device().resize(scale).to_host()
            np.testing.assert_array_equal(dv, ev)
            np.testing.assert_array_equal(dm, em)
    small, large = (20, 40), (30, 80)
    m_small = np.ones(small, bool)
    m_small[:6, :20] = False
    m_large = np.ones(large, bool)
    m_large[:9, :40] = False
    fl = of.Flow.from_transforms([['rotation', 0, 0, 30]], small, 't', m_small).resize((1.5, 2))
    np.testing.assert_array_equal(fl.mask, m_large)
    big = of.Flow.from_transforms([['rotation', 0, 0, 30]], (150, 240), 't')
    np.testing.assert_allclose(big.resize(1 / 3).vecs, of.Flow.from_transforms([['rotation', 0, 0, 30]], (50, 80), 't').vecs,
                               atol=1, rtol=.1)
    # full size: 4K -> 1080p -> 4K keeps an affine field (interior) and runs the big grid
    f4k = of.Flow.from_transforms([['scaling', 1000, 800, 0.9]], (2160, 3840), 't')
    half = f4k.resize(0.5)
    assert half.shape == (1080, 1920)
    np.testing.assert_allclose(half.vecs[2:-2, 2:-2], of.Flow.from_transforms([['scaling', 500 - 0.25, 400 - 0.25, 0.9]], (1080, 1920), 't').vecs[2:-2, 2:-2], atol=2e-3)


def test_is_zero_and_get_padding_on_device(gpu, oracle):
    """Flow.is_zero / is_zero_flow (reference tests/test_flow_class.py:1007-1024, tests/test_utils.py:520-540) and
    Flow.get_padding (tests/test_flow_class.py:982-1004 known answers) run as device reductions."""
    of = gpu
    shape = (10, 10)
    mask = np.ones(shape, bool)
    mask[0, 0] = False
    v = np.zeros(shape + (2,))
    v[0, 0] = 10
    f = of.Flow(v, mask=mask)
    assert f.is_zero() is True and f.is_zero(masked=True) is True and f.is_zero(masked=False) is False
    assert f.is_zero(thresholded=False) is True
    v = np.zeros(shape + (2,), np.float32)
    v[1, 1] = [9e-4, -9e-4]
    assert of.is_zero_flow(v) and of.is_zero_flow(v, True) and not of.is_zero_flow(v, False)
    v[1, 1] = [1e-3, 0]
    assert not of.is_zero_flow(v)
    # zero flow: the target itself comes back (reference utils.py:215-216)
    tgt = np.ones((10, 10), np.float32)
    assert of.apply_flow(np.zeros((10, 10, 2)), tgt, 't') is tgt
    assert of.apply_flow(np.full((10, 10, 2), 9e-4), tgt, 's') is tgt
    w, vv = of.Flow.zero((10, 10)).apply(tgt, return_valid_area=True)
    assert np.array_equal(w, tgt) and vv.all()
    z = of.Flow.zero((20, 20), 's')
    assert z.switch_ref().ref == 't' and not z.switch_ref().vecs.any()       # exact-zero short cut (flow_class.py:716-718)
    f = of.Flow.from_transforms([['rotation', 0, 0, 45]], (7, 7), 's')
    assert f.get_padding() == [5, 0, 0, 3]                       # reference test_get_padding
    assert of.get_flow_padding(f.vecs, f.ref) == [5, 0, 0, 3]
    ft = of.Flow.from_transforms([['rotation', 0, 0, 45]], (7, 7), 't')
    assert ft.get_padding() == [0, 3, 5, 0]
    m = np.ones((7, 7), bool)
    m[:, 4:] = False
    assert of.Flow.from_transforms([['rotation', 0, 0, 45]], (7, 7), 's', m).get_padding() == [3, 0, 0, 1]
    with pytest.raises(ValueError):
        of.Flow(ft.vecs, 't', np.zeros((7, 7), bool)).get_padding()
    # restated reference arithmetic (float32 positions) on ragged shapes, both references, random masks
    rng = np.random.default_rng(5)
    for shape in ((37, 53), (128, 257), (1, 9), (600, 801)):
        for ref in ('t', 's'):
            vecs = (rng.standard_normal(shape + (2,)) * 7).astype(np.float32)
            vecs[rng.random(shape) < 0.3] *= 1e-4                 # below the threshold
            msk = rng.random(shape) > 0.4
            fl = of.Flow(vecs, ref, msk)
            t = vecs.copy()
            t[(t < 1e-3) & (t > -1e-3)] = 0
            if ref == 's':
                t *= -1
            yy, xx = np.mgrid[:shape[0], :shape[1]]
            t[..., 0] -= xx
            t[..., 1] -= yy
            t *= -1
            exp = [max(-np.min(t[msk, 1]), 0), max(np.max(t[msk, 1]) - (shape[0] - 1), 0),
                   max(-np.min(t[msk, 0]), 0), max(np.max(t[msk, 0]) - (shape[1] - 1), 0)]
            assert fl.get_padding() == [int(np.ceil(p)) for p in exp]


def test_c_abi_rejects_bad_arguments(gpu):
    """Every entry validates its arguments on the host and returns OFL_E_INVALID (-1) with a message instead of
    launching a kernel on mismatched shapes (a faulting kernel can take the whole node down)."""
    import ctypes
    of = gpu
    from oflibnumpy_amd import device as dev
    nat = of.native
    lib = nat.load()
    n = 64 * 64
    buf = dev.DeviceBuffer(n * 16)
    m = dev.DeviceBuffer(n)
    p, q = buf.ptr, m.ptr
    INVALID = -1
    assert lib.ofl_compose3_dev(p, q, p, q, -1, 0, 64, 1, p, q, None, 0, None) == INVALID
    assert lib.ofl_compose3_dev(p, q, p, q, 2, 64, 64, 1, p, q, None, 0, None) == INVALID
    assert lib.ofl_compose3_dev(None, q, p, q, -1, 64, 64, 1, p, q, None, 0, None) == INVALID
    assert lib.ofl_gather_bilinear_dev(p, nat.F32, 1, 64, 64, p, 64, 64, 1, 0, -1, None, None, p, None, 0, 0, 0, None) == INVALID   # flow does not fit
    assert lib.ofl_gather_bilinear_dev(p, 9, 1, 64, 64, p, 64, 64, 0, 0, -1, None, None, p, None, 0, 0, 0, None) == INVALID       # dtype
    assert lib.ofl_gather_bilinear_dev(p, nat.F32, 1, 40000, 64, p, 64, 64, 0, 0, -1, None, None, p, None, 0, 0, 0, None) == INVALID  # > int16 coordinates
    assert lib.ofl_gather_rows_dev(p, nat.F32, 1, 64, 64, 60, 8, p, -1, None, None, p, None, 0, 0, 0, None) == INVALID
    assert lib.ofl_scatter_linear_dev(p, 1, 0, None, p, 2, None, 64, 64, None, p, q, 0, p, 16, None, None) == INVALID              # workspace too small
    assert lib.ofl_scatter_linear_dev(p, 1, 0, None, p, 2, None, 64, 64, None, p, q, 7, p, n * 16, None, None) == INVALID          # valid_rule
    assert lib.ofl_scatter_rows_dev(p, 1, 0, None, p, 2, None, 64, 64, 60, 8, p, q, 0, p, n * 16, None, None) == INVALID
    assert lib.ofl_resize_flow_dev(p, q, 64, 64, 32, 32, 2.0, 2.0, 0.5, 0.5, p, None, None) == INVALID                              # mask without mout
    assert lib.ofl_resize_flow_dev(p, None, 64, 64, 0, 32, 2.0, 2.0, 0.5, 0.5, p, None, None) == INVALID
    assert lib.ofl_convert_dev(p, nat.U8, p, nat.I16, n, None) == INVALID
    assert lib.ofl_flow_extent_dev(p, None, 64, 64, 0, 1e-3, p, None) == INVALID
    assert b"ofl_flow_extent" in lib.ofl_last_error()
    with pytest.raises(RuntimeError):
        nat.check(lib.ofl_axpy_dev(None, None, None, None, 1.0, n, None, None, None))


def test_sintel_flo_roundtrip_through_pinned_buffers(gpu, tmp_path):
    """the .flo reader of the reference (utils.py:447-470) on device-pinned buffers, plus a writer.  The
    reference's own fixture goes disk -> pinned host buffer -> HBM (asynchronous upload), is warped on the device, written
    back through a pinned buffer and re-read by the plain reader: bytes survive, and the device copy equals load_sintel."""
    import os
    of = gpu
    from oflibnumpy_amd import device as dev
    path = os.path.join(os.path.dirname(__file__), "golden", "sintel.flo")
    want = of.load_sintel(path)
    d, pin = dev.load_sintel_device(path)
    assert d.ref == 's' and d.shape == want.shape[:2]
    vecs, mask = d.to_host()
    np.testing.assert_array_equal(vecs, want)
    assert mask.all()
    np.testing.assert_array_equal(pin.array, want)
    out = str(tmp_path / "copy.flo")
    dev.save_sintel_device(out, d)
    assert open(out, 'rb').read() == open(path, 'rb').read()
    # a result computed on the device, saved and re-read
    r = (d + d)
    out2 = str(tmp_path / "twice.flo")
    dev.save_sintel_device(out2, r)
    np.testing.assert_array_equal(of.load_sintel(out2), want * 2)
    of.save_sintel(str(tmp_path / "host.flo"), want)
    assert open(str(tmp_path / "host.flo"), 'rb').read() == open(path, 'rb').read()
    with pytest.raises(ValueError):
        dev.load_sintel_device(os.path.join(os.path.dirname(__file__), "golden", "sintel_wrong.flo"))


def test_gather_batch_equals_field_by_field(gpu):
    """ofl_gather_bilinear_batch_dev: B fields in one launch == B launches of ofl_gather_bilinear_dev, bit for bit -- own and
    shared source image / target mask, three dtypes, odd and even widths (paired and one-pixel kernels), validity-only
    launches (C = 0), and DeviceFlowBatch.apply_images == DeviceFlow.apply_image field by field"""
    from oflibnumpy_amd import device as dev
    from oflibnumpy_amd.batch import DeviceFlowBatch
    of, nat = gpu, gpu.native
    rng = np.random.default_rng(77)
    B = 5
    for h, w in ((72, 200), (41, 67)):
        n = h * w
        flows = [(rng.standard_normal((h, w, 2)) * (0.5 + 3 * i)).astype(np.float32) + np.float32(1.5 * i) for i in range(B)]
        fmasks = [(rng.random((h, w)) > 0.1).astype(np.uint8) for _ in range(B)]
        smasks = [(rng.random((h, w)) > 0.1).astype(np.uint8) for _ in range(B)]
        fb = dev.DeviceBuffer.from_host(np.stack(flows))
        mb = dev.DeviceBuffer.from_host(np.stack(fmasks))
        sb = dev.DeviceBuffer.from_host(np.stack(smasks))
        for dt, C, kw in ((np.float32, 3, {}), (np.uint8, 3, dict(arith=nat.ARITH_NATIVE, rule=nat.RULE_GE_HALF)),
                          (np.uint16, 1, dict(rule=nat.RULE_GT_HALF)), (np.float32, 2, {})):
            imgs = [(rng.random((h, w, C)) * (255 if dt != np.float32 else 1)).astype(dt) for _ in range(B)]
            ib = dev.DeviceBuffer.from_host(np.stack(imgs))
            for shared in (False, True):
                dst, valid = dev.gather_bilinear_batch(ib, dt, C, h, w, B, fb, -1, smask=sb, fmask=mb, valid=True,
                                                       shared_src=shared, shared_smask=shared, **kw)
                got = dst.to_host((B, h, w, C), dt)
                gotv = valid.to_host((B, h, w), np.uint8)
                for i in range(B):
                    j = 0 if shared else i
                    d1, v1 = dev.gather_bilinear(dev.DeviceImage.from_host(imgs[j]), dev.DeviceBuffer.from_host(flows[i]), (h, w), -1,
                                                 smask=dev.DeviceBuffer.from_host(smasks[j]), fmask=dev.DeviceBuffer.from_host(fmasks[i]),
                                                 want_valid=True, **kw)
                    assert np.array_equal(got[i].view(np.uint8), d1.to_host().view(np.uint8)), (h, w, dt, C, shared, i)
                    assert np.array_equal(gotv[i], v1.to_host((h, w), np.uint8)), (h, w, dt, C, shared, i)
        # validity only (valid_target of B fields at once)
        _, valid = dev.gather_bilinear_batch(None, np.uint8, 0, h, w, B, fb, -1, fmask=mb, valid=True)
        gotv = valid.to_host((B, h, w), np.uint8)
        for i in range(B):
            v1 = dev.gather_valid_only(h, w, dev.DeviceBuffer.from_host(flows[i]), (h, w), -1, fmask=dev.DeviceBuffer.from_host(fmasks[i]))
            assert np.array_equal(gotv[i], v1.to_host((h, w), np.uint8))
        # the Flow-level batch: uint8 images with the default (int8) mask take the float / round-half-even arithmetic
        fl = [of.Flow(flows[i], 't', fmasks[i].astype(bool)) for i in range(B)]
        batch = DeviceFlowBatch.from_flows(fl)
        img8 = [(rng.random((h, w, 3)) * 255).astype(np.uint8) for _ in range(B)]
        for masks in (None, sb):
            wb, vb = batch.apply_images(dev.DeviceBuffer.from_host(np.stack(img8)), np.uint8, 3, target_masks=masks)
            got, gotv = wb.to_host((B, h, w, 3), np.uint8), vb.to_host((B, h, w), np.uint8)
            for i in range(B):
                tm = None if masks is None else dev.DeviceBuffer.from_host(smasks[i])
                d1, v1 = fl[i].to_device().apply_image(dev.DeviceImage.from_host(img8[i]), target_mask=tm)
                assert np.array_equal(got[i], d1.to_host()) and np.array_equal(gotv[i], v1.to_host((h, w), np.uint8))
    lib = nat.load()
    assert lib.ofl_gather_bilinear_batch_dev(ib.ptr, 0, nat.F32, 2, h, w, 0, fb.ptr, h, w, 0, 0, -1, None, 0, None, dst.ptr, None, 0, 0, 0, None) == nat.E_INVALID


def test_compose3_on_packed_mask_planes_is_bit_identical(gpu):
    """ofl_compose3_bits_dev (masks as one bit per pixel, device-resident chains) == ofl_compose3_dev once unpacked: vectors, valid
    mask and flag words, bit for bit -- axis-aligned and rotated sampling grids (the transposed gather), samples outside the
    source, ragged tile edges (W not a multiple of 32 or 128, H not of 8), both signs, batches; pack / unpack round trips for
    any width; DeviceFlowBatch composes packed batches on their planes."""
    from oflibnumpy_amd import device as dev
    from oflibnumpy_amd.batch import DeviceFlowBatch
    of, nat = gpu, gpu.native
    lib = nat.load()
    rng = np.random.default_rng(5)
    for h, w in ((37, 70), (64, 128), (130, 258), (96, 200)):
        m = (rng.random((3, h, w)) > 0.2).astype(np.uint8)
        back = dev.mask_unpack(dev.mask_pack(dev.DeviceBuffer.from_host(m), h, w, 3), h, w, 3).to_host((3, h, w), np.uint8)
        assert np.array_equal(back, m)
    m = (rng.random((2, 9, 45)) > 0.5).astype(np.uint8)          # odd width: the planes exist, the compose kernel refuses
    assert np.array_equal(dev.mask_unpack(dev.mask_pack(dev.DeviceBuffer.from_host(m), 9, 45, 2), 9, 45, 2).to_host((2, 9, 45), np.uint8), m)
    bits = dev.mask_pack(dev.DeviceBuffer.from_host(m), 9, 45, 2)
    v = dev.DeviceBuffer.zeros(2 * 9 * 45 * 8)
    assert lib.ofl_compose3_bits_dev(v.ptr, bits.ptr, v.ptr, bits.ptr, -1, 9, 45, 2, v.ptr, bits.ptr, None, None) == nat.E_INVALID
    B = 3
    for h, w in ((37, 70), (72, 256), (130, 258), (61, 392)):
        n = h * w
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        samplers = {
            "shift": np.stack([np.full((h, w), 3.3, np.float32), np.full((h, w), -2.7, np.float32)], -1),
            "rotated": of.from_transforms([['rotation', w / 2, h / 2, -30]], [h, w], 't'),
            "scaled": of.from_transforms([['scaling', w / 3, h / 3, 0.7]], [h, w], 't'),
            "noise": (rng.standard_normal((h, w, 2)) * 6).astype(np.float32),
        }
        for name, fbv in samplers.items():
            for sign in (-1, 1):
                fa = (rng.standard_normal((B, h, w, 2)) * 2).astype(np.float32)
                fb = np.stack([fbv + np.float32(0.01 * i) for i in range(B)]).astype(np.float32)
                if name == "noise":
                    fb[1] = 0.0                                  # a zero sampling field: flag words 4 .. 7 stay clear
                ma = (rng.random((B, h, w)) > 0.1).astype(np.uint8)
                mb = (rng.random((B, h, w)) > 0.1).astype(np.uint8)
                d = {k: dev.DeviceBuffer.from_host(a) for k, a in (("fa", fa), ("fb", fb), ("ma", ma), ("mb", mb))}
                out, mout, st = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer.zeros(32 * B)
                nat.check(lib.ofl_compose3_dev(d["fa"].ptr, d["ma"].ptr, d["fb"].ptr, d["mb"].ptr, sign, h, w, B, out.ptr, mout.ptr, st.ptr, 0, None))
                ba, bb = dev.mask_pack(d["ma"], h, w, B), dev.mask_pack(d["mb"], h, w, B)
                out2, bo, st2 = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer.zeros(dev.mask_bits_bytes(h, w, B)), dev.DeviceBuffer.zeros(32 * B)
                dev.compose3_bits_launch(d["fa"], ba, d["fb"], bb, sign, (h, w), out2, bo, st2, batch=B)
                tag = (h, w, name, sign)
                assert np.array_equal(out2.to_host((B, h, w, 2), np.float32).view(np.uint32), out.to_host((B, h, w, 2), np.float32).view(np.uint32)), tag
                assert np.array_equal(dev.mask_unpack(bo, h, w, B).to_host((B, h, w), np.uint8), mout.to_host((B, h, w), np.uint8)), tag
                assert np.array_equal(st2.to_host((B, 8), np.uint32), st.to_host((B, 8), np.uint32)), tag
    # the batch class: packed in, packed out, same fields
    h, w = 72, 256
    f1 = [of.Flow((rng.standard_normal((h, w, 2)) * 2).astype(np.float32), 't', rng.random((h, w)) > 0.1) for _ in range(4)]
    f2 = [of.Flow(of.from_transforms([['rotation', 100, 30, 10 + i]], [h, w], 't'), 't', rng.random((h, w)) > 0.1) for i in range(4)]
    plain, words, _ = DeviceFlowBatch.from_flows(f1).compose3(DeviceFlowBatch.from_flows(f2))
    packed, words2, _ = DeviceFlowBatch.from_flows(f1, packed=True).compose3(DeviceFlowBatch.from_flows(f2, packed=True))
    assert packed.packed and np.array_equal(words, words2)
    for a, b in zip(plain.to_flows(), packed.to_flows()):
        assert np.array_equal(a.vecs.view(np.uint32), b.vecs.view(np.uint32)) and np.array_equal(a.mask, b.mask)
