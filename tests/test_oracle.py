"""CPU suite, part 1: pins the oracle (oracle/) against
  (a) outputs of the REAL reference captured in tests/golden/ref_scipy_paths.npz (SciPy-backed paths),
  (b) the reference's own known-answer tests for the cv2.remap-backed paths
      (/root/reference/tests/test_flow_class.py:852-980, test_utils.py:277-283, :1020-1057).
"""
import numpy as np
import pytest
from scipy import ndimage


def _run_case(O, g, tag):
    op = tag.split('/')[0]
    f = O.OFlow(g[tag + '/in_vecs'], str(g[tag + '/in_ref']), g[tag + '/in_mask'])
    if op == 'invert':
        return f.invert()
    if op == 'switch_ref':
        return f.switch_ref()
    if op in ('valid_target', 'valid_target_nomask'):
        return f.valid_target(op == 'valid_target')
    if op in ('valid_source', 'valid_source_nomask'):
        return f.valid_source(op == 'valid_source')
    if op in ('apply_img', 'apply_img_nomask'):
        return f.apply(g['img_f32'], return_valid_area=True, consider_mask=(op == 'apply_img'))
    if op == 'apply_u8':
        return f.apply(g['img_u8'])
    if op.startswith('combine2'):
        return f.combine_with(O.OFlow(g[tag + '/in2_vecs'], f.ref, g[tag + '/in2_mask']), 2)
    if op.startswith('disc_'):
        name = tag.split('/')[1]
        if op == 'disc_apply':
            return f.apply(g['disc/' + name + '/img'], return_valid_area=True)
        return f.invert() if op == 'disc_invert' else f.valid_target()
    if op == 'k7':
        n = tag.split('/')[1]
        fn = f.valid_target if n.startswith('valid_target') else f.valid_source
        return fn(not n.endswith('nomask'))
    raise KeyError(tag)


def golden_tags(g):
    return sorted({k.rsplit('/', 1)[0] for k in g.files if '/' in k and not k.startswith(('track/', 'disc'))})


def disc_tags(g):
    """Discontinuous fields (a block moving over a static background): disc_apply/..., disc_invert/..., disc_valid_target/..."""
    return sorted({k.rsplit('/', 1)[0] for k in g.files if k.startswith('disc_')})


def test_oracle_matches_reference_outputs(oracle, golden):
    """Every captured reference output is reproduced bit for bit by the restated algebra."""
    tags = golden_tags(golden) + disc_tags(golden)
    assert len(tags) >= 86
    for tag in tags:
        r = _run_case(oracle, golden, tag)
        if isinstance(r, oracle.OFlow):
            np.testing.assert_array_equal(r.vecs, golden[tag + '/out_vecs'], err_msg=tag)
            np.testing.assert_array_equal(r.mask, golden[tag + '/out_mask'], err_msg=tag)
            assert r.ref == str(golden[tag + '/out_ref']), tag
        elif isinstance(r, tuple):
            np.testing.assert_array_equal(r[0], golden[tag + '/out'], err_msg=tag)
            np.testing.assert_array_equal(r[1], golden[tag + '/out_valid'], err_msg=tag)
        else:
            np.testing.assert_array_equal(r, golden[tag + '/out'], err_msg=tag)


def delaunay_tags(g):
    return sorted({k.split('/')[0] for k in g.files})


def test_oracle_matches_reference_outputs_round2(oracle, golden2):
    """The round-2 cases (folded tiled Sintel field of BASELINE config 5 as loaded, holes with image values, curved
    borders, sheared cells, speckled point masks, generic affine fields): bit for bit."""
    tags = delaunay_tags(golden2)
    assert len(tags) == 10 and 'sintel4x4' in tags
    for tag in tags:
        f = oracle.OFlow(golden2[tag + '/in_vecs'], 's', golden2[tag + '/in_mask'])
        w, v = f.apply(golden2[tag + '/img'], return_valid_area=True)
        np.testing.assert_array_equal(w, golden2[tag + '/apply'], err_msg=tag)
        np.testing.assert_array_equal(v, golden2[tag + '/apply_valid'], err_msg=tag)
        np.testing.assert_array_equal(f.valid_target(), golden2[tag + '/valid_target'], err_msg=tag)
        r = f.invert()
        np.testing.assert_array_equal(r.vecs, golden2[tag + '/invert_vecs'], err_msg=tag)
        np.testing.assert_array_equal(r.mask, golden2[tag + '/invert_mask'], err_msg=tag)


def B(rows):
    return np.array(rows).astype(bool)


# known-answer matrices of the reference's tests (rotation by 45 deg about (0, 0), 7 x 7)
K7_VALID_TARGET_T = B([[1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1, 1], [0, 0, 0, 1, 1, 1, 0],
                       [0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]])
K7_VALID_TARGET_T_MASKED = B([[1, 1, 1, 1, 0, 0, 0], [0, 1, 1, 1, 0, 0, 0], [0, 0, 1, 1, 0, 0, 0], [0, 0, 0, 1, 1, 1, 0],
                              [0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]])
K7_VALID_SOURCE_S = B([[1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0, 0],
                       [1, 1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0]])
K7_VALID_SOURCE_S_MASKED = B([[1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0], [1, 1, 1, 1, 0, 0, 0],
                              [0, 0, 0, 1, 1, 0, 0], [0, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0]])
K7_VALID_TARGET_S = K7_VALID_TARGET_T
K7_VALID_TARGET_S_MASKED_CONSIDER = B([[1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 1, 1, 1], [0, 0, 1, 1, 1, 1, 1]] + [[0] * 7] * 4)
K7_VALID_TARGET_S_MASKED = B([[1, 1, 1, 1, 1, 1, 1], [0, 1, 1, 1, 0, 0, 1], [0, 0, 1, 0, 0, 0, 0]] + [[0] * 7] * 4)
K7_VALID_SOURCE_T = K7_VALID_SOURCE_S
K7_VALID_SOURCE_T_MASKED_CONSIDER = B([[1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0]] + [[1, 1, 1, 0, 0, 0, 0]] * 5)
K7_VALID_SOURCE_T_MASKED = B([[1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0], [1, 1, 1, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0],
                              [1, 0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0, 0], [1, 1, 0, 0, 0, 0, 0]])


def k7_flows(make):
    t, s7 = [['rotation', 0, 0, 45]], (7, 7)
    ms = np.ones(s7, bool)
    ms[4:, :3] = False
    mt = np.ones(s7, bool)
    mt[:3, 4:] = False
    return make(t, s7, 's'), make(t, s7, 's', ms), make(t, s7, 't'), make(t, s7, 't', mt)


def test_oracle_known_answer_masks(oracle):
    f_s, f_sm, f_t, f_tm = k7_flows(oracle.from_transforms)
    np.testing.assert_array_equal(f_t.valid_target(), K7_VALID_TARGET_T)
    np.testing.assert_array_equal(f_tm.valid_target(), K7_VALID_TARGET_T_MASKED)
    np.testing.assert_array_equal(f_s.valid_source(), K7_VALID_SOURCE_S)
    np.testing.assert_array_equal(f_sm.valid_source(), K7_VALID_SOURCE_S_MASKED)
    np.testing.assert_array_equal(f_s.valid_target(), K7_VALID_TARGET_S)
    np.testing.assert_array_equal(f_sm.valid_target(), K7_VALID_TARGET_S_MASKED_CONSIDER)
    np.testing.assert_array_equal(f_sm.valid_target(False), K7_VALID_TARGET_S_MASKED)
    np.testing.assert_array_equal(f_t.valid_source(), K7_VALID_SOURCE_T)
    np.testing.assert_array_equal(f_tm.valid_source(), K7_VALID_SOURCE_T_MASKED_CONSIDER)
    np.testing.assert_array_equal(f_tm.valid_source(False), K7_VALID_SOURCE_T_MASKED)


def test_oracle_integer_translation_exact(oracle):
    """reference tests/test_utils.py:277-283: both refs equal scipy.ndimage.shift for an integer shift."""
    img = (np.random.default_rng(0).random((96, 120, 3)) * 255).astype(np.uint8)
    for ref in ('t', 's'):
        f = oracle.from_transforms([['translation', 10, 20]], img.shape[:2], ref)
        np.testing.assert_array_equal(oracle.apply_flow(f.vecs, img, ref), ndimage.shift(img, [20, 10, 0]))


@pytest.mark.parametrize("ref", ['s', 't'])
def test_oracle_mode3_analytic(oracle, ref):
    """reference tests/test_flow_class.py:1050-1057: f1 (+) f2 equals the analytic composite inside the masks."""
    shape = (160, 200)
    tr = [['rotation', 80.5, 90.5, -30], ['scaling', 40, 40, 0.8]]
    f1, f2, f3 = (oracle.from_transforms(t, shape, ref) for t in (tr[:1], tr[1:], tr))
    r = f1.combine_with(f2, 3)
    m = r.mask & f3.mask
    assert m.sum() > 1000
    np.testing.assert_allclose(r.vecs[m], f3.vecs[m], atol=5e-2)
    # the fused closed form is the same function as the object algebra
    if ref == 't':
        o, mo = oracle.compose3_raw(f1.vecs, f1.mask, f2.vecs, f2.mask, -1)
    else:
        o, mo = oracle.compose3_raw(f2.vecs, f2.mask, f1.vecs, f1.mask, +1)
    np.testing.assert_array_equal(o, r.vecs)
    np.testing.assert_array_equal(mo, r.mask)


def test_oracle_zero_predicates(oracle):
    v = np.zeros((10, 12, 2), np.float32)
    assert oracle.is_zero_raw(v, None, True) and oracle.is_zero_raw(v, None, False)
    v[3, 4, 1] = 5e-4
    assert oracle.is_zero_raw(v, None, True) and not oracle.is_zero_raw(v, None, False)
    v[3, 4, 1] = 1e-3          # float32(1e-3) is not < float32(1e-3)
    assert not oracle.is_zero_raw(v, None, True)
    m = np.ones((10, 12), bool)
    m[3, 4] = False
    assert oracle.is_zero_raw(v, m, True) and oracle.is_zero_raw(v, m, False)
    assert oracle.is_zero_flow(v, True) == oracle.is_zero_raw(v, None, True)


def test_oracle_track_pts_matches_reference(oracle, golden):
    """Sparse point tracking (utils.py:547-622) against outputs of the real reference."""
    g = golden
    pf, pi = g['track/pts_f'], g['track/pts_i']
    n = 0
    for name in ('rot', 'wob'):
        for ref in ('s', 't'):
            tag = 'track/{}_{}'.format(name, ref)
            flow = g[tag + '/flow']
            np.testing.assert_array_equal(oracle.track_pts(flow, ref, pf), g[tag + '/float'])
            r = oracle.track_pts(flow, ref, pf, int_out=True)
            np.testing.assert_array_equal(r, g[tag + '/float_int_out'])
            assert r.dtype == g[tag + '/float_int_out'].dtype
            if ref == 's':
                np.testing.assert_array_equal(oracle.track_pts(flow, ref, pi), g[tag + '/int'])
                np.testing.assert_array_equal(oracle.track_pts(flow, ref, pf, s_exact_mode=True), g[tag + '/exact'])
            n += 1
    assert n == 4


def test_oracle_resize_reference_known_answers(oracle):
    """Restates the reference's resize tests on the oracle: the exact mask of tests/test_flow_class.py:380-389,
    the shapes and corner vectors of tests/test_utils.py:472-496, and the field test of :498-504."""
    O = oracle
    small, large = (20, 40), (30, 80)
    m_small = np.ones(small, bool)
    m_small[:6, :20] = False
    m_large = np.ones(large, bool)
    m_large[:9, :40] = False
    f = O.from_transforms([['rotation', 0, 0, 30]], small, 't', m_small).resize((1.5, 2))
    assert f.vecs.shape == large + (2,) and f.ref == 't'
    np.testing.assert_array_equal(f.mask, m_large)

    shape = [20, 10]
    flow = O.from_transforms([['rotation', 30, 50, 30]], shape, 's').vecs
    for scale in (.2, .5, 1, 1.5, 2, 10):
        r = O.resize_flow(flow, scale)
        np.testing.assert_array_equal(r.shape[:2], scale * np.array(shape))
        np.testing.assert_allclose(r[0, 0], flow[0, 0] * scale, rtol=.1)
    np.testing.assert_array_equal(O.resize_flow(flow, 1), flow)                   # identity is exact
    for scale in ([.5, 2], (2, .5)):
        r = O.resize_flow(flow, scale)
        np.testing.assert_array_equal(r.shape[:2], np.array(scale) * np.array(shape))
        np.testing.assert_allclose(r[0, 0], flow[0, 0] * np.array(scale)[::-1], rtol=.1)
    f_small = O.from_transforms([['rotation', 0, 0, 30]], (50, 80), 't').vecs
    f_large = O.from_transforms([['rotation', 0, 0, 30]], (150, 240), 't').vecs
    np.testing.assert_allclose(O.resize_flow(f_large, 1 / 3), f_small, atol=1, rtol=.1)
    # an affine field is reproduced by linear interpolation away from the replicated border: resizing by an
    # integer factor k maps output node d to source (d + 0.5) / k - 0.5
    aff = O.from_transforms([['rotation', 10, 20, 25], ['scaling', 0, 0, 1.1]], (40, 60), 't').vecs
    up = O.resize_flow(aff, 2)
    yy, xx = np.mgrid[:80, :120]
    sy, sx = (yy + 0.5) / 2 - 0.5, (xx + 0.5) / 2 - 0.5
    inner = (sy >= 0) & (sy <= 39) & (sx >= 0) & (sx <= 59)
    m = O.matrix_from_transforms([['rotation', 10, 20, 25], ['scaling', 0, 0, 1.1]])
    inv = np.linalg.inv(m)
    expect_u = (sx - (inv[0, 0] * sx + inv[0, 1] * sy + inv[0, 2])) * 2
    expect_v = (sy - (inv[1, 0] * sx + inv[1, 1] * sy + inv[1, 2])) * 2
    np.testing.assert_allclose(up[..., 0][inner], expect_u[inner], atol=2e-4)
    np.testing.assert_allclose(up[..., 1][inner], expect_v[inner], atol=2e-4)
