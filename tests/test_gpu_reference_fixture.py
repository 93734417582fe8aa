"""The reference's own image fixture, tests/smudge.png (512 x 512 x 3 uint8 photograph; a byte-identical copy is
tests/golden/smudge.png), through the tests the reference runs on it -- /root/reference/tests/test_utils.py:267-283
(TestApplyFlow.test_rotation / test_translation) and tests/test_flow_class.py:418-467 (TestFlow.test_apply) -- with the HIP
path in place of cv2.remap / griddata, plus the comparison the reference cannot make: GPU vs the CPU oracle on the same image.

cv2.imread is not available (nor is it what is under test): the PNG is read by oflibnumpy_amd._png, which returns R, G, B where
cv2 returns B, G, R -- the channel order does not enter any of these tests; cv2.imread(path, 0) (grey) is replaced by the
luma sum OpenCV documents for it (0.299 R + 0.587 G + 0.114 B, rounded), which the loose rotation test does not depend on.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def smudge():
    from oflibnumpy_amd import _png
    img = _png.read_png(os.path.join(GOLDEN, "smudge.png"))
    assert img.shape == (512, 512, 3) and img.dtype == np.uint8
    return np.ascontiguousarray(img)


def test_rotation_and_translation_on_the_reference_image(gpu, smudge):
    """test_utils.py:267-283: rotation by -30 deg vs scipy.ndimage.rotate on the centre crop (atol 20, rtol 0.05, as there);
    integer translation (10, 20) == scipy.ndimage.shift, exactly -- both references"""
    from scipy import ndimage
    of = gpu
    grey = np.rint(smudge.astype(np.float64) @ np.array([0.299, 0.587, 0.114])).astype(np.uint8)
    for ref in ('t', 's'):
        flow = of.Flow.from_transforms([['rotation', 255.5, 255.5, -30]], grey.shape[:2], ref).vecs
        control = ndimage.rotate(grey, -30, reshape=False)
        warped = of.apply_flow(flow, grey, ref)
        np.testing.assert_allclose(control[200:300, 200:300], warped[200:300, 200:300], atol=20, rtol=0.05)
        flow = of.Flow.from_transforms([['translation', 10, 20]], smudge.shape[:2], ref).vecs
        np.testing.assert_array_equal(of.apply_flow(flow, smudge, ref), ndimage.shift(smudge, [20, 10, 0]))


def test_flow_apply_on_the_reference_image(gpu, smudge):
    """test_flow_class.py:418-467: Flow.apply == apply_flow for 3-D / 2-D / Flow targets with and without a mask, both
    references; a smaller flow on the padded image, cut and uncut, image and Flow targets (exact, as there)"""
    of = gpu
    img = smudge
    for ref in ('t', 's'):
        flow = of.Flow.from_transforms([['rotation', 30, 50, 30]], img.shape[:2], ref)
        mask = np.ones(img.shape[:2], 'bool')
        want = of.apply_flow(flow.vecs, img, ref)
        np.testing.assert_array_equal(flow.apply(img), want)
        np.testing.assert_array_equal(flow.apply(img, mask, return_valid_area=True)[0], want)
        want2 = of.apply_flow(flow.vecs, img[..., 0], ref)
        np.testing.assert_array_equal(flow.apply(img[..., 0]), want2)
        np.testing.assert_array_equal(flow.apply(img[..., 0], mask, return_valid_area=True)[0], want2)
        np.testing.assert_array_equal(flow.apply(flow).vecs, of.apply_flow(flow.vecs, flow.vecs, ref))
    ref = 't'
    flow = of.Flow.from_transforms([['rotation', 30, 50, 30]], img.shape[:2], ref)
    want = of.apply_flow(flow.vecs, img, ref)
    shape = [img.shape[0] - 90, img.shape[1] - 110]
    padding = [50, 40, 30, 80]
    inner = (slice(padding[0], -padding[1]), slice(padding[2], -padding[3]))
    cut_flow = of.Flow.from_transforms([['rotation', 0, 0, 30]], shape, ref)
    np.testing.assert_array_equal(cut_flow.apply(img, padding=padding, cut=False)[inner], want[inner])
    np.testing.assert_array_equal(cut_flow.apply(img, padding=padding, cut=True), want[inner])
    target_flow = of.Flow.from_transforms([['rotation', 30, 50, 30]], img.shape[:2], ref)
    want_f = of.apply_flow(flow.vecs, target_flow.vecs, ref)
    np.testing.assert_array_equal(cut_flow.apply(target_flow, padding=padding, cut=False).vecs[inner], want_f[inner])
    np.testing.assert_array_equal(cut_flow.apply(target_flow, padding=padding, cut=True).vecs, want_f[inner])


def test_reference_image_gpu_vs_oracle(gpu, oracle, smudge):
    """What the reference's tests do not pin: the warped photograph itself.  't': bit-exact against the oracle's restatement of
    cv2.remap for uint8 (fixed point) and float32, with the valid area; 's' (the oracle calls the same scipy griddata the
    reference calls; a 192 x 256 crop keeps Qhull to seconds): valid area bit-exact; float32 values within 1e-4 relative and
    the np.round-ed uint8 values equal at every node whose covering simplex of SciPy's triangulation is uniquely Delaunay
    (scatter_util.nonunique_nodes: a similarity transform has co-circular cells, where Qhull's diagonal is arbitrary and a
    photograph can tell the alternatives apart)."""
    from scatter_util import ambiguous_for
    of, O = gpu, oracle
    for tr in ([['rotation', 255.5, 255.5, -30]], [['rotation', 30, 50, 30], ['scaling', 200, 300, 1.1]]):
        f = of.Flow.from_transforms(tr, smudge.shape[:2], 't')
        o = O.OFlow(f.vecs, 't', f.mask)
        for img in (smudge, smudge.astype(np.float32), smudge[..., 1]):
            got, gv = f.apply(img, return_valid_area=True)
            want, wv = o.apply(img, return_valid_area=True)
            assert got.dtype == want.dtype
            np.testing.assert_array_equal(got, want)
            np.testing.assert_array_equal(gv, wv)
    crop = np.ascontiguousarray(smudge[160:352, 128:384])
    for tr in ([['rotation', 127.5, 95.5, -30]], [['rotation', 100, 60, 17], ['scaling', 100, 60, 0.9]]):
        f = of.Flow.from_transforms(tr, crop.shape[:2], 's')
        o = O.OFlow(f.vecs, 's', f.mask)
        got, gv = f.apply(crop, return_valid_area=True)
        want, wv = o.apply(crop, return_valid_area=True)
        np.testing.assert_array_equal(gv, wv)
        sure = ~ambiguous_for(f.vecs)
        assert sure.mean() > 0.5
        gotf, _ = f.apply(crop.astype(np.float32), return_valid_area=True)
        wantf, _ = o.apply(crop.astype(np.float32), return_valid_area=True)
        np.testing.assert_allclose(gotf[sure], wantf[sure], rtol=1e-4, atol=2e-3)
        # integer targets are np.round-ed from the float64 interpolant (utils.py:256-257): one within 1e-9 of .5 may fall either way
        d = np.abs(got.astype(int) - want.astype(int))[sure]
        assert d.max() <= 1 and (d != 0).mean() < 1e-4
