"""Slab-wise scatter (SURVEY 8e, config 5 with ref 's'): the star passes of the Delaunay path sharded over ranks with ONE
exchange of the unfinished sites (include/ofl.h: ofl_scatter_slab_stars_dev, ofl_comm_allgather, ofl_scatter_slab_finish_dev).

The bar is the one the replicated band entry (ofl_scatter_rows_dev) already meets: the bands of all ranks, concatenated,
equal the whole-field result BIT FOR BIT -- values and validity -- for any number of ranks.  The ranks are rehearsed on the
one GPU of the test box (tests/slab_util.py); the exchange between real ranks is ofl_comm_allgather, exercised here on a
one-rank communicator.  What the whole-field result itself is worth against SciPy is the business of
tests/test_gpu_scatter*.py.
"""
import ctypes
import os
import time

import numpy as np
import pytest

from slab_util import Slab, probe_values

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rotation_field(h, w):
    import oflibnumpy_amd as of
    return of.Flow.from_transforms([['rotation', w / 2, h / 2, -20], ['scaling', w / 3.84, h / 2.7, 0.9]], [h, w], 's').vecs


def make_field(kind, h, w):
    """(vectors, point mask or None): fields the certificate refuses, one per reason the star passes exist"""
    rng = np.random.default_rng(11)
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    m = None
    if kind == "speckle":                       # 5 % of the sites dropped at random: fans and clips everywhere
        v, m = rotation_field(h, w), rng.random((h, w)) > 0.05
    elif kind == "hole":                        # a large hole: its rim goes through the wave pass, crosses several bands
        v, m = rotation_field(h, w), np.ones((h, w), bool)
        m[h // 5:h // 5 + h // 2, w // 4:w // 4 + w // 5] = False
    elif kind == "stripes":                     # motion boundaries top to bottom: tears in every band
        v = np.zeros((h, w, 2), np.float32)
        v[..., 0] = (np.floor(xx / 32) % 2) * 9.0
    elif kind == "object":                      # a block moving over a still background: folds and a tear
        v = np.zeros((h, w, 2), np.float32)
        v[h // 4:h // 4 * 3, w // 4:w // 4 * 3] = [13.0, -7.0]
    elif kind == "sintel":                      # BASELINE config 5's field: lattice sites, duplicates, ragged right border
        import oflibnumpy_amd as of
        flo = of.load_sintel(os.path.join(GOLDEN, "sintel.flo"))
        v = np.ascontiguousarray(np.tile(flo, (h // flo.shape[0], w // flo.shape[1], 1)))
    elif kind == "vertical":                    # sites leave their source rows by tens of pixels: slabs are not index ranges
        v = np.stack([3.0 * np.sin(yy / 17.0), 40.0 * np.sin(xx / 61.0) + 0.3 * yy], -1).astype(np.float32)
    elif kind == "outliers":                    # garbage vectors far outside the bucket grid, next to the first and last band
        v = (0.6 * rng.standard_normal((h, w, 2))).astype(np.float32)
        v[0, 5] = [0.0, -9e4]; v[h - 1, 9] = [3.0, 7e4]; v[h // 2, 0] = [-6e4, 0.0]
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(v, np.float32), m


def assert_bands_equal_full(sl, world, align=8):
    h, w, C = sl.h, sl.w, sl.C
    fo, fv, info = sl.full()
    fo, fv = fo.to_host((h, w, C), np.float32), fv.to_host((h, w), np.uint8)
    out, valid, lists, bands = sl.play(world, align)
    counts, errs = sl.counts(lists, world)
    assert not any(errs), errs
    assert sum(counts) == info[1], (counts, info)            # the ranks' lists partition the unfinished sites
    diff = (out.view(np.uint32) != fo.view(np.uint32)).any(-1) | (valid != fv)
    assert not diff.any(), (int(diff.sum()), np.argwhere(diff)[:5].tolist(), bands)
    return info, counts


@pytest.mark.parametrize("kind", ["speckle", "hole", "stripes", "object", "sintel", "vertical", "outliers"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_slab_bands_equal_whole_field(gpu, kind, world):
    h, w = (430, 760) if kind == "sintel" else (384, 512)
    vecs, m = make_field(kind, h, w)
    h, w = vecs.shape[:2]
    sl = Slab(vecs, m, probe_values(h, w))
    info, counts = assert_bands_equal_full(sl, world)
    assert info[1] > 0                                       # every field leaves sites to the exchange


def test_bands_thinner_than_the_slab_margin(gpu):
    """96 rows over 8 and 12 ranks: every slab overlaps several bands (and with 13 ranks of 8-row tiles one band is empty)"""
    vecs, m = make_field("speckle", 96, 160)
    sl = Slab(vecs, m, probe_values(96, 160))
    for world in (8, 12, 13):
        assert_bands_equal_full(sl, world)
    assert_bands_equal_full(sl, 5, align=1)                  # band edges off the 8-row tiles


def test_randomised_fields_worlds_and_band_edges(gpu):
    """seeded sweep: smooth fields with jumps, random point masks of every density, any number of ranks, band edges on and off
    the 8-row tiles -- bands == whole field, bit for bit"""
    rng = np.random.default_rng(2024)
    for case in range(16):
        h, w = int(rng.integers(24, 180)), int(rng.integers(40, 260))
        yy, xx = np.mgrid[:h, :w].astype(np.float32)
        amp = float(rng.choice([0.3, 2.0, 9.0]))
        v = np.stack([amp * np.sin(xx / rng.uniform(5, 40) + yy / rng.uniform(7, 60)),
                      amp * np.cos(yy / rng.uniform(5, 40) - xx / rng.uniform(9, 70))], -1).astype(np.float32)
        if rng.random() < 0.5:                                  # a block that moves on its own: folds and tears
            y0, x0 = int(rng.integers(0, h // 2)), int(rng.integers(0, w // 2))
            v[y0:y0 + h // 3, x0:x0 + w // 3] += rng.uniform(-12, 12, 2).astype(np.float32)
        if rng.random() < 0.3:
            v = np.round(v)                                     # lattice sites, duplicates
        keep = float(rng.choice([1.0, 0.97, 0.7, 0.3]))
        m = None if keep == 1.0 else rng.random((h, w)) < keep
        sl = Slab(v, m, probe_values(h, w), entries=1 << 15)
        world, align = int(rng.integers(2, 10)), int(rng.choice([1, 8]))
        try:
            assert_bands_equal_full(sl, world, align)
        except AssertionError as e:
            raise AssertionError(f"case {case}: {h} x {w}, amp {amp}, keep {keep}, world {world}, align {align}: {e}")


def test_float32_points_and_negated_field(gpu):
    """the two other ways the entries are called: positions rounded to float32 (mode 2), sign -1 (flow_class.py:1398-1400)"""
    vecs, m = make_field("speckle", 200, 320)
    for sign, pp in ((1, 1), (-1, 0)):
        sl = Slab(vecs, m, probe_values(200, 320), sign=sign, point_precision=pp)
        assert_bands_equal_full(sl, 3)


def test_value_masks_and_valid_rules(gpu):
    """the resolve step of a band is the whole-field one: a mask on the VALUES (Flow.apply's target mask, flow_class.py:643-668)
    under the three validity rules, with and without the rounding of integer targets"""
    h, w = 200, 320
    vecs, m = make_field("speckle", h, w)
    vm = np.random.default_rng(3).random((h, w)) > 0.2
    vals = np.round(40.0 * probe_values(h, w))
    for rule in (0, 1, 2, 2 | gpu.native.SCATTER_ROUND, gpu.native.SCATTER_NEGATE):
        sl = Slab(vecs, m, vals, vmask=vm, valid_rule=rule)
        assert_bands_equal_full(sl, 3)


def test_list_overflow_blanks_every_band(gpu):
    """a list buffer too small for a rank's unfinished sites: step 2 reports it ON EVERY RANK and returns an all-invalid band"""
    vecs, m = make_field("hole", 256, 384)
    sl = Slab(vecs, m, probe_values(256, 384), entries=64)
    bands = sl.bands(2)
    lists = sl.gather(bands)
    counts, _ = sl.counts(lists, 2)
    assert max(counts) > 64
    scratch = gpu.device.DeviceBuffer(sl.nb)
    for r0, r1 in bands:
        sl.stars(r0, r1 - r0, scratch.ptr)
        o, v = gpu.device.DeviceBuffer((r1 - r0) * 384 * 2 * 4), gpu.device.DeviceBuffer((r1 - r0) * 384)
        rc, _ = sl.finish(r0, r1 - r0, lists, 2, o, v, check=False)
        assert rc == gpu.native.E_INVALID
        assert "32" in gpu.native.last_error()
        assert not v.to_host((r1 - r0, 384), np.uint8).any()
        assert not o.to_host((r1 - r0, 384, 2), np.float32).any()


def test_a_list_that_is_not_of_this_field_is_refused(gpu):
    """the records carry site numbers the later passes take positions from: one that leaves the field (a list of another
    field, a damaged transfer) raises flag 32 and blanks the band instead of reading out of bounds"""
    h, w = 128, 192
    vecs, m = make_field("hole", h, w)
    sl = Slab(vecs, m, probe_values(h, w), entries=4096)
    lists = sl.gather([(0, h)])
    host = lists.to_host((sl.nb // 4,), np.uint32).copy()
    assert host[0] > 8
    for word, value in ((4 + 16 * 3 + 0, h * w + 5), (4 + 16 * 5 + 2, 0x7FFFFFFF)):      # a site index, a seed
        bad = host.copy()
        assert bad[4 + 16 * 5 + 1] >= 1                                                  # (record 5 has at least one seed)
        bad[word] = value
        dl = gpu.device.DeviceBuffer.from_host(bad)
        scratch = gpu.device.DeviceBuffer(sl.nb)
        sl.stars(0, h, scratch.ptr)
        o, v = gpu.device.DeviceBuffer(h * w * 2 * 4), gpu.device.DeviceBuffer(h * w)
        rc, _ = sl.finish(0, h, dl, 1, o, v, check=False)
        assert rc == gpu.native.E_INVALID and "32" in gpu.native.last_error()
        assert not v.to_host((h, w), np.uint8).any()


def test_step_2_refuses_a_workspace_that_is_not_its_step_1s(gpu):
    """step 2 trusts the lists and counts in the workspace, so it checks the stamp step 1 left: no step 1, a step 1 for another
    band, a whole-field call in between, or a second step 2 on the same state -> OFL_E_INVALID, outputs zeroed, no kernel runs"""
    h, w = 128, 192
    vecs, m = make_field("speckle", h, w)
    sl = Slab(vecs, m, probe_values(h, w))
    bands = sl.bands(2)
    lists = sl.gather(bands)
    (a0, a1), (b0, b1) = bands
    o, v = gpu.device.DeviceBuffer((a1 - a0) * w * 2 * 4), gpu.device.DeviceBuffer((a1 - a0) * w)
    scratch = gpu.device.DeviceBuffer(sl.nb)

    def refused():
        sl.nat.check(sl.lib.ofl_memset(o.ptr, 0x3F, o.nbytes, None))
        sl.nat.check(sl.lib.ofl_memset(v.ptr, 1, v.nbytes, None))
        rc, _ = sl.finish(a0, a1 - a0, lists, 2, o, v, check=False)
        assert rc == gpu.native.E_INVALID and "workspace" in gpu.native.last_error()
        assert not o.to_host((a1 - a0, w, 2), np.float32).any() and not v.to_host((a1 - a0, w), np.uint8).any()

    sl.stars(b0, b1 - b0, scratch.ptr)                      # the OTHER band's state
    refused()
    sl.stars(a0, a1 - a0, scratch.ptr)
    sl.full()                                               # a whole-field call on the same workspace in between
    refused()
    sl.stars(a0, a1 - a0, scratch.ptr)
    rc, _ = sl.finish(a0, a1 - a0, lists, 2, o, v)
    assert rc == 0 and v.to_host((a1 - a0, w), np.uint8).any()
    refused()                                               # a state is finished once


def test_argument_checks(gpu):
    vecs, m = make_field("speckle", 64, 96)
    sl = Slab(vecs, m, probe_values(64, 96))
    lib, nat = sl.lib, sl.nat
    lst = gpu.device.DeviceBuffer(sl.nb)
    bad = [
        lambda: lib.ofl_scatter_slab_stars_dev(None, 1, 0, None, 64, 96, 0, 64, lst.ptr, sl.nb, sl.ws.ptr, sl.ws.nbytes, None),
        lambda: lib.ofl_scatter_slab_stars_dev(sl.flow.ptr, 1, 0, None, 64, 96, 60, 8, lst.ptr, sl.nb, sl.ws.ptr, sl.ws.nbytes, None),
        lambda: lib.ofl_scatter_slab_stars_dev(sl.flow.ptr, 1, 0, None, 64, 96, 0, 64, lst.ptr, 24, sl.ws.ptr, sl.ws.nbytes, None),
        lambda: lib.ofl_scatter_slab_stars_dev(sl.flow.ptr, 1, 0, None, 64, 96, 0, 64, lst.ptr, sl.nb, sl.ws.ptr, 4096, None),
        lambda: lib.ofl_scatter_slab_stars_dev(sl.flow.ptr, 3, 0, None, 64, 96, 0, 64, lst.ptr, sl.nb, sl.ws.ptr, sl.ws.nbytes, None),
        lambda: lib.ofl_scatter_slab_finish_dev(sl.flow.ptr, 1, 0, sl.vals.ptr, 2, None, 64, 96, 0, 64, None, sl.nb, 1, lst.ptr, lst.ptr, 0,
                                                sl.ws.ptr, sl.ws.nbytes, None, None),
        lambda: lib.ofl_scatter_slab_finish_dev(sl.flow.ptr, 1, 0, sl.vals.ptr, 2, None, 64, 96, 0, 64, lst.ptr, sl.nb, 0, lst.ptr, lst.ptr, 0,
                                                sl.ws.ptr, sl.ws.nbytes, None, None),
        lambda: lib.ofl_scatter_slab_finish_dev(sl.flow.ptr, 1, 0, sl.vals.ptr, 2, None, 64, 96, 0, 64, lst.ptr, sl.nb + 4, 1, lst.ptr, lst.ptr, 0,
                                                sl.ws.ptr, sl.ws.nbytes, None, None),
    ]
    for call in bad:
        assert call() == nat.E_INVALID


def test_device_api_single_rank_and_allgather(gpu):
    """device.scatter_slab on a one-rank communicator: the RCCL all-gather entry runs (in place, one rank), the result is the
    whole-field one"""
    from oflibnumpy_amd import device as dev
    nat, lib = gpu.native, gpu.native.load()
    h, w = 200, 320
    vecs, m = make_field("speckle", h, w)
    vals = probe_values(h, w)
    sl = Slab(vecs, m, vals)
    fo, fv, _ = sl.full()
    fo, fv = fo.to_host((h, w, 2), np.float32), fv.to_host((h, w), np.uint8)
    out, valid = dev.DeviceBuffer(h * w * 2 * 4), dev.DeviceBuffer(h * w)
    info = dev.scatter_slab(sl.flow, 1, sl.pm, sl.vals, 2, None, h, w, 0, h, out, valid)
    assert np.array_equal(out.to_host((h, w, 2), np.float32).view(np.uint32), fo.view(np.uint32))
    assert np.array_equal(valid.to_host((h, w), np.uint8), fv)
    assert info[1] > 0
    with pytest.raises(ValueError):
        dev.scatter_slab(sl.flow, 1, sl.pm, sl.vals, 2, None, h, w, 8, 64, out, valid)       # a band without the other ranks
    uid = np.zeros(128, np.uint8)
    nat.check(lib.ofl_comm_unique_id(uid.ctypes.data))
    nat.check(lib.ofl_comm_init(uid.ctypes.data, 0, 1))
    try:
        calls = []

        def gather(send, recv, nbytes, stream):
            calls.append(nbytes)
            dev.comm_allgather(send, recv, nbytes, stream)

        # (world = 1 skips the exchange; play a "world" of one list through the gather hook by hand)
        nb = dev.slab_list_bytes(1 << 12)
        lists = dev.DeviceBuffer(nb)
        dev.scatter_slab_stars(sl.flow, 1, sl.pm, h, w, 0, h, lists.ptr, nb)
        before = lists.to_host((nb // 4,), np.uint32).copy()
        gather(lists.ptr, lists, nb, None)
        nat.check(lib.ofl_stream_sync(None))
        assert np.array_equal(lists.to_host((nb // 4,), np.uint32), before) and calls == [nb]
        dev.scatter_slab_finish(sl.flow, 1, sl.vals, 2, None, h, w, 0, h, lists, nb, 1, out, valid)
        assert np.array_equal(out.to_host((h, w, 2), np.float32).view(np.uint32), fo.view(np.uint32))
    finally:
        nat.check(lib.ofl_comm_destroy())


def test_config5_8k_slab_bands_and_time(gpu):
    """BASELINE config 5 at full size, 8 ranks: bands == whole field bit for bit, and one rank's share (step 1 + step 2 of band
    3) takes well under half of the whole-field time (measured 0.35; replicated stars: 0.91)."""
    flo = gpu.load_sintel(os.path.join(GOLDEN, "sintel.flo"))
    big = np.ascontiguousarray(np.tile(flo, (432, 384, 1)))
    h, w = big.shape[:2]
    assert (h, w) == (4320, 7680)
    sl = Slab(big, None, probe_values(h, w))
    info, counts = assert_bands_equal_full(sl, 8)
    assert info[0] == h * w and 50_000 < info[1] < 100_000
    bands = sl.bands(8)
    lists = sl.gather(bands)
    r0, r1 = bands[3]
    o, v = gpu.device.DeviceBuffer((r1 - r0) * w * 2 * 4), gpu.device.DeviceBuffer((r1 - r0) * w)
    scratch = gpu.device.DeviceBuffer(sl.nb)
    lib = sl.lib

    def ms(fn, iters=4):
        fn()
        sl.nat.check(lib.ofl_stream_sync(None))
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        sl.nat.check(lib.ofl_stream_sync(None))
        return (time.perf_counter() - t0) * 1e3 / iters

    t_full = ms(lambda: sl.full())
    t_band = ms(lambda: (sl.stars(r0, r1 - r0, scratch.ptr), sl.finish(r0, r1 - r0, lists, 8, o, v)))
    print(f"config 5 's' at 8K: whole field {t_full:.2f} ms, band 3/8 slab-wise {t_band:.2f} ms ({t_band / t_full:.2f})")
    assert t_band < 0.45 * t_full


def test_apply_image_rows_equals_apply_image(gpu):
    """DeviceFlow.apply_image_rows, the user-level entry of SURVEY 8e: 't' (gather band), 's' on a certified mesh (walk kernel
    on the band), 's' on a field the certificate refuses (slab-wise path; the other ranks are played by the `gather` hook on
    a second stream, i.e. a second workspace) -- bands of all ranks == apply_image, bit for bit, values and valid area"""
    from oflibnumpy_amd import device as dev
    import oflibnumpy_amd as of
    nat, lib = gpu.native, gpu.native.load()
    h, w = 200, 320
    rng = np.random.default_rng(5)
    img = dev.DeviceImage.from_host(rng.random((h, w, 3), dtype=np.float32))
    tmask = dev.DeviceBuffer.from_host((rng.random((h, w)) > 0.1).astype(np.uint8))
    affine = of.Flow.from_transforms([['rotation', 120, 80, -15], ['scaling', 100, 60, 0.9]], (h, w), 's')
    ragged, m = make_field("speckle", h, w)
    sp = ctypes.c_void_p()
    nat.check(lib.ofl_stream_create(ctypes.byref(sp)))
    other = sp.value
    try:
        cases = [("t", dev.DeviceFlow.from_host(affine.vecs, 't', m), None),
                 ("t + target mask", dev.DeviceFlow.from_host(ragged, 't'), tmask),
                 ("s certified", dev.DeviceFlow.from_host(affine.vecs, 's'), tmask),
                 ("s slab", dev.DeviceFlow.from_host(ragged, 's', m), None),
                 ("s zero flow", dev.DeviceFlow.from_host(np.zeros((h, w, 2), np.float32), 's', m), tmask)]
        for name, d, tm in cases:
            full, fvalid = d.apply_image(img, target_mask=tm)
            fo, fv = full.to_host(), fvalid.to_host((h, w), np.uint8)
            for world in (1, 3):
                got, gotv = np.zeros_like(fo), np.zeros_like(fv)
                for rank in range(world):
                    nb = dev.slab_list_bytes(1 << 17)
                    played = {}

                    def hook(send_ptr, recv, nbytes, stream, rank=rank, played=played):
                        from oflibnumpy_amd.sharding import row_band
                        if not played:                      # the other ranks' step 1, on their "own GPU" (stream = workspace)
                            for r in range(world):
                                if r == rank:
                                    continue
                                a, b = row_band(h, r, world)
                                played[r] = dev.DeviceBuffer(nb)
                                dev.scatter_slab_stars(d.vecs, +1, d._point_mask(True), h, w, a, b - a, played[r].ptr, nb, stream=other)
                            nat.check(lib.ofl_stream_sync(other))
                        for r in range(world):
                            src = send_ptr if r == rank else played[r].ptr
                            nat.check(lib.ofl_copy_dev(recv.ptr + r * nbytes, src, nbytes, stream))

                    o, v, (r0, r1) = d.apply_image_rows(img, rank, world, target_mask=tm, gather=hook)
                    got[r0:r1] = o.to_host()
                    gotv[r0:r1] = v.to_host((r1 - r0, w), np.uint8)
                    assert (name != "s slab" or world == 1) == (not played), name
                assert np.array_equal(got.view(np.uint32), fo.view(np.uint32)) and np.array_equal(gotv, fv), (name, world)
    finally:
        nat.check(lib.ofl_stream_destroy(other))


def _play_apply_image_rows(dev, nat, lib, d, img, tm, h, w, world, other, entries=1 << 17, peers_entries=1 << 17):
    """every rank of `world` through DeviceFlow.apply_image_rows; the peers' step 1 runs inside the gather hook on a second
    stream (a second workspace).  Returns the assembled image / valid area and the number of hook calls per rank."""
    from oflibnumpy_amd.sharding import row_band
    got, gotv, calls = None, None, []
    for rank in range(world):
        played, n_calls = {}, [0]

        def hook(send_ptr, recv, nbytes, stream, rank=rank, played=played, n_calls=n_calls):
            n_calls[0] += 1
            if not played:
                nb = dev.slab_list_bytes(peers_entries)
                for r in range(world):
                    if r == rank:
                        continue
                    a, b = row_band(h, r, world)
                    played[r] = dev.DeviceBuffer(nb)
                    if b > a:
                        dev.scatter_slab_stars(d.vecs, +1, d._point_mask(True), h, w, a, b - a, played[r].ptr, nb, stream=other)
                    else:                               # a peer with an empty band sends an empty list
                        nat.check(lib.ofl_memset(played[r].ptr, 0, nb, other))
                nat.check(lib.ofl_stream_sync(other))
            for r in range(world):
                src = send_ptr if r == rank else played[r].ptr
                nat.check(lib.ofl_copy_dev(recv.ptr + r * nbytes, src, nbytes, stream))

        if entries != 1 << 17:                          # (apply_image_rows has no `entries`: go one level down)
            r0, r1 = row_band(h, rank, world)
            o = dev.DeviceImage(dev.DeviceBuffer(max(r1 - r0, 0) * w * 3 * 4), (max(r1 - r0, 0), w, 3), np.float32)
            v = dev.DeviceBuffer(max(r1 - r0, 0) * w)
            dev.scatter_slab(d.vecs, +1, d._point_mask(True), img.buf, 3, d.mask, h, w, r0, r1 - r0, o.buf, v, rank, world,
                             entries=entries, gather=hook)
        else:
            o, v, (r0, r1) = d.apply_image_rows(img, rank, world, target_mask=tm, gather=hook)
        calls.append(n_calls[0])
        if r1 > r0:
            if got is None:
                got, gotv = np.zeros((h, w, 3), np.float32), np.zeros((h, w), np.uint8)
            got[r0:r1] = o.to_host()
            gotv[r0:r1] = v.to_host((r1 - r0, w), np.uint8)
        else:
            assert o is None or o.shape[0] == 0
    return got, gotv, calls


def test_more_ranks_than_row_tiles(gpu):
    """H = 40 over 8 ranks: five 8-row tiles, three ranks with EMPTY bands.  On the slab-wise path those ranks must still join
    both all-gathers (they used to return before the exchange and leave the others waiting in ncclAllGather for ever); on the
    other paths nothing is exchanged.  Bands of the non-empty ranks == apply_image."""
    from oflibnumpy_amd import device as dev
    import oflibnumpy_amd as of
    nat, lib = gpu.native, gpu.native.load()
    h, w, world = 40, 96, 8
    rng = np.random.default_rng(9)
    img = dev.DeviceImage.from_host(rng.random((h, w, 3), dtype=np.float32))
    ragged = (0.8 * rng.standard_normal((h, w, 2))).astype(np.float32)
    m = rng.random((h, w)) > 0.06
    affine = of.Flow.from_transforms([['rotation', 40, 20, -10]], (h, w), 's')
    sp = ctypes.c_void_p()
    nat.check(lib.ofl_stream_create(ctypes.byref(sp)))
    try:
        for name, d, want_calls in (("s slab", dev.DeviceFlow.from_host(ragged, 's', m), 2),
                                    ("s certified", dev.DeviceFlow.from_host(affine.vecs, 's'), 0),
                                    ("t", dev.DeviceFlow.from_host(ragged, 't', m), 0)):
            full, fvalid = d.apply_image(img)
            got, gotv, calls = _play_apply_image_rows(dev, nat, lib, d, img, None, h, w, world, sp.value)
            assert calls == [want_calls] * world, (name, calls)          # EVERY rank joined both gathers -- or none did
            assert np.array_equal(got.view(np.uint32), full.to_host().view(np.uint32)), name
            assert np.array_equal(gotv, fvalid.to_host((h, w), np.uint8)), name
    finally:
        nat.check(lib.ofl_stream_destroy(sp.value))


def test_scatter_slab_retries_once_when_a_list_overflows(gpu):
    """`entries` too small for this rank's unfinished sites: the gathered heads tell every rank so, and all of them repeat the
    exchange with lists sized for the fullest (heads, heads again, lists: 3 gathers instead of 2) -- same result as the whole-field call"""
    from oflibnumpy_amd import device as dev
    nat, lib = gpu.native, gpu.native.load()
    h, w, world = 256, 384, 2
    vecs, m = make_field("hole", h, w)
    d = dev.DeviceFlow.from_host(vecs, 's', m)
    d.stats()
    img = dev.DeviceImage.from_host(np.random.default_rng(2).random((h, w, 3), dtype=np.float32))
    full, fvalid = d.apply_image(img)
    sp = ctypes.c_void_p()
    nat.check(lib.ofl_stream_create(ctypes.byref(sp)))
    try:
        got, gotv, calls = _play_apply_image_rows(dev, nat, lib, d, img, None, h, w, world, sp.value, entries=64)
        assert calls == [3, 3], calls
        assert np.array_equal(got.view(np.uint32), full.to_host().view(np.uint32))
        assert np.array_equal(gotv, fvalid.to_host((h, w), np.uint8))
    finally:
        nat.check(lib.ofl_stream_destroy(sp.value))


def test_a_certificate_pass_voids_a_slab_state(gpu):
    """ofl_scatter_certify_dev writes its record over the front of the workspace header: a slab state left there by step 1 is
    void afterwards (the stamp is cleared) and step 2 refuses it instead of running on a trampled grid"""
    from oflibnumpy_amd import device as dev
    nat, lib = gpu.native, gpu.native.load()
    h, w = 128, 192
    vecs, m = make_field("speckle", h, w)
    sl = Slab(vecs, m, probe_values(h, w))
    lst = dev.DeviceBuffer(sl.nb)
    sl.stars(0, h, lst.ptr)
    cert = nat.MeshCert()
    nat.check(lib.ofl_scatter_certify_dev(sl.flow.ptr, 1, 0, sl.pm.ptr, h, w, sl.ws.ptr, sl.ws.nbytes, ctypes.byref(cert), None, None))
    o, v = dev.DeviceBuffer(h * w * 2 * 4), dev.DeviceBuffer(h * w)
    rc, _ = sl.finish(0, h, lst, 1, o, v, check=False)
    assert rc == nat.E_INVALID and "does not hold the state" in nat.last_error()
