"""One-GPU rehearsal of the slab-wise scatter (include/ofl.h: ofl_scatter_slab_stars_dev / ofl_scatter_slab_finish_dev).

Every rank of a `world` is played in turn on the raw C ABI with ONE workspace: step 1 of all ranks first -- the all-gather is
then a concatenation of their lists in one device buffer -- and step 1 + step 2 rank by rank (step 2 needs the star state of
ITS rank's step 1 in the workspace).  tests/test_gpu_slab.py and tools/slab_check.py use it.
"""
import ctypes

import numpy as np


class Slab:
    def __init__(self, vecs, pmask, vals, entries=1 << 17, sign=1, point_precision=0, vmask=None, valid_rule=0):
        from oflibnumpy_amd import _native as nat
        from oflibnumpy_amd import device as dev
        self.nat, self.dev, self.lib = nat, dev, nat.load()
        self.h, self.w = vecs.shape[:2]
        self.C = vals.shape[2]
        self.sign, self.pp = sign, point_precision
        self.flow = dev.DeviceBuffer.from_host(np.ascontiguousarray(vecs, np.float32))
        self.pm = dev.DeviceBuffer.from_host(np.ascontiguousarray(pmask).astype(np.uint8)) if pmask is not None else None
        self.vals = dev.DeviceBuffer.from_host(np.ascontiguousarray(vals, np.float32))
        self.vm = dev.DeviceBuffer.from_host(np.ascontiguousarray(vmask).astype(np.uint8)) if vmask is not None else None
        self.rule = valid_rule
        n = ctypes.c_size_t(0)
        nat.check(self.lib.ofl_scatter_workspace_bytes(self.h, self.w, self.C, ctypes.byref(n)))
        self.ws = dev.DeviceBuffer(n.value)
        self.nb = dev.slab_list_bytes(entries)

    def _p(self, b):
        return b.ptr if b is not None else None

    def full(self, row0=0, rows=None):
        """the whole-field call (or the replicated-stars band call) on the Delaunay path"""
        h, w, C = self.h, self.w, self.C
        rows = h if rows is None else rows
        out, valid = self.dev.DeviceBuffer(rows * w * C * 4), self.dev.DeviceBuffer(rows * w)
        info = (ctypes.c_uint64 * 3)()
        self.nat.check(self.lib.ofl_scatter_rows_dev(self.flow.ptr, self.sign, self.pp, self._p(self.pm), self.vals.ptr, C, self._p(self.vm), h, w,
                                                     row0, rows, out.ptr, valid.ptr, self.rule | self.nat.SCATTER_UNCERTIFIED,
                                                     self.ws.ptr, self.ws.nbytes, info, None))
        return out, valid, tuple(info)

    def stars(self, row0, rows, list_ptr):
        self.nat.check(self.lib.ofl_scatter_slab_stars_dev(self.flow.ptr, self.sign, self.pp, self._p(self.pm), self.h, self.w, row0, rows,
                                                           list_ptr, self.nb, self.ws.ptr, self.ws.nbytes, None))

    def finish(self, row0, rows, lists, n_lists, out, valid, check=True):
        info = (ctypes.c_uint64 * 3)()
        rc = self.lib.ofl_scatter_slab_finish_dev(self.flow.ptr, self.sign, self.pp, self.vals.ptr, self.C, self._p(self.vm), self.h, self.w, row0, rows,
                                                  lists.ptr, self.nb, n_lists, out.ptr, valid.ptr, self.rule,
                                                  self.ws.ptr, self.ws.nbytes, info, None)
        if check:
            self.nat.check(rc)
        return rc, tuple(info)

    def bands(self, world, align=8):
        from oflibnumpy_amd.sharding import row_band
        return [row_band(self.h, r, world, align) for r in range(world)]

    def gather(self, bands):
        """step 1 of every rank; the lists side by side as an all-gather leaves them"""
        lists = self.dev.DeviceBuffer(self.nb * len(bands))
        for r, (r0, r1) in enumerate(bands):
            if r1 > r0:
                self.stars(r0, r1 - r0, lists.ptr + r * self.nb)
            else:                                                   # more ranks than row tiles: an empty band reports nothing
                self.nat.check(self.lib.ofl_memset(lists.ptr + r * self.nb, 0, self.nb, None))
        return lists

    def counts(self, lists, world):
        head = lists.to_host((world, self.nb // 4), np.uint32)[:, :2]
        return head[:, 0].tolist(), head[:, 1].tolist()

    def play(self, world, align=8):
        """all ranks in turn -> (out [H][W][C], valid [H][W]) assembled from the bands, the gathered lists, the bands"""
        h, w, C = self.h, self.w, self.C
        bands = self.bands(world, align)
        lists = self.gather(bands)
        out = np.zeros((h, w, C), np.float32)
        valid = np.zeros((h, w), np.uint8)
        scratch = self.dev.DeviceBuffer(self.nb)
        for r0, r1 in bands:
            if r1 <= r0:
                continue
            self.stars(r0, r1 - r0, scratch.ptr)                 # this rank's star state back into the one workspace
            o, v = self.dev.DeviceBuffer((r1 - r0) * w * C * 4), self.dev.DeviceBuffer((r1 - r0) * w)
            self.finish(r0, r1 - r0, lists, world, o, v)
            out[r0:r1] = o.to_host((r1 - r0, w, C), np.float32)
            valid[r0:r1] = v.to_host((r1 - r0, w), np.uint8)
        return out, valid, lists, bands


def probe_values(h, w):
    yy, xx = np.mgrid[:h, :w].astype(np.float32)
    return np.stack([np.sin(xx / 37.0) + yy / 500.0, np.cos(yy / 23.0) * xx / 700.0], -1).astype(np.float32)
