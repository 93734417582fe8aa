"""Strong-scaling entries of bench.py (`"strong": {...}` in its JSON line): the two BASELINE configs that are defined over a whole
node, measured for whatever world size bench.py was started with -- N = 1 gives the base of the curve, the driver's
`--gpus N` runs give the points.

  c4_256pairs     BASELINE config 4: 256 independent 1080 x 1920 pairs, combine_with mode 3; rank r owns the contiguous block
                  sharding.shard(256, r, N) and runs it in launches of up to 32 pairs (ofl_compose3_dev).  The timed region also
                  carries the one exchange the config has: an RCCL broadcast of a shared 1080p field (vectors + mask, 18.7 MB)
                  from rank 0 on the launch stream (N > 1 on distinct GPUs only).  value = 256 / slowest rank's time.
  c5_t_bands      BASELINE config 5 wrapped as 't': the 4320 x 7680 tiled Sintel field warps an RGB float32 image, rank r computes
                  row band sharding.row_band(H, r, N) (ofl_gather_rows_dev; image replicated before the timed region, nothing is
                  exchanged inside it).
  c5_s_slab       the same field as loaded ('s'): slab-wise scatter (ofl_scatter_slab_stars_dev, TWO all-gathers over
                  ofl_comm_allgather inside the timed region, ofl_scatter_slab_finish_dev); N = 1: the whole-field call.

Ranks that share one GPU (a rehearsal on a 1-GPU box) exchange through the host (gloo) instead of RCCL, and say so.
Nothing here is imported by the product.
"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np


def run(of, dist, rank, world, comm_ok, log=lambda s: None, steps=6, which=("c4", "c5t", "c5s")):
    """comm_ok: a live RCCL communicator spans the ranks (bench.py created it).  Returns the dict (identical on every rank
    up to timing noise; rank 0's is printed)."""
    from oflibnumpy_amd import device as dev, sharding
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from bench_ops import affine_flow
    nat = of.native
    lib = nat.load()
    shared_gpu = world > 1 and nat.device_count() < world

    def barrier():
        nat.check(lib.ofl_device_sync())
        if dist is not None:
            dist.barrier()

    def slowest(seconds):
        return sharding.max_over_ranks(dist, [seconds])[0] if dist is not None else seconds

    def timed(fn, iters, warm=1):
        for _ in range(warm):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        barrier()
        return slowest((time.perf_counter() - t0) / iters)

    out = {"n_gpus": world, "scaling": "strong", "exchange": "none (1 GPU)" if world == 1 else ("through the host (ranks share a GPU): rehearsal" if shared_gpu or not comm_ok
                                                                           else "RCCL over xGMI")}
    # ------------------------------------------------------------------ config 4: 256 pairs, strong
    if "c4" in which:
        h, w, total, per_launch = 1080, 1920, 256, 32
        n = h * w
        mine = sharding.shard(total, rank, world)
        B = len(mine)
        va, ma, vb, mb = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n)
        vo, mo, st = dev.DeviceBuffer(B * n * 8), dev.DeviceBuffer(B * n), dev.DeviceBuffer.zeros(32 * max(B, 1))
        gen = min(B, 8)                                   # 8 generated pairs per rank, device-copied into the rank's slots:
        for i in range(gen):                              # the measurement is about where the bytes live, not their values
            j = mine[i]
            fa = affine_flow([['rotation', w / 2, h / 2, -30 + 60 * j / 255]], h, w, 't')
            fb = affine_flow([['translation', 40 * math.cos(j), 40 * math.sin(j)]], h, w, 't')
            m = (np.random.default_rng(j).random((h, w)) > 0.05).astype(np.uint8)
            for dst, a in ((va.ptr + i * n * 8, fa), (vb.ptr + i * n * 8, fb), (ma.ptr + i * n, m), (mb.ptr + i * n, m)):
                a = np.ascontiguousarray(a)
                nat.check(lib.ofl_upload(dst, a.ctypes.data, a.nbytes, None))
                nat.check(lib.ofl_stream_sync(None))
        for i in range(gen, B):
            for buf, sz in ((va, n * 8), (vb, n * 8), (ma, n), (mb, n)):
                nat.check(lib.ofl_copy_dev(buf.ptr + i * sz, buf.ptr + (i % gen) * sz, sz, None))
        shared_v, shared_m = dev.DeviceBuffer(n * 8), dev.DeviceBuffer(n)
        nat.check(lib.ofl_copy_dev(shared_v.ptr, va.ptr, n * 8, None))
        nat.check(lib.ofl_copy_dev(shared_m.ptr, ma.ptr, n, None))
        use_rccl = world > 1 and comm_ok and not shared_gpu

        def job():
            if use_rccl:                                  # the config's one exchange: a shared source field, rank 0 -> all
                nat.check(lib.ofl_comm_broadcast(shared_v.ptr, n * 8, 0, None))
                nat.check(lib.ofl_comm_broadcast(shared_m.ptr, n, 0, None))
            for i0 in range(0, B, per_launch):
                k = min(per_launch, B - i0)
                nat.check(lib.ofl_compose3_dev(va.ptr + i0 * n * 8, ma.ptr + i0 * n, vb.ptr + i0 * n * 8, mb.ptr + i0 * n, -1, h, w, k,
                                               vo.ptr + i0 * n * 8, mo.ptr + i0 * n, st.ptr + 32 * i0, 0, None))
        t = timed(job, max(3, steps), warm=2)
        out["c4_256pairs"] = {"ms_per_256_pairs": round(t * 1e3, 4), "pairs_per_s": round(total / t, 1), "pairs_per_rank": B,
                              "broadcast_in_timed_region": bool(use_rccl), "algorithmic_bytes": 27 * n * total,
                              "frac_of_node_roofline": round(27 * n * total / t / (8e12 * world), 4), "kernel": "compose3_xpose_kernel"}
        log("strong c4_256pairs: {:.3f} ms".format(t * 1e3))
        del va, ma, vb, mb, vo, mo, st, shared_v, shared_m
        dev.empty_cache()
    # ------------------------------------------------------------------ config 5: one 8K field, strong
    if "c5t" in which or "c5s" in which:
        flo = of.load_sintel(os.path.join(ROOT, "tests", "golden", "sintel.flo"))
        big = np.ascontiguousarray(np.tile(flo, (432, 384, 1)))
        h, w = big.shape[:2]
        img = np.random.default_rng(2).random((h, w, 3), dtype=np.float32)       # (the same seed on every rank: replicated)
        dimg = dev.DeviceImage.from_host(img)
        del img
        if world > 1 and comm_ok and not shared_gpu:
            nat.check(lib.ofl_comm_broadcast(dimg.buf.ptr, h * w * 12, 0, None))   # the replicated source image, untimed set-up
            nat.check(lib.ofl_device_sync())
        r0, r1 = sharding.row_band(h, rank, world)
        rows = r1 - r0
        if "c5t" in which:
            flow_rows = dev.DeviceBuffer.from_host(np.ascontiguousarray(big[r0:r1]))
            t = timed(lambda: dev.gather_rows(dimg, r0, rows, flow_rows, -1, want_valid=True), max(6, steps), warm=2)
            out["c5_t_bands"] = {"ms_per_field": round(t * 1e3, 4), "fields_per_s": round(1 / t, 2), "rows_per_rank": rows,
                                 "algorithmic_bytes": 34 * h * w, "frac_of_node_roofline": round(34 * h * w / t / (8e12 * world), 4),
                                 "kernel": "gather2_kernel"}
            log("strong c5_t_bands: {:.3f} ms".format(t * 1e3))
            del flow_rows
        if "c5s" in which:
            flow_full = dev.DeviceBuffer.from_host(big)
            out_s, valid_s = dev.DeviceBuffer(max(rows, 1) * w * 12), dev.DeviceBuffer(max(rows, 1) * w)
            if world == 1:
                fn = lambda: dev.scatter_linear(flow_full, +1, None, dimg.buf, 3, None, h, w, None, out_s, valid_s, nat.SCATTER_UNCERTIFIED)
            else:
                hook = dev.comm_allgather if (comm_ok and not shared_gpu) else sharding.host_allgather(dist)
                fn = lambda: dev.scatter_slab(flow_full, +1, None, dimg.buf, 3, None, h, w, r0, rows, out_s, valid_s, rank, world, gather=hook)
            t = timed(fn, max(3, steps // 2), warm=1)
            out["c5_s_slab"] = {"ms_per_field": round(t * 1e3, 4), "fields_per_s": round(1 / t, 2), "rows_per_rank": rows,
                                "algorithmic_bytes": 34 * h * w, "frac_of_node_roofline": round(34 * h * w / t / (8e12 * world), 4),
                                "kernel": "dl_* slab-wise + 2 all-gathers" if world > 1 else "dl_* whole field"}
            log("strong c5_s_slab: {:.3f} ms".format(t * 1e3))
        dev.empty_cache()
    return out
