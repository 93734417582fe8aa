#!/usr/bin/env python
"""BASELINE config 5 on N GPUs: ONE 4320 x 7680 field (tests/golden/sintel.flo tiled) warps an RGB float32 image;
every rank computes one row band of the result.  The image (and, for ref 's', the whole flow) is
replicated -- rank 0's copy is broadcast over RCCL when the ranks own different GPUs -- and nothing is exchanged
after the launch.  Strong scaling: the field is fixed, the bands shrink with N.

    python tools/bench_bands.py                      # 1 GPU (the whole field)
    python tools/bench_bands.py --gpus N             # spawns its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/bench_bands.py --gpus N [--steps K] [--check]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tiles", type=int, nargs=2, default=[432, 384], help="repetitions of the 10 x 20 fixture (rows, cols)")
    ap.add_argument("--check", action="store_true", help="rank 0 also computes the whole field and compares its band")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:       # launcher mode: spawn the ranks before touching the GPU
        from oflibnumpy_amd import sharding
        raise SystemExit(sharding.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], args.gpus, timeout=900.0))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus {} but WORLD_SIZE {}".format(args.gpus, world))
    os.environ.setdefault("OFL_DEVICE", os.environ.get("LOCAL_RANK", "0"))
    import oflibnumpy_amd as of
    from oflibnumpy_amd import device as dev, sharding
    nat = of.native
    nat.ensure_device()
    lib = nat.load()
    dist = None
    if world > 1:
        import torch.distributed as dist
        sys.stdout.flush()
        saved_stdout = os.dup(1)            # gloo announces its connections on stdout
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    flo = of.load_sintel(os.path.join(ROOT, "tests", "golden", "sintel.flo"))
    big = np.ascontiguousarray(np.tile(flo, (args.tiles[0], args.tiles[1], 1)))
    h, w = big.shape[:2]
    img = np.random.default_rng(2).random((h, w, 3), dtype=np.float32) if rank == 0 else np.zeros((h, w, 3), np.float32)
    dimg = dev.DeviceImage.from_host(img)
    note = "not needed (1 GPU)"
    if world > 1 and nat.device_count() >= world:
        uid = np.zeros(128, np.uint8)
        if rank == 0:
            nat.check(lib.ofl_comm_unique_id(uid.ctypes.data))
        uid = np.ascontiguousarray(sharding.broadcast_bytes(dist, uid, 0))
        nat.check(lib.ofl_comm_init(uid.ctypes.data, rank, world))
        nat.check(lib.ofl_comm_broadcast(dimg.buf.ptr, h * w * 12, 0, None))        # the replicated source image
        nat.check(lib.ofl_device_sync())
        note = "image broadcast from rank 0 over RCCL"
    elif world > 1:
        dimg = dev.DeviceImage.from_host(np.random.default_rng(2).random((h, w, 3), dtype=np.float32))
        note = "rehearsal: ranks share a GPU, every rank generated the image itself"

    r0, r1 = sharding.row_band(h, rank, world)
    rows = r1 - r0
    flow_rows = dev.DeviceBuffer.from_host(np.ascontiguousarray(big[r0:r1]))       # 't': a rank needs its rows only
    flow_full = dev.DeviceBuffer.from_host(big)                                     # 's': any cell may land in any band
    out_s, valid_s = dev.DeviceBuffer(rows * w * 12), dev.DeviceBuffer(rows * w)

    def run(fn, steps, warm):
        for _ in range(warm):
            fn()
        nat.check(lib.ofl_device_sync())
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        nat.check(lib.ofl_device_sync())
        if dist is not None:
            dist.barrier()
        dt = (time.perf_counter() - t0) / steps
        return sharding.max_over_ranks(dist, [dt])[0] if dist is not None else dt

    t_t = run(lambda: dev.gather_rows(dimg, r0, rows, flow_rows, -1, want_valid=True), args.steps, args.warmup)
    t_s = run(lambda: dev.scatter_rows(flow_full, +1, None, dimg.buf, 3, None, h, w, r0, rows, out_s, valid_s),
              max(2, args.steps // 4), 1)
    # the same band with the star passes sharded: step 1, the exchange of the unfinished sites, step 2 (include/ofl.h)
    hook = dev.comm_allgather if (world > 1 and nat.device_count() >= world) else (sharding.host_allgather(dist) if world > 1 else None)
    slab = (lambda: dev.scatter_slab(flow_full, +1, None, dimg.buf, 3, None, h, w, r0, rows, out_s, valid_s, rank, world, gather=hook)) \
        if world > 1 else (lambda: dev.scatter_slab(flow_full, +1, None, dimg.buf, 3, None, h, w, 0, h, out_s, valid_s))
    t_slab = run(slab, max(2, args.steps // 4), 1)
    ok = None
    if args.check:
        slab()
        band_slab = out_s.to_host((rows, w, 3), np.float32).copy()
        bvalid_slab = valid_s.to_host((rows, w), np.uint8).copy()
        dev.scatter_rows(flow_full, +1, None, dimg.buf, 3, None, h, w, r0, rows, out_s, valid_s)
        same = bool(np.array_equal(band_slab.view(np.uint32), out_s.to_host((rows, w, 3), np.float32).view(np.uint32)) and
                    np.array_equal(bvalid_slab, valid_s.to_host((rows, w), np.uint8)))
        slab_ok = (sharding.max_over_ranks(dist, [0.0 if same else 1.0])[0] == 0.0) if dist is not None else same
    if args.check and rank == 0:
        full, fvalid = dev.gather_bilinear(dimg, flow_full, (h, w), -1, want_valid=True)
        band, bvalid = dev.gather_rows(dimg, r0, rows, flow_rows, -1, want_valid=True)
        ok = bool(np.array_equal(full.to_host()[r0:r1], band.to_host()) and
                  np.array_equal(fvalid.to_host((h, w), np.uint8)[r0:r1], bvalid.to_host((rows, w), np.uint8)))
        o2, v2 = dev.DeviceBuffer(h * w * 12), dev.DeviceBuffer(h * w)
        dev.scatter_linear(flow_full, +1, None, dimg.buf, 3, None, h, w, None, o2, v2, 0)
        ok = ok and bool(np.array_equal(o2.to_host((h, w, 3), np.float32)[r0:r1], out_s.to_host((rows, w, 3), np.float32)))
    if rank == 0:
        print(json.dumps({"workload": "config 5: {}x{} tiled Sintel field warps an RGB float32 image + valid area".format(h, w),
                          "n_gpus": world, "rows_per_rank": rows, "scaling": "strong",
                          "ref_t_ms_per_field": round(t_t * 1e3, 4), "ref_t_fields_per_s": round(1 / t_t, 1),
                          "ref_s_ms_per_field": round(t_s * 1e3, 4), "ref_s_fields_per_s": round(1 / t_s, 1),
                          "ref_s_slab_ms_per_field": round(t_slab * 1e3, 4), "ref_s_slab_fields_per_s": round(1 / t_slab, 1),
                          "slab_exchange": "none (1 rank)" if world == 1 else ("RCCL all-gather" if hook is dev.comm_allgather else "through the host (gloo): rehearsal"),
                          "slab_band_equals_replicated_band": slab_ok if args.check else None,
                          "exchange": note, "band_equals_full": ok}), flush=True)
    if dist is not None:
        nat.check(lib.ofl_comm_destroy())
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
