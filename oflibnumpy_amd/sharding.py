"""Rank-level plumbing of the sharded workload (one process per GPU, launched by torch.distributed.run).

The path shards by independent units -- pairs of flow fields (BASELINE config 4) or output row bands of
one huge field (config 5) -- so there is no data-path collective.  The only exchange is the one-off RCCL
broadcast of a shared source field; its 128-byte unique id travels over the launcher's process group.
Nothing here touches the GPU: the functions are exercised with gloo on CPU (tests/test_multirank.py).
"""
import numpy as np


def shard(n_items, rank, world):
    """Contiguous block of item indices owned by `rank` (blocks differ by at most one item)."""
    if world < 1 or not 0 <= rank < world or n_items < 0:
        raise ValueError("bad shard request: n_items={}, rank={}, world={}".format(n_items, rank, world))
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def row_band(height, rank, world, align=8):
    """Output rows [start, stop) of rank's band when one field is split over `world` GPUs; band edges are
    multiples of `align` rows (the compose kernel's tile height) except the last."""
    tiles = -(-height // align)
    r = shard(tiles, rank, world)
    return min(r.start * align, height), min(r.stop * align, height)


def local_device(local_rank, n_devices):
    if n_devices <= 0:
        raise RuntimeError("no HIP device visible")
    return local_rank % n_devices


def broadcast_bytes(dist, payload, src=0):
    """Broadcast a fixed-size uint8 array (e.g. the RCCL unique id) from `src` over the process group."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(payload, np.uint8).copy())
    dist.broadcast(t, src)
    return t.numpy()


def allgather_bytes(dist, payload):
    """uint8 arrays of equal length from every rank, in rank order (gloo)."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(payload, np.uint8).copy())
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return np.stack([p.numpy() for p in parts])


def slab_payload_entries(counts, entries):
    """Records per rank the second all-gather of a slab-wise scatter carries: as many as the fullest list holds (every rank
    computes this from the same gathered heads), never more than the lists' capacity -- a rank whose list overflowed keeps
    a count above it, which step 2 reports on every rank -- and at least one (the smallest list the C ABI accepts)."""
    return max(1, min(int(max(counts)), int(entries)))


def host_allgather(dist):
    """An all-gather of device memory THROUGH THE HOST (download, gloo, upload) with the signature device.scatter_slab
    expects of its `gather` hook: for rehearsals in which the ranks share one GPU -- RCCL refuses two ranks on a device.
    Real multi-GPU runs use the default hook, ofl_comm_allgather."""
    def gather(send_ptr, recv, nbytes, stream=None):
        import numpy as np
        from . import device as dev

        class _At:                                  # (to_host needs nothing but the address)
            ptr = send_ptr
        from . import _native as nat
        mine = dev.DeviceBuffer.to_host(_At, (nbytes,), np.uint8, stream)
        every = np.ascontiguousarray(allgather_bytes(dist, mine))
        nat.check(nat.load().ofl_upload(recv.ptr, every.ctypes.data, every.size, stream))
        nat.check(nat.load().ofl_stream_sync(stream))                 # (`every` is a temporary)
    return gather


def bounded_call(fn, timeout, what, exit_code=3):
    """Run fn() -- a collective, or the synchronisation behind one -- with a bound on the wait.  A rank whose peers never
    arrive (one of them took another code path, or died) would otherwise sit in the collective for ever; after `timeout`
    seconds this rank says so on stderr and the PROCESS exits with `exit_code` (os._exit from the waiting thread's parent:
    neither the communicator nor the process group can be torn down while a rank is inside them; nothing is re-exec'ed).
    Its peers -- stuck in the same collective -- run into the same bound, so every rank of the job ends non-zero.
    timeout None or <= 0: no bound.  Returns fn's result; an exception raised by fn is re-raised here."""
    if timeout is None or timeout <= 0:
        return fn()
    import os
    import sys
    import threading
    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:      # noqa: BLE001 - handed to the caller
            box["error"] = e

    worker = threading.Thread(target=run, daemon=True)
    worker.start()
    worker.join(timeout)
    if worker.is_alive():
        sys.stderr.write("rank {}: {} did not complete within {:.0f} s -- a peer never joined it; leaving with exit code {}\n".format(
            os.environ.get("RANK", "0"), what, timeout, exit_code))
        sys.stderr.flush()
        sys.stdout.flush()
        os._exit(exit_code)
    if "error" in box:
        raise box["error"]
    return box.get("value")


def max_over_ranks(dist, values):
    """Element-wise maximum of a list of floats over all ranks (timings are reported as the slowest rank's)."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t]


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(script, argv, world, timeout=None, env=None, grace=10.0):
    """Self-launch: run `script argv` once per rank as FRESH child processes (subprocess, never os.exec*) with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set the way torch.distributed.run sets them, relay rank 0's
    stdout to ours and return the worst child exit code.  The caller must not have touched the GPU: the parent stays a
    plain launcher.  Ranks > 0 write their stdout to our stderr (the contract is ONE JSON line on stdout).

    Every child is polled: the first rank that exits with an error takes the others down after `grace` seconds (a rank
    that died at start-up would otherwise leave its peers in the rendezvous until torch's own timeout, tens of minutes),
    and `timeout` seconds bound the whole launch (exit code 124, as timeout(1)).  timeout=None: the environment variable
    OFL_SPAWN_TIMEOUT (seconds) if set, else no bound -- soaks and cold builds in the children may legitimately run long;
    bench.py and tools/bench_bands.py pass their own finite bound."""
    import os
    import subprocess
    import sys
    import threading
    import time
    if world < 1:
        raise ValueError("world must be >= 1")
    if timeout is None and os.environ.get("OFL_SPAWN_TIMEOUT"):
        timeout = float(os.environ["OFL_SPAWN_TIMEOUT"])
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    base.setdefault("MASTER_PORT", str(free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this pool
    procs = []
    for rank in range(world):
        e = dict(base, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world))
        procs.append(subprocess.Popen([sys.executable, script] + list(argv), env=e,
                                      stdout=subprocess.PIPE if rank == 0 else sys.stderr))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # drains rank 0's pipe
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    first_bad, bad_since, timed_out = 0, None, False
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        for p in procs:
            rc = p.poll()
            if rc not in (None, 0) and first_bad == 0:
                first_bad, bad_since = (rc if rc > 0 else 128 - rc), now
        if deadline is not None and now > deadline:
            timed_out = True
        if timed_out or (bad_since is not None and now - bad_since > grace):
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    reader.join(5.0)
    sys.stdout.write(b"".join(c for c in chunks if c).decode("utf-8", "replace"))
    sys.stdout.flush()
    if timed_out:
        return 124
    worst = first_bad
    for p in procs:
        rc = p.returncode
        if rc != 0 and worst == 0:
            worst = rc if rc > 0 else 128 - rc          # a signal: shell convention
    return worst
