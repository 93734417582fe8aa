"""CPU suite, part 3: the N > 1 plumbing with world_size 2 over gloo (no GPU): shards partition the work
disjointly and completely, the unique-id broadcast delivers rank 0's bytes, timings reduce to the maximum."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from oflibnumpy_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_partitions():
    for n in (0, 1, 7, 8, 256, 1000):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in sharding.shard(n, r, world)]
            assert got == list(range(n))
            sizes = [len(sharding.shard(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert list(sharding.shard(256, 3, 8)) == list(range(96, 128))      # BASELINE config 4: 32 pairs per GPU
    with pytest.raises(ValueError):
        sharding.shard(4, 2, 2)


def test_row_bands_cover_field():
    for h in (7, 8, 2160, 4320, 1001):
        for world in (1, 2, 8):
            bands = [sharding.row_band(h, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            for (a0, a1), (b0, b1) in zip(bands, bands[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(b[0] % 8 == 0 or b[0] == h for b in bands)


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch.distributed as dist
    from oflibnumpy_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    uid = np.arange(128, dtype=np.uint8) * (3 if rank == 0 else 0) + (7 if rank == 0 else 0)
    got = sharding.broadcast_bytes(dist, uid, 0)
    mine = list(sharding.shard(10, rank, world))
    t = sharding.max_over_ranks(dist, [1.0 + rank, 5.0 - rank])
    # the exchange of a slab-wise scatter as device.scatter_slab runs it, on host buffers: heads first, then as many
    # 64-byte records per rank as the fullest list holds
    cap, count = 100, [3, 41][rank]
    lst = np.zeros(4 + 16 * cap, np.uint32)
    lst[0] = count
    lst[4:4 + 16 * count] = np.arange(16 * count, dtype=np.uint32) + 1000 * rank
    heads = sharding.allgather_bytes(dist, lst[:4].view(np.uint8)).view(np.uint32)
    m = sharding.slab_payload_entries(heads[:, 0], cap)
    lists = sharding.allgather_bytes(dist, lst[:4 + 16 * m].view(np.uint8)).view(np.uint32)
    dist.barrier()
    print(json.dumps({{"rank": rank, "uid_sum": int(got.astype(int).sum()), "items": mine, "t": t, "m": m,
                      "counts": lists[:, 0].tolist(), "shape": list(lists.shape),
                      "rec": [int(lists[0, 4 + 16 * 2]), int(lists[1, 4 + 16 * 40 + 15])]}}), flush=True)
    dist.destroy_process_group()
""")


def test_two_ranks_over_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
        assert p.returncode == 0, err[-2000:]
        outs.append(__import__("json").loads(out.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    want_sum = int((np.arange(128, dtype=np.uint8) * 3 + 7).astype(int).sum())
    assert [d["uid_sum"] for d in outs] == [want_sum, want_sum]
    assert outs[0]["items"] + outs[1]["items"] == list(range(10))
    assert outs[0]["t"] == outs[1]["t"] == [2.0, 5.0]
    for d in outs:                                  # both ranks hold both lists, trimmed to the fuller one
        assert d["m"] == 41 and d["counts"] == [3, 41] and d["shape"] == [2, 4 + 16 * 41]
        assert d["rec"] == [32, 1000 + 16 * 40 + 15]
    assert sharding.slab_payload_entries([0, 0], 64) == 1 and sharding.slab_payload_entries([5, 900], 64) == 64


SPAWNED = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    if "WORLD_SIZE" not in os.environ:
        from oflibnumpy_amd import sharding
        raise SystemExit(sharding.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], int(sys.argv[1]), timeout=120))
    import torch.distributed as dist
    from oflibnumpy_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    fd = os.dup(1); os.dup2(2, 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    os.dup2(fd, 1)
    t = sharding.max_over_ranks(dist, [float(rank)])
    print(json.dumps({{"rank": rank, "world": world, "max_rank": t[0], "argv": sys.argv[1:]}}), flush=True)
    dist.destroy_process_group()
    sys.exit(int(sys.argv[2]) if rank == 1 else 0)
""")


def test_self_spawn_relays_rank0_and_worst_exit_code(tmp_path):
    """`python bench.py --gpus N` without a launcher: the parent spawns N fresh rank processes with the torchrun
    environment, prints rank 0's stdout only, and exits with the worst child code."""
    script = tmp_path / "spawned.py"
    script.write_text(SPAWNED.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for want_rc in (0, 5):
        p = subprocess.run([sys.executable, str(script), "2", str(want_rc)], env=env, capture_output=True, text=True, timeout=180)
        assert p.returncode == want_rc, p.stderr[-2000:]
        lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        d = __import__("json").loads(lines[0])
        assert d == {"rank": 0, "world": 2, "max_rank": 1.0, "argv": ["2", str(want_rc)]}


STUCK = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, {root!r})
    if "WORLD_SIZE" not in os.environ:
        from oflibnumpy_amd import sharding
        raise SystemExit(sharding.spawn_ranks(os.path.abspath(__file__), sys.argv[1:], 2, timeout=float(sys.argv[2]), grace=1.0))
    rank = int(os.environ["RANK"])
    if sys.argv[1] == "dies" and rank == 1:
        sys.exit(7)                     # a rank that fails at start-up (bad device, import error)
    print("rank0 waits", flush=True)
    time.sleep(600)                     # its peer sits in the rendezvous / a barrier
""")


def test_self_spawn_does_not_hang_on_a_dead_or_stuck_rank(tmp_path):
    """A rank that dies at start-up takes the launch down with ITS exit code within the grace period, and a launch whose
    ranks never finish ends with 124 at the timeout -- the driver's `python bench.py --gpus N` cannot hang forever."""
    import time
    script = tmp_path / "stuck.py"
    script.write_text(STUCK.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    for mode, limit, want in (("dies", "300", 7), ("stuck", "3", 124)):
        t0 = time.monotonic()
        p = subprocess.run([sys.executable, str(script), mode, limit], env=env, capture_output=True, text=True, timeout=120)
        assert p.returncode == want, (mode, p.returncode, p.stderr[-1000:])
        assert time.monotonic() - t0 < 60, mode
        assert "rank0 waits" in p.stdout           # what rank 0 printed before it was stopped is still relayed


ALONE = textwrap.dedent("""
    import os, sys, time
    sys.path.insert(0, {root!r})
    if "WORLD_SIZE" not in os.environ:
        from oflibnumpy_amd import sharding
        raise SystemExit(sharding.spawn_ranks(os.path.abspath(__file__), [], 2, timeout=120, grace=2))
    import numpy as np
    import torch.distributed as dist
    from oflibnumpy_amd import sharding
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    fd = os.dup(1); os.dup2(2, 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dist.barrier()
    os.dup2(fd, 1)
    if rank == 1:
        time.sleep(60)              # took another code path: never joins the gather of the slab list heads
        sys.exit(0)
    sharding.bounded_call(lambda: sharding.allgather_bytes(dist, np.zeros(16, np.uint8)), 3.0, "the all-gather of the slab list heads")
    print("not reached", flush=True)
""")


def test_a_rank_alone_in_the_slab_exchange_leaves_with_an_error(tmp_path):
    """device.scatter_slab bounds its two all-gathers with sharding.bounded_call: a rank whose peer never joins does not hang --
    the process ends with exit code 3 (and takes the job down through the launcher), nothing is re-exec'ed"""
    import time
    assert sharding.bounded_call(lambda: 41 + 1, 5.0, "x") == 42 and sharding.bounded_call(lambda: 7, None, "x") == 7
    with pytest.raises(KeyError):
        sharding.bounded_call(lambda: {}["missing"], 5.0, "x")
    script = tmp_path / "stuck.py"
    script.write_text(ALONE.format(root=ROOT))
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=100,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode == 3, (p.returncode, p.stderr[-1500:])
    assert time.monotonic() - t0 < 45
    assert "did not complete within 3 s" in p.stderr and "not reached" not in p.stdout


def test_bench_launcher_mode_is_reached_before_the_engine_loads():
    """bench.py / tools/bench_bands.py with --gpus N > 1 and no WORLD_SIZE must hand over to spawn_ranks before
    loading the native library / selecting a device (the parent never touches the GPU)."""
    for rel in ("bench.py", os.path.join("tools", "bench_bands.py")):
        src = open(os.path.join(ROOT, rel)).read()
        i_spawn, i_engine = src.index("spawn_ranks("), src.index("ensure_device()")
        assert 0 < i_spawn < i_engine, rel          # the package import itself is pure Python: no dlopen, no HIP call
