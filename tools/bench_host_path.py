#!/usr/bin/env python
"""PCIe-inclusive rate of the host API and the CPU restatement at 1 and 16 threads.
    python tools/bench_host_path.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oflibnumpy_amd as of
from oracle import np_oracle as O

h, w = 2160, 3840
f1 = of.Flow.from_transforms([['rotation', w / 2, h / 2, -30]], [h, w], 't')
f2 = of.Flow.from_transforms([['scaling', 800, 600, 0.8]], [h, w], 't')
of.native.ensure_device()
f1.combine_with(f2, 3)
t0 = time.perf_counter(); n = 5
for _ in range(n): f1.combine_with(f2, 3)
dt = (time.perf_counter() - t0) / n
print(json.dumps({"op": "host Flow.combine_with(mode=3) 4K incl. H2D/D2H + Flow construction", "ms": round(dt * 1e3, 2), "fields_per_s": round(1 / dt, 1)}))
img = np.random.default_rng(1).random((1080, 1920, 3), dtype=np.float32)
g = of.Flow.from_transforms([['rotation', 960, 540, -30]], [1080, 1920], 't')
g.apply(img, return_valid_area=True)
t0 = time.perf_counter()
for _ in range(n): g.apply(img, return_valid_area=True)
dt = (time.perf_counter() - t0) / n
print(json.dumps({"op": "host Flow.apply 1080p RGB f32 + valid incl. H2D/D2H", "ms": round(dt * 1e3, 2)}))
a, b = O.OFlow(f1.vecs, 't', f1.mask), O.OFlow(f2.vecs, 't', f2.mask)
for threads in (1, 16):
    O.set_threads(threads)
    a.combine_with(b, 3)
    t0 = time.perf_counter(); k = 3
    for _ in range(k): a.combine_with(b, 3)
    dt = (time.perf_counter() - t0) / k
    t0 = time.perf_counter()
    for _ in range(k): O.compose3_raw(f1.vecs, f1.mask, f2.vecs, f2.mask, -1)
    dr = (time.perf_counter() - t0) / k
    print(json.dumps({"op": "oracle OFlow.combine_with(mode=3) 4K", "threads": threads, "ms": round(dt * 1e3, 1),
                      "fields_per_s": round(1 / dt, 2), "fused_C_closed_form_ms": round(dr * 1e3, 1)}))
