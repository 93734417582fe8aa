#!/usr/bin/env python
"""Calibration: a pure streaming kernel with the SAME byte profile as mode-3 composition (read 2 x (8+1),
write 8+1 bytes per pixel): out = a + b, mout = ma & mb (ofl_axpy_dev), rotating over buffer sets larger
than the Infinity Cache.  Gives the HBM rate this access mix reaches without any gather."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev
nat = of.native; nat.ensure_device(); lib = nat.load()
h, w, nsets, iters = 2160, 3840, 6, 120
n = h * w
sets = [(dev.DeviceBuffer(n * 8), dev.DeviceBuffer(n), dev.DeviceBuffer(n * 8), dev.DeviceBuffer(n), dev.DeviceBuffer(n * 8), dev.DeviceBuffer(n)) for _ in range(nsets)]
for s in sets:
    for b in s:
        nat.check(lib.ofl_memset(b.ptr, 1, b.nbytes, None))
def step(i):
    a, ma, b, mb, o, mo = sets[i % nsets]
    nat.check(lib.ofl_axpy_dev(a.ptr, ma.ptr, b.ptr, mb.ptr, np.float32(1.0), n, o.ptr, mo.ptr, None))
for i in range(10): step(i)
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
nat.check(lib.ofl_event_create(ctypes.byref(e0))); nat.check(lib.ofl_event_create(ctypes.byref(e1)))
nat.check(lib.ofl_device_sync()); nat.check(lib.ofl_event_record(e0, None))
for i in range(iters): step(i)
nat.check(lib.ofl_event_record(e1, None)); nat.check(lib.ofl_device_sync())
ms = ctypes.c_float(); nat.check(lib.ofl_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
t = ms.value / iters * 1e-3
print(json.dumps({"op": "axpy stream 27 B/px", "us": round(t * 1e6, 2), "GBps": round(27 * n / t / 1e9, 1), "frac_of_8TBps": round(27 * n / t / 8e12, 4)}))
