#!/usr/bin/env python
"""Calibration: device-to-device copy rate (hipMemcpyAsync DtoD) on rotating 300 MB buffers, read + write bytes."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oflibnumpy_amd as of
from oflibnumpy_amd import device as dev
nat = of.native; nat.ensure_device(); lib = nat.load()
nbytes, nsets, iters = 300 << 20, 4, 60
bufs = [(dev.DeviceBuffer(nbytes), dev.DeviceBuffer(nbytes)) for _ in range(nsets)]
for a, b in bufs:
    nat.check(lib.ofl_memset(a.ptr, 1, nbytes, None))
def step(i):
    a, b = bufs[i % nsets]
    nat.check(lib.ofl_copy_dev(b.ptr, a.ptr, nbytes, None))
for i in range(6): step(i)
e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
nat.check(lib.ofl_event_create(ctypes.byref(e0))); nat.check(lib.ofl_event_create(ctypes.byref(e1)))
nat.check(lib.ofl_device_sync()); nat.check(lib.ofl_event_record(e0, None))
for i in range(iters): step(i)
nat.check(lib.ofl_event_record(e1, None)); nat.check(lib.ofl_device_sync())
ms = ctypes.c_float(); nat.check(lib.ofl_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
t = ms.value / iters * 1e-3
print(json.dumps({"op": "hipMemcpy DtoD 300 MiB", "us": round(t * 1e6, 1), "GBps_read_plus_write": round(2 * nbytes / t / 1e9, 1), "frac_of_8TBps": round(2 * nbytes / t / 8e12, 4)}))
