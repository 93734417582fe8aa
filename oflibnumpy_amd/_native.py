"""ctypes binding of libofl_hip.so -- the C ABI declared in include/ofl.h.

There is no CPU fallback: if the library is missing, or no HIP device can be initialised, every
hot-path call raises.  (Host-side validation, constructors and operators work without a GPU.)
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OFL_LIB") or os.path.join(_HERE, "libofl_hip.so")   # OFL_LIB: A/B builds only
ABI_VERSION = 4          # OFL_ABI_VERSION of the include/ofl.h these prototypes were written for

# enums of include/ofl.h
OK, E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_NOPOINTS, E_RCCL = 0, -1, -2, -3, -4, -5, -6
U8, I16, U16, F32, F64 = 0, 1, 2, 3, 4
QUANT_OPENCV, QUANT_EXACT = 0, 1
ARITH_NATIVE, ARITH_FLOAT_RNE = 0, 1
RULE_EQ1, RULE_GE_HALF, RULE_GT_HALF = 0, 1, 2
SCATTER_ROUND, SCATTER_NEGATE, SCATTER_UNCERTIFIED = 0x100, 0x200, 0x400
STAT_NONZERO_MASKED, STAT_NONZERO_TH_MASKED, STAT_NONZERO, STAT_NONZERO_TH, STAT_NONFINITE, STAT_MASK_HAS_ZERO = 1, 2, 4, 8, 16, 32


class MeshCert(ctypes.Structure):
    """ofl_mesh_cert of include/ofl.h"""
    _fields_ = [("certified", ctypes.c_uint32), ("folded_cells", ctypes.c_uint32), ("bad_edges", ctypes.c_uint32),
                ("dropped", ctypes.c_uint32), ("border_dev", ctypes.c_double), ("corner", (ctypes.c_double * 2) * 4),
                ("diag_bits", ctypes.c_void_p)]


class NativeError(RuntimeError):
    """A libofl_hip entry point returned an error code."""

    def __init__(self, code, message):
        super().__init__("libofl_hip error {}: {}".format(code, message))
        self.code = code


class NoDeviceError(NativeError):
    """No usable HIP device: the MI355X engine has no CPU fallback."""


_vp, _ci, _cs, _cf, _cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float, ctypes.c_double
_pvp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); must list every symbol of include/ofl.h (tests check this)
SIGNATURES = {
    "ofl_abi_version": (_ci, []),
    "ofl_last_error": (ctypes.c_char_p, []),
    "ofl_device_count": (_ci, [ctypes.POINTER(_ci)]),
    "ofl_init": (_ci, [_ci]),
    "ofl_device_name": (_ci, [ctypes.c_char_p, _cs]),
    "ofl_malloc": (_ci, [_pvp, _cs]),
    "ofl_free": (_ci, [_vp]),
    "ofl_memset": (_ci, [_vp, _ci, _cs, _vp]),
    "ofl_upload": (_ci, [_vp, _vp, _cs, _vp]),
    "ofl_download": (_ci, [_vp, _vp, _cs, _vp]),
    "ofl_download_async": (_ci, [_vp, _vp, _cs, _vp]),
    "ofl_copy_dev": (_ci, [_vp, _vp, _cs, _vp]),
    "ofl_host_alloc": (_ci, [_pvp, _cs]),
    "ofl_host_free": (_ci, [_vp]),
    "ofl_stream_create": (_ci, [_pvp]),
    "ofl_stream_destroy": (_ci, [_vp]),
    "ofl_stream_sync": (_ci, [_vp]),
    "ofl_device_sync": (_ci, []),
    "ofl_event_create": (_ci, [_pvp]),
    "ofl_event_destroy": (_ci, [_vp]),
    "ofl_event_record": (_ci, [_vp, _vp]),
    "ofl_event_sync": (_ci, [_vp]),
    "ofl_event_elapsed_ms": (_ci, [_vp, _vp, ctypes.POINTER(_cf)]),
    "ofl_mem_info": (_ci, [ctypes.POINTER(_cs), ctypes.POINTER(_cs)]),
    "ofl_png_unfilter": (_ci, [_vp, _cs, _ci, _ci, _ci, _vp]),
    "ofl_compose3_dev": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _ci, _vp]),
    "ofl_compose3": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _ci]),
    "ofl_mask_bits_bytes": (_ci, [_ci, _ci, _ci, ctypes.POINTER(_cs)]),
    "ofl_mask_pack_dev": (_ci, [_vp, _ci, _ci, _ci, _vp, _vp]),
    "ofl_mask_unpack_dev": (_ci, [_vp, _ci, _ci, _ci, _vp, _vp]),
    "ofl_compose3_bits_dev": (_ci, [_vp, _vp, _vp, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp]),
    "ofl_gather_rows_dev": (_ci, [_vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _vp, _ci, _ci, _ci, _vp]),
    "ofl_gather_bilinear_dev": (_ci, [_vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp,
                                      _ci, _ci, _ci, _vp]),
    "ofl_gather_bilinear": (_ci, [_vp, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _vp, _vp, _vp,
                                  _ci, _ci, _ci]),
    "ofl_gather_bilinear_batch_dev": (_ci, [_vp, _ci, _ci, _ci, _ci, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _ci, _vp, _ci, _vp,
                                            _vp, _vp, _ci, _ci, _ci, _vp]),
    "ofl_flow_stats_dev": (_ci, [_vp, _vp, _cs, _cf, _vp, _vp]),
    "ofl_flow_stats": (_ci, [_vp, _vp, _cs, _cf, _vp]),
    "ofl_axpy_dev": (_ci, [_vp, _vp, _vp, _vp, _cf, _cs, _vp, _vp, _vp]),
    "ofl_scatter_linear_f64_dev": (_ci, [_vp, _ci, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _vp, _vp, _ci, _vp, _cs, _vp, _vp]),
    "ofl_scatter_rows_dev": (_ci, [_vp, _ci, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _cs, _vp, _vp]),
    "ofl_scatter_slab_stars_dev": (_ci, [_vp, _ci, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _cs, _vp, _cs, _vp]),
    "ofl_scatter_slab_finish_dev": (_ci, [_vp, _ci, _ci, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _cs, _ci, _vp, _vp, _ci, _vp, _cs, _vp, _vp]),
    "ofl_scatter_linear_dev": (_ci, [_vp, _ci, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _vp, _vp, _vp, _ci, _vp, _cs, _vp, _vp]),
    "ofl_scatter_diag_bytes": (_ci, [_ci, _ci, ctypes.POINTER(_cs)]),
    "ofl_scatter_certify_dev": (_ci, [_vp, _ci, _ci, _vp, _ci, _ci, _vp, _cs, _vp, _vp, _vp]),
    "ofl_scatter_certified_dev": (_ci, [_vp, _ci, _ci, _vp, _ci, _vp, _ci, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp]),
    "ofl_scatter_workspace_bytes": (_ci, [_ci, _ci, _ci, ctypes.POINTER(_cs)]),
    "ofl_scatter_linear": (_ci, [_vp, _ci, _ci, _vp, _vp, _ci, _vp, _ci, _ci, _vp, _vp, _vp, _ci]),
    "ofl_sample_points_dev": (_ci, [_vp, _ci, _ci, _vp, _cs, _vp, _vp]),
    "ofl_scatter_query_dev": (_ci, [_vp, _ci, _ci, _vp, _vp, _ci, _ci, _ci, _vp, _cs, _vp, _vp, _vp, _cs, _vp]),
    "ofl_flow_extent_dev": (_ci, [_vp, _vp, _ci, _ci, _ci, _cf, _vp, _vp]),
    "ofl_convert_dev": (_ci, [_vp, _ci, _vp, _ci, _cs, _vp]),
    "ofl_mask_and_dev": (_ci, [_vp, _vp, _vp, _cs, _vp]),
    "ofl_grid_offset_dev": (_ci, [_vp, _ci, _ci, _ci, _vp, _vp]),
    "ofl_resize_flow": (_ci, [_vp, _vp, _ci, _ci, _ci, _ci, _cd, _cd, _cf, _cf, _vp, _vp]),
    "ofl_resize_flow_dev": (_ci, [_vp, _vp, _ci, _ci, _ci, _ci, _cd, _cd, _cf, _cf, _vp, _vp, _vp]),
    "ofl_comm_unique_id": (_ci, [_vp]),
    "ofl_comm_init": (_ci, [_vp, _ci, _ci]),
    "ofl_comm_broadcast": (_ci, [_vp, _cs, _ci, _vp]),
    "ofl_comm_allgather": (_ci, [_vp, _vp, _cs, _vp]),
    "ofl_comm_size": (_ci, [ctypes.POINTER(_ci)]),
    "ofl_comm_destroy": (_ci, []),
}

_lib = None
_lock = threading.RLock()
_device = None


def load():
    """dlopen libofl_hip.so and attach the prototypes.  Raises if the library was not built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        "{} is missing: build it with `python -m oflibnumpy_amd.build_native` "
                        "(hipcc, gfx950).  oflibnumpy_amd has no CPU fallback.".format(LIB_PATH))
                lib = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(lib, name)      # AttributeError if the ABI lacks a declared symbol
                    fn.restype, fn.argtypes = res, args
                got = lib.ofl_abi_version()
                if got != ABI_VERSION:
                    # a stale build (the .so files are git-ignored and travel with the snapshot; OFL_LIB may point at an old
                    # experiments build): same symbol names, shifted arguments -- refuse rather than call it
                    raise ImportError("{} has ABI version {}, these bindings need {}: rebuild it with "
                                      "`python -m oflibnumpy_amd.build_native --force --experiments`".format(LIB_PATH, got, ABI_VERSION))
                _lib = lib
    return _lib


def last_error():
    return load().ofl_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != OK:
        msg = last_error()
        if rc == E_NODEVICE:
            raise NoDeviceError(rc, msg)
        if rc == E_NOPOINTS:
            raise ValueError("No points given")      # what qhull raises in the reference
        raise NativeError(rc, msg)


def device_count():
    n = _ci(0)
    check(load().ofl_device_count(ctypes.byref(n)))
    return n.value


def ensure_device():
    """Select this process's GPU once (LOCAL_RANK under torchrun, else OFL_DEVICE, else 0)."""
    global _device
    if _device is None:
        with _lock:
            if _device is None:
                n = device_count()
                if n <= 0:
                    raise NoDeviceError(E_NODEVICE, "no HIP device visible; oflibnumpy_amd needs an AMD GPU "
                                                    "(MI355X / gfx950) and has no CPU fallback")
                dev = int(os.environ.get("OFL_DEVICE", os.environ.get("LOCAL_RANK", "0"))) % n
                check(load().ofl_init(dev))
                _device = dev
    return _device


def device_name():
    ensure_device()
    buf = ctypes.create_string_buffer(256)
    check(load().ofl_device_name(buf, 256))
    return buf.value.decode()
