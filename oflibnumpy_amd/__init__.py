"""oflibnumpy_amd -- MI355X-native engine for the oflibnumpy hot path.

Same import surface as the reference for that path:

    import oflibnumpy_amd as of
    f3 = of.Flow.from_transforms(...).combine_with(other, mode=3)
    of.combine_flows(a, b, 3, 't'); of.apply_flow(flow, img, 't')

Everything numeric runs in hand-written HIP kernels for gfx950 behind the C ABI of include/ofl.h
(libofl_hip.so, bound with ctypes).  There is no CPU fallback.
"""
from .flow_class import Flow
from .flow_operations import *
from .utils import from_matrix, from_transforms, load_sintel, save_sintel, load_sintel_mask, load_kitti, apply_flow, is_zero_flow, threshold_vectors, track_pts, resize_flow
from .device import DeviceFlow, DeviceImage, DeviceBuffer
from .batch import DeviceFlowBatch, combine_flows_batch
from . import _native as native

__version__ = "0.1.0"
