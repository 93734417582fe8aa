"""Array-in / array-out wrappers over the `Flow` hot-path methods (masks are dropped), with the same
names and signatures as the reference's src/oflibnumpy/flow_operations.py:70-228.
The visualisation helpers of that module are display-only and out of scope."""
from typing import Union

import numpy as np

from .flow_class import Flow

nd = np.ndarray
__all__ = ['combine_flows', 'switch_flow_ref', 'invert_flow', 'valid_target', 'valid_source', 'get_flow_padding']


def combine_flows(input_1: Union[Flow, nd], input_2: Union[Flow, nd], mode: int, ref: str = None,
                  thresholded: bool = None) -> Union[Flow, nd]:
    """flow_1 (+) flow_2 = flow_3; `mode` k computes flow_k from the two inputs (in formula order).
    Arrays in -> array out; two Flow objects are still accepted (reference flow_operations.py:153-161)."""
    if isinstance(input_1, Flow) and isinstance(input_2, Flow):
        print("AVOID - future deprecation warning: using combine_flows(flow_obj1, flow_obj2) is deprecated and may "
              "not work anymore in future versions - use flow_obj1.combine_with(flow_obj2) instead. combine_flows() "
              "will be reserved for use with NumPy arrays only.")
        return input_1.combine_with(input_2, mode=mode, thresholded=thresholded)
    return Flow(input_1, ref).combine_with(Flow(input_2, ref), mode=mode, thresholded=thresholded).vecs


def switch_flow_ref(flow: nd, input_ref: str) -> nd:
    """Vectors recalculated for the other reference (reference flow_operations.py:164-174)."""
    return Flow(flow, input_ref).switch_ref().vecs


def invert_flow(flow: nd, input_ref: str, output_ref: str = None) -> nd:
    """Inverse flow vectors (reference flow_operations.py:177-189)."""
    output_ref = input_ref if output_ref is None else output_ref
    return Flow(flow, input_ref).invert(output_ref).vecs


def valid_target(flow: nd, ref: str) -> nd:
    """Boolean valid area in the target domain (reference flow_operations.py:192-210)."""
    return Flow(flow, ref).valid_target()


def valid_source(flow: nd, ref: str) -> nd:
    """Boolean valid area in the source domain (reference flow_operations.py:213-228)."""
    return Flow(flow, ref).valid_source()


def get_flow_padding(flow: nd, ref: str) -> list:
    """Padding [top, bottom, left, right] needed to keep every warped pixel (reference flow_operations.py:231-248)."""
    return Flow(flow, ref).get_padding()
