"""Drop-in for ISM/utils/bbox_utils.py::compute_iou (:197-222)."""
from sam6d_hip import ism as _ism


def compute_iou(bb_a, bb_b):
    return _ism.compute_iou(bb_a, bb_b)
