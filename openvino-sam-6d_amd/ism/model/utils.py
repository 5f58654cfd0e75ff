"""Drop-in for the tensor bookkeeping of Instance_Segmentation_Model/model/utils.py (`BatchedData`, `Detections`): same class
and method names, argument meaning and in-place behaviour, with every tensor operation (area filter, NMS, row filtering)
running as HIP kernels from libsam6d_hip.so on the tensors' device.  File I/O (save_to_file / load_from_file / the BOP json
conversion) belongs to the reference's dataset tooling and is out of scope.

    remove_very_small_detections   ISM/model/utils.py:96-105
    apply_nms_per_object_id        ISM/model/utils.py:107-119
    apply_nms                      ISM/model/utils.py:121-126
    filter                         ISM/model/utils.py:188-190
"""
import numpy as np
import torch

from sam6d_hip import ism as _ism


class BatchedData:
    """Chunked view over a list / tensor (ISM/model/utils.py:44-80)."""

    def __init__(self, batch_size, data=None, **kwargs) -> None:
        self.batch_size = batch_size
        self.data = data if data is not None else []

    def __len__(self):
        assert self.batch_size is not None, "batch_size is not defined"
        return np.ceil(len(self.data) / self.batch_size).astype(int)

    def __getitem__(self, idx):
        assert self.batch_size is not None, "batch_size is not defined"
        return self.data[idx * self.batch_size: (idx + 1) * self.batch_size]

    def cat(self, data, dim=0):
        self.data = data if len(self.data) == 0 else torch.cat([self.data, data], dim=dim)

    def append(self, data):
        self.data.append(data)

    def stack(self, dim=0):
        self.data = torch.stack(self.data, dim=dim)


class Detections:
    """Proposals of one image: `masks`, `boxes` (int64 xyxy) and whatever add_attribute attaches (scores, object_ids)."""

    def __init__(self, data) -> None:
        if isinstance(data, str):
            raise NotImplementedError("Detections(file path): npz loading is part of the reference's dataset tooling")
        for key, value in data.items():
            setattr(self, key, value)
        self.keys = list(data.keys())
        if "boxes" in self.keys:
            if isinstance(self.boxes, np.ndarray):
                self.to_torch()
            self.boxes = self.boxes.long()

    def _take(self, idxs):
        for key in self.keys:
            setattr(self, key, _ism.take_rows(getattr(self, key), idxs))

    def remove_very_small_detections(self, config):
        keep = _ism.small_detection_keep(self.boxes, self.masks, config.min_box_size, config.min_mask_size)
        self._take(keep)

    def apply_nms_per_object_id(self, nms_thresh=0.5):
        self._take(_ism.nms(self.boxes, self.scores, nms_thresh, object_ids=self.object_ids))

    def apply_nms(self, nms_thresh=0.5):
        self._take(_ism.nms(self.boxes, self.scores, nms_thresh))

    def add_attribute(self, key, value):
        setattr(self, key, value)
        self.keys.append(key)

    def __len__(self):
        return len(self.boxes)

    def check_size(self):
        sizes = [len(self.masks), len(self.boxes), len(self.scores), len(self.object_ids)]
        assert len(set(sizes)) == 1, "Size mismatch %d %d %d %d" % tuple(sizes)

    def to_numpy(self):
        for key in self.keys:
            setattr(self, key, getattr(self, key).cpu().numpy())

    def to_torch(self):
        for key in self.keys:
            setattr(self, key, torch.from_numpy(getattr(self, key)))

    def filter(self, idxs):
        self._take(idxs)

    def clone(self):
        return Detections({k: getattr(self, k) for k in self.keys})
