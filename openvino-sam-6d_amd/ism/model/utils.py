"""Drop-in for the tensor bookkeeping of Instance_Segmentation_Model/model/utils.py (`BatchedData`, `Detections`): same class
and method names, argument meaning and in-place behaviour, with every tensor operation (area filter, NMS, row filtering)
running as HIP kernels from libsam6d_hip.so on the tensors' device.  The on-disk seam to PEM is here too: save_to_file /
load_from_file (npz) and convert_npz_to_json + save_json_bop23 (detection_ism.json), with the per-pixel Python loop of
mask_to_rle replaced by the RLE kernels (sam6d_mask_rle_count / _encode).

    remove_very_small_detections   ISM/model/utils.py:96-105
    apply_nms_per_object_id        ISM/model/utils.py:107-119
    apply_nms                      ISM/model/utils.py:121-126
    filter                         ISM/model/utils.py:188-190
    save_to_file / load_from_file  ISM/model/utils.py:153-186
    mask_to_rle, convert_npz_to_json  ISM/model/utils.py:25-43, 199-216;  save_json_bop23  ISM/utils/inout.py:57-60
"""
import json

import numpy as np
import torch

from sam6d_hip import ism as _ism


lmo_object_ids = np.array([1, 5, 6, 8, 9, 10, 11, 12])  # object ID of occlusionLINEMOD is different (ISM/model/utils.py:8-22)


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("ism.model.utils: the RLE kernels need a HIP device (there is no CPU path)")
    return torch.device("cuda", torch.cuda.current_device())


def xyxy_to_xywh(bbox):
    """ISM/utils/bbox_utils.py:129-138."""
    if len(bbox.shape) == 1:
        x1, y1, x2, y2 = bbox
        return [x1, y1, x2 - x1 + 1, y2 - y1 + 1]
    elif len(bbox.shape) == 2:
        return np.stack([bbox[:, 0], bbox[:, 1], bbox[:, 2] - bbox[:, 0], bbox[:, 3] - bbox[:, 1]], axis=1)
    raise ValueError("bbox must be a numpy array of shape (4,) or (N, 4)")


def xywh_to_xyxy(bbox):
    """ISM/utils/bbox_utils.py:141-153."""
    if len(bbox.shape) == 1:
        x, y, w, h = bbox
        return [x, y, x + w - 1, y + h - 1]
    elif len(bbox.shape) == 2:
        return np.stack([bbox[:, 0], bbox[:, 1], bbox[:, 0] + bbox[:, 2], bbox[:, 1] + bbox[:, 3]], axis=1)
    raise ValueError("bbox must be a numpy array of shape (4,) or (N, 4)")


def mask_to_rle(binary_mask):
    """One (H,W) mask -> {"counts", "size"} (ISM/model/utils.py:25-43), on the HIP device."""
    m = torch.as_tensor(np.asarray(binary_mask)) if not torch.is_tensor(binary_mask) else binary_mask
    return _ism.mask_to_rle(m[None].to(_device()))[0]


def convert_npz_to_json(idx, list_npz_paths):
    """ISM/model/utils.py:199-216: one npz written by Detections.save_to_file -> the list of BOP-style records; all masks of
    the file are RLE-encoded in one launch pair (force_binary_mask = `> 0` inside the kernel)."""
    detections = np.load(list_npz_paths[idx])
    seg = _ism.mask_to_rle(torch.from_numpy(np.asarray(detections["segmentation"])).to(_device()))
    results = []
    for idx_det in range(len(detections["bbox"])):
        results.append({
            "scene_id": int(detections["scene_id"]),
            "image_id": int(detections["image_id"]),
            "category_id": int(detections["category_id"][idx_det]),
            "bbox": detections["bbox"][idx_det].tolist(),
            "score": float(detections["score"][idx_det]),
            "time": float(detections["time"]),
            "segmentation": seg[idx_det],
        })
    return results


def save_json_bop23(path, info):
    """ISM/utils/inout.py:57-60: dump without sorting keys or changing format."""
    with open(path, "w") as f:
        json.dump(info, f)


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
            data = self.load_from_file(data)
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

    def save_to_file(self, scene_id, frame_id, runtime, file_path, dataset_name, return_results=False):
        """scene_id, image_id, category_id, bbox, time (ISM/model/utils.py:153-173); expects to_numpy() first, like the reference."""
        boxes = xyxy_to_xywh(self.boxes)
        results = {
            "scene_id": scene_id,
            "image_id": frame_id,
            "category_id": self.object_ids + 1 if dataset_name != "lmo" else lmo_object_ids[self.object_ids],
            "score": self.scores,
            "bbox": boxes,
            "time": runtime,
            "segmentation": self.masks,
        }
        np.savez_compressed(file_path, **results)
        if return_results:
            return results

    def load_from_file(self, file_path):
        """ISM/model/utils.py:175-186 (numeric arrays only: np.load refuses pickled members by default)."""
        data = np.load(file_path)
        return {
            "object_ids": data["category_id"] - 1,
            "bbox": xywh_to_xyxy(np.array(data["bbox"])),
            "scores": data["score"],
            "masks": data["segmentation"],
        }

    def filter(self, idxs):
        self._take(idxs)

    def clone(self):
        return Detections({k: getattr(self, k) for k in self.keys})
