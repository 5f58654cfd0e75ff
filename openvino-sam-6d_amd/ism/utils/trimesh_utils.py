"""Drop-in for ISM/utils/trimesh_utils.py::depth_image_to_pointcloud_translate_torch (:77-105): the mean back-projected
point of each masked depth map.  depth here is the already masked (N,H,W) tensor, as the detector passes it."""
import torch

from sam6d_hip import ism as _ism


def depth_image_to_pointcloud_translate_torch(depth, scale, K):
    N, H, W = depth.shape
    dev = depth.device
    ones = torch.ones(N, H, W, device=dev)
    ident = torch.eye(4, device=dev).unsqueeze(0)
    pc = torch.zeros(1, 1, 3, device=dev)
    zero = torch.zeros(N, dtype=torch.int32, device=dev)
    tr = []
    # the kernel multiplies mask * depth itself: feed each masked depth map as `depth` with an all-ones mask
    for i in range(N):
        _, _, t = _ism.project_template_to_image(zero[:1], zero[:1], ident, pc, ones[i:i + 1], depth[i], K, float(scale))
        tr.append(t)
    return torch.cat(tr, 0)
