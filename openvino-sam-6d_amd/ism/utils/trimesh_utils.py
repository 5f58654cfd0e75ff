"""Drop-in for ISM/utils/trimesh_utils.py::depth_image_to_pointcloud_translate_torch (:77-105): the mean back-projected
point of each masked depth map.  depth here is the already masked (N,H,W) tensor, as the detector passes it."""
from sam6d_hip import ism as _ism


def depth_image_to_pointcloud_translate_torch(depth, scale, K):
    """depth (N,H,W) masked depth maps, scale (depth_scale), K (3,3) -> (N,3): one launch for all N maps (sam6d_ism_translate_maps);
    no (N,H,W) helper tensor is built."""
    return _ism.translate_masked_depth_maps(depth, K, float(scale))
