"""ctypes binding of libsam6d_hip.so (the C ABI declared in include/sam6d_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises, and every op
raises RuntimeError on a non-zero return code (message from sam6d_last_error()).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAM6D_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libsam6d_hip.so")  # (SAM6D_LIB: a diagnostic build)

c_f = ctypes.c_float
c_i = ctypes.c_int
ABI_VERSION = 2  # include/sam6d_hip.h SAM6D_ABI_VERSION (tests/test_abi.py keeps the two equal)
c_l = ctypes.c_long
c_p = ctypes.c_void_p

# name -> argtypes (all return int).  Must mirror include/sam6d_hip.h exactly; tests/test_abi.py parses the header
# and checks that every declared symbol is exported and listed here.
SIGNATURES = {
    "sam6d_furthest_point_sampling": [c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "sam6d_fps_debug_spin_cap": [c_l],
    "sam6d_gather_points": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "sam6d_ball_query": [c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_p, c_p],
    "sam6d_ball_query2": [c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_p, c_f, c_i, c_p, c_p],
    "sam6d_ball_query2_grid_workspace_bytes": [c_i, c_i],
    "sam6d_ball_query2_grid": [c_p, c_p, c_i, c_i, c_i, c_f, c_i, c_p, c_f, c_i, c_p, c_p, ctypes.c_size_t, c_p],
    "sam6d_group_points": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p],
    "sam6d_gather_rows": [c_p, c_p, c_i, c_i, c_i, c_i, c_l, c_l, c_i, c_p, c_p],
    "sam6d_gather_rows_lead": [c_p, c_p, c_i, c_i, c_i, c_i, c_l, c_l, c_i, c_p, c_l, c_p, c_p],
    "sam6d_gemm_nt": [c_p] * 6 + [c_i] * 3 + [c_l] * 4 + [c_i] + [c_l] * 4 + [c_f, c_i, c_p],
    "sam6d_gemm_nt_w16": [c_p] * 4 + [c_f] + [c_p] * 4 + [c_i] * 3 + [c_l] * 4 + [c_i] + [c_l] * 4 + [c_f, c_i, c_p],
    "sam6d_set_matmul_mode": [c_i],
    "sam6d_set_thread_matmul_mode": [c_i],
    "sam6d_layernorm256": [c_p, c_p, c_p, c_p, c_l, c_l, c_l, c_f, c_p],
    "sam6d_gemm_ln256": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_l, c_l, c_l, c_l, c_f, c_p],
    "sam6d_geo_embedding": [c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p, c_p],
    "sam6d_geo_indices": [c_p, c_i, c_i, c_f, c_f, c_i, c_p, c_p, c_p],
    "sam6d_geo_embed": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p, c_p],
    "sam6d_split_f16": [c_p, c_l, c_f, c_p, c_p, c_p],
    "sam6d_geo_embed_h3": [c_p, c_l, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p],
    "sam6d_gemm_nt_b2": [c_p, c_p, c_p, c_i, c_i, c_i, c_l, c_l, c_l, c_i, c_l, c_l, c_l, c_i, c_l, c_l, c_l, c_p],
    "sam6d_geo_outliers": [c_p, c_l, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "sam6d_rpe_scores": [c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p],
    "sam6d_rpe_geo_scores": [c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_l, c_i, c_i, c_p],
    "sam6d_rpe_self_attention": [c_p, c_p, c_p, c_i, c_i, c_i, c_p],
    "sam6d_geo_outliers2": [c_p, c_l, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    "sam6d_rpe_scores2": [c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_p],
    "sam6d_transpose": [c_p, c_l, c_l, c_i, c_i, c_i, c_p, c_l, c_l, c_p],
    "sam6d_geo_embed_cheb": [c_p, c_l, c_p, c_f, c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p],
    "sam6d_attention": [c_p] * 6 + [c_i] * 3 + [c_l] * 8 + [c_p],
    "sam6d_scaled_softmax": [c_p, c_p, c_f, c_l, c_i, c_l, c_l, c_p, c_l, c_p],
    "sam6d_sinusoid_embed": [c_p, c_l, c_p, c_i, c_p, c_p],
    "sam6d_linattn_focus_k": [c_p, c_p, c_l, c_l, c_p],
    "sam6d_linattn_kv": [c_p, c_p, c_i, c_i, c_l, c_l, c_l, c_l, c_p, c_p, c_p],
    "sam6d_linattn_focus_q": [c_p, c_p, c_p, c_i, c_l, c_l, c_p],
    "sam6d_pe_mlp_max": [c_p, c_p, c_i, c_i, c_i] + [c_p] * 10 + [c_l, c_i, c_p],
    "sam6d_pe_mlp_max_wg": [c_p, c_p, c_i, c_i, c_i] + [c_p] * 10 + [c_l, c_i, c_i, c_p],
    "sam6d_rigid_inverse": [c_p, c_p, c_p, c_i, c_i, c_p, c_p],
    "sam6d_put_rows": [c_p, c_l, c_l, c_p, c_l, c_l, c_i, c_i, c_i, c_p],
    "sam6d_prepend_bg_point": [c_p, c_i, c_i, c_p, c_p],
    "sam6d_add_scalar": [c_p, c_f, c_l, c_p, c_p],
    "sam6d_copy_f32": [c_p, c_p, c_l, c_p],
    "sam6d_batch_rows_equal": [c_p, c_i, c_l, c_p, c_p],
    "sam6d_l2norm256": [c_p, c_p, c_l, c_l, c_l, c_p],
    "sam6d_soft_assign": [c_p, c_i, c_i, c_i] + [c_p] * 7 + [c_l, c_p],
    "sam6d_coarse_weights": [c_p, c_i, c_i, c_i] + [c_p] * 9,
    "sam6d_coarse_soft_assign": [c_p, c_i, c_i, c_i] + [c_p] * 9,
    "sam6d_weighted_sample": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "sam6d_coarse_hypotheses": [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "sam6d_select_smallest": [c_p, c_i, c_i, c_i, c_p, c_p],
    "sam6d_score_select_hypotheses": [c_p] * 7 + [c_i] * 5 + [c_p] * 5,
    "sam6d_score_select_workspace_bytes": [c_i, c_i, c_i],
    "sam6d_score_select_hypotheses_ws": [c_p] * 7 + [c_i] * 5 + [c_p] * 5 + [ctypes.c_size_t, c_p],
    "sam6d_fine_assign": [c_p, c_i, c_i, c_i] + [c_p] * 10,
    "sam6d_weighted_procrustes": [c_p, c_p, c_p, c_i, c_i, c_f, c_f, c_p, c_p, c_p],
    "sam6d_ism_cosine": [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p],
    "sam6d_ism_semantic": [c_p, c_i, c_i, c_i, c_i, c_f] + [c_p] * 6,
    "sam6d_ism_patch_scores": [c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p],
    "sam6d_ism_patch_fused_workspace_bytes": [c_i, c_i],
    "sam6d_ism_patch_fused": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, ctypes.c_size_t, c_p],
    "sam6d_ism_patch_fused_scores": [c_p, c_i, c_i, c_f, c_p, c_p, c_p],
    "sam6d_ism_project": [c_p, c_p, c_p, ctypes.c_double, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "sam6d_ism_project_workspace_doubles": [c_i, c_i, c_i],
    "sam6d_ism_project2": [c_p, c_i, c_p, c_p, c_p, ctypes.c_double, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    "sam6d_ism_translate_maps": [c_p, c_p, ctypes.c_double, c_i, c_i, c_i, c_p, c_p, c_p],
    "sam6d_ism_iou": [c_p, c_p, c_i, c_p, c_p, c_p],
    "sam6d_ism_semantic_compact": [c_p, c_i, c_i, c_i, c_i, c_f] + [c_p] * 9,
    "sam6d_ism_final_score_flag": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "sam6d_ism_final_score": [c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_p],
    "sam6d_radius_normalize": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p],
    "sam6d_masked_patch_normalize": [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p],
    "sam6d_depth_to_cloud": [c_p] + [c_i] * 6 + [c_f] * 4 + [c_p, c_p],
    "sam6d_detections_small_keep": [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p],
    "sam6d_mask_to_indices": [c_p, c_i, c_p, c_p, c_p],
    "sam6d_take_rows": [c_p, c_p, c_l, c_i, c_l, c_p, c_p],
    "sam6d_mask_rle_count": [c_p, c_i, c_i, c_i, c_p, c_p],
    "sam6d_mask_rle_encode": [c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "sam6d_mask_rle_decode": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "sam6d_nms_workspace_bytes": [c_i],
    "sam6d_nms": [c_p, c_p, c_p, c_i, c_f, c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "sam6d_mask_bbox": [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p],
    "sam6d_crop_masked_points": [c_p, c_p, c_i, c_i, c_i, c_p, c_f, c_f, c_f, c_f, c_i, c_p, c_p, c_p, c_p],
    "sam6d_radius_filter": [c_i, c_i, c_p, c_f, c_p, c_p, c_p, c_p, c_p],
    "sam6d_choose_points": [c_i, c_i, c_p, c_p, c_p, c_p, c_i, c_i, c_p, c_p, c_p],
    "sam6d_pairwise_distance": [c_p, c_p, c_i, c_i, c_i, c_p, c_p],
    "sam6d_fine_score": [c_p] * 6 + [c_i] * 3 + [c_f, c_p, c_p, c_p],
    "sam6d_fine_match_workspace_bytes": [c_i],
    "sam6d_fine_match_workspace_bytes_n": [c_i, c_i],
    "sam6d_fine_match": [c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "sam6d_fine_match_split": [c_p, c_p, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_size_t, c_p],
    "sam6d_rows_linear": [c_p, c_p, c_i, c_p, c_f, c_p, c_l, c_i, c_l, c_l, c_l, c_l, c_p],
    "sam6d_linear_norm_split": [c_p, c_p, c_p, c_f, c_p, c_p, c_l, c_p],
    "sam6d_cross_attention": [c_p, c_p, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p],
    "sam6d_cross_attention_kv": [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_f, c_p, c_i, c_i, c_i, c_p],
    "sam6d_pack_panels": [c_p, c_l, c_i, c_i, c_i, c_f, c_p, c_p],
    "sam6d_rpe_front_image_bytes": [],
    "sam6d_rpe_front": [c_p, c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_p, c_l, c_p],
    "sam6d_rpe_front_vt": [c_p, c_p, c_p, c_f, c_f, c_f, c_p, c_p, c_p, c_l, c_p, c_i, c_i, c_p],
    "sam6d_token_block_image_bytes": [c_i],
    "sam6d_linattn_kv_image_bytes": [],
    "sam6d_cross_attention_kv_image_bytes": [],
    "sam6d_linattn_kv_pack": [c_p, c_i, c_p, c_p, c_p],
    "sam6d_linattn_kv_image": [c_p, c_p, c_i, c_i, c_l, c_l, c_p, c_p, c_p, c_p],
    "sam6d_token_block": [c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_p],
    "sam6d_linattn_layer": [c_p] * 7 + [c_i, c_i, c_i, c_f, c_p],
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsam6d_hip.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C openvino-sam-6d_amd/csrc). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.sam6d_last_error.restype = ctypes.c_char_p
    lib.sam6d_last_error.argtypes = []
    lib.sam6d_abi_version.restype = c_i
    lib.sam6d_abi_version.argtypes = []
    if lib.sam6d_abi_version() != ABI_VERSION:
        raise ImportError("libsam6d_hip.so has ABI version %d, this binding was written for %d (include/sam6d_hip.h SAM6D_ABI_VERSION): "
                          "rebuild the library" % (lib.sam6d_abi_version(), ABI_VERSION))
    lib.sam6d_get_matmul_mode.restype = c_i
    lib.sam6d_get_matmul_mode.argtypes = []
    lib.sam6d_get_thread_matmul_mode.restype = c_i
    lib.sam6d_get_thread_matmul_mode.argtypes = []
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: loud by design
        fn.argtypes = args
        fn.restype = (c_l if "image_bytes" in name else ctypes.c_size_t) if name.endswith(("_bytes", "_bytes_n", "_doubles")) else c_i
    mode = os.environ.get("SAM6D_MATMUL_MODE")  # 0 = exact fp32 MFMA, 1 = fp16x3 split (library default)
    if mode is not None:
        if lib.sam6d_set_matmul_mode(int(mode)) != 0:
            raise RuntimeError("SAM6D_MATMUL_MODE=%s: %s" % (mode, lib.sam6d_last_error().decode()))
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (name, rc, lib.sam6d_last_error().decode()))
