"""Drop-in boundary B2: the PEM module classes keep the reference's names, constructor arguments and
state_dict inventory.  CPU-only checks here; the GPU behaviour is in test_dropin_gpu.py."""
import importlib

import pytest
import torch

from tests._util import ROOT  # noqa: F401
from sam6d_hip import synth


@pytest.fixture(scope="module")
def net():
    mod = importlib.import_module("pose_estimation_model")  # as PEM/run_inference_custom_pytorch.py:383-386 does
    return mod.Net(synth.default_model_cfg())


def test_state_dict_inventory(net):
    sd = net.state_dict()
    hot = {k: v for k, v in sd.items() if not k.startswith("feature_extraction.")}
    want = dict(synth.pem_param_shapes())
    assert set(hot) == set(want)
    for k, shape in want.items():
        assert tuple(hot[k].shape) == tuple(shape), k
    # timm-compatible ViT keys of the (out-of-path) backbone
    for k in ("feature_extraction.rgb_net.vit.pos_embed", "feature_extraction.rgb_net.vit.blocks.11.attn.qkv.weight",
              "feature_extraction.rgb_net.vit.blocks.0.mlp.fc1.bias", "feature_extraction.rgb_net.output_upscaling.weight"):
        assert k in sd
    assert tuple(sd["feature_extraction.rgb_net.output_upscaling.weight"].shape) == (4096, 3072)


def test_load_reference_keyed_weights_strict(net):
    res = net.load_state_dict(synth.make_pem_weights(1), strict=False)
    assert not res.unexpected_keys
    assert all(k.startswith("feature_extraction.") for k in res.missing_keys)


def test_forward_refuses_cpu(net):
    net.eval()
    with pytest.raises(RuntimeError):
        net.geo_embedding(torch.zeros(1, 197, 3))


def test_module_names_match_reference_layout():
    for name, attrs in (("transformer", ["GeometricStructureEmbedding", "SinusoidalPositionalEmbedding",
                                         "RPEMultiHeadAttention", "MultiHeadAttention", "AttentionOutput",
                                         "GeometricTransformer", "LinearAttention", "SparseToDenseTransformer"]),
                        ("model_utils", ["sample_pts_feats", "compute_feature_similarity", "compute_coarse_Rt",
                                         "compute_fine_Rt", "weighted_procrustes", "WeightedProcrustes"]),
                        ("coarse_point_matching", ["CoarsePointMatching"]),
                        ("fine_point_matching", ["FinePointMatching", "PositionalEncoding"]),
                        ("pointnet2_utils", ["furthest_point_sample", "gather_operation", "ball_query",
                                             "grouping_operation", "QueryAndGroup"])):
        m = importlib.import_module(name)
        for a in attrs:
            assert hasattr(m, a), (name, a)
