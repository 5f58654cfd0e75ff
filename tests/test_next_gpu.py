"""GPU parity of the rows SURVEY 8f marks "next": ViTEncoder's radius normalisation (PEM/model/feature_extraction.py:133-137)
and CustomDINOv2's masked patch-descriptor post-processing (ISM/model/dinov2.py:265-269), through the C ABI, against the
CPU oracle (the same torch-CPU ops the reference calls)."""
import numpy as np
import pytest
import torch

from tests._util import PKG  # noqa: F401  (sys.path)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,Npo,Npm", [(1, 1, 1), (3, 2048, 2048), (2, 5000, 777), (32, 2048, 2048)])
def test_radius_normalize_bit_exact(dev, B, Npo, Npm):
    from oracle import pem_oracle as O
    from sam6d_hip import pem
    g = torch.Generator().manual_seed(B * 131 + Npo)
    po = (torch.rand(B, Npo, 3, generator=g) - 0.5) * torch.rand(B, 1, 1, generator=g) * 0.4
    pts = torch.randn(B, Npm, 3, generator=g) * 0.1
    pm_w, po_w, r_w = O.radius_normalize(pts, po)
    pm, po_g, r = pem.radius_normalize(pts.to(dev), po.to(dev))
    assert torch.equal(r.cpu(), r_w), "radius differs: %s vs %s" % (r.cpu(), r_w)
    assert torch.equal(pm.cpu(), pm_w)
    assert torch.equal(po_g.cpu(), po_w)


def test_radius_normalize_rejects_cpu_tensor():
    from sam6d_hip import pem
    with pytest.raises((RuntimeError, ValueError, AssertionError)):
        pem.radius_normalize(torch.zeros(1, 4, 3), torch.zeros(1, 4, 3))


@pytest.mark.parametrize("N,D,HW,patch", [(1, 1024, 224, 14), (5, 1024, 224, 14), (3, 64, 56, 7), (42, 1024, 224, 14)])
def test_masked_patch_features(dev, N, D, HW, patch):
    from oracle import ism_oracle as O
    from sam6d_hip import ism
    g = torch.Generator().manual_seed(N * 7 + D)
    P = (HW // patch) ** 2
    feats = torch.randn(N, P, D, generator=g)
    # binary proposal masks (what the detector passes): blobs with ragged borders so that many patches sit near 50 %
    yy, xx = torch.meshgrid(torch.arange(HW), torch.arange(HW), indexing="ij")
    masks = torch.zeros(N, HW, HW)
    for n in range(N):
        cx, cy, r = [float(v) for v in torch.rand(3, generator=g) * torch.tensor([HW, HW, HW / 2.0])]
        masks[n] = ((((xx - cx) ** 2 + (yy - cy) ** 2) < r * r) & (torch.rand(HW, HW, generator=g) > 0.3)).float()
    masks[0, : patch, : patch] = 0
    masks[0, 0 : patch // 2, 0:patch] = 1  # exactly 50 % coverage (even patch): "> 0.5" must drop this patch
    want = O.masked_patch_features(feats, masks, patch, 0.5)
    got = ism.masked_patch_features(feats.to(dev), masks.to(dev), patch, 0.5).cpu()
    assert torch.equal(got == 0, want == 0), "kept-patch sets differ"
    assert float(got[0, 0].abs().max()) == 0.0
    d = float((got - want).abs().max())
    assert d <= 2e-7, "max abs diff %.3e" % d


def test_masked_patch_features_soft_masks(dev):
    """Float (non-binary) masks: the pooled mean is compared with the threshold in fp32 just like nn.AvgPool2d."""
    from oracle import ism_oracle as O
    from sam6d_hip import ism
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(4, 256, 128, generator=g)
    masks = torch.rand(4, 224, 224, generator=g) * torch.linspace(0.2, 1.8, 224)[None, None, :]
    want = O.masked_patch_features(feats, masks, 14, 0.5)
    got = ism.masked_patch_features(feats.to(dev), masks.to(dev), 14, 0.5).cpu()
    pooled = torch.nn.AvgPool2d(14)(masks).flatten(-2)
    sure = (pooled - 0.5).abs() > 1e-5  # patches whose mean is not within summation-order noise of the threshold
    assert torch.equal((got == 0)[sure], (want == 0)[sure])
    same = (got == 0) == (want == 0)
    assert float((got - want)[same].abs().max()) <= 2e-7


def test_vit_encoder_forward_uses_hip_radius(dev):
    """The drop-in ViTEncoder.forward returns the reference's five outputs with the normalisation done on the GPU."""
    import importlib
    fe = importlib.import_module("feature_extraction")
    src = open(fe.__file__).read()
    assert "_pem.radius_normalize" in src and "torch.norm(dense_po" not in src
