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


# ---------------------------------------------------------------------------- depth back-projection (row f2)
def test_depth_to_cloud_golden_bit_exact(dev):
    from tests._util import golden
    from sam6d_hip import pem
    g = golden("depth_cloud")
    depth = torch.from_numpy(g["depth"]).to(dev)
    crop = pem.depth_to_cloud(depth, g["K"], [int(v) for v in g["bbox"]]).cpu().numpy()
    assert np.array_equal(crop, g["crop"])
    import hashlib
    full = pem.depth_to_cloud(depth, g["K"]).cpu().numpy()
    assert hashlib.sha256(np.ascontiguousarray(full).tobytes()).hexdigest() == str(g["full_sha"])
    with pytest.raises(RuntimeError):
        pem.depth_to_cloud(depth, g["K"], [0, 121, 0, 10])  # bbox outside the map


# ---------------------------------------------------------------------------- Detections bookkeeping (row f3)
def _rand_boxes(g, n, size=480):
    xy = torch.rand(n, 2, generator=g) * (size - 80)
    wh = torch.rand(n, 2, generator=g) * 70 + 4
    return torch.cat([xy, xy + wh], 1).round()


@pytest.mark.parametrize("N,thr", [(0, 0.5), (1, 0.5), (37, 0.25), (300, 0.5), (1000, 0.25), (4500, 0.7)])
def test_nms_matches_oracle(dev, N, thr):
    from oracle import ism_oracle as O
    from sam6d_hip import ism
    g = torch.Generator().manual_seed(N + 11)
    boxes = _rand_boxes(g, N)
    if N > 20:  # clusters of near-duplicates around the first boxes + exact score ties
        boxes[N // 2:] = boxes[: N - N // 2] + torch.randint(-3, 4, (N - N // 2, 4), generator=g).float()
    scores = (torch.rand(N, generator=g) * 50).round() / 50  # many ties -> the stable order matters
    want = O.nms(boxes, scores, thr)
    got = ism.nms(boxes.to(dev), scores.to(dev), thr).cpu()
    assert got.dtype == torch.int64 and torch.equal(got, want)


@pytest.mark.parametrize("N,n_obj", [(1, 1), (64, 3), (500, 7)])
def test_nms_per_object_id_matches_oracle(dev, N, n_obj):
    from oracle import ism_oracle as O
    from sam6d_hip import ism
    g = torch.Generator().manual_seed(N * 3 + n_obj)
    boxes = _rand_boxes(g, N, 200)
    scores = torch.rand(N, generator=g)
    oid = torch.randint(0, n_obj, (N,), generator=g) * 5 - 3  # non-contiguous ids, one negative
    want = O.nms_per_object_id(boxes, scores, oid, 0.5)
    got = ism.nms(boxes.to(dev), scores.to(dev), 0.5, object_ids=oid.to(dev)).cpu()
    assert torch.equal(got, want)


def test_detections_class_matches_reference_semantics(dev):
    """The drop-in Detections (ISM/model/utils.py:84-196): area filter -> attributes -> per-id NMS -> filter, against the
    oracle's restatement of every step."""
    import importlib
    import sys
    import os
    from oracle import ism_oracle as O
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(PKG, "ism"))
    U = importlib.import_module("model.utils")

    class Cfg:
        min_box_size = 0.05
        min_mask_size = 3e-4

    g = torch.Generator().manual_seed(5)
    N, H, W = 120, 96, 128
    boxes = _rand_boxes(g, N, 128).clamp(0, 95).long()
    boxes[:10, 2:] = boxes[:10, :2] + 2  # tiny boxes
    masks = torch.zeros(N, H, W)
    for i in range(N):
        x0, y0, x1, y1 = boxes[i].tolist()
        masks[i, y0:y1, x0:x1] = (torch.rand(max(y1 - y0, 0), max(x1 - x0, 0), generator=g) > 0.4).float()
    masks[10:14] = 0  # empty masks
    det = U.Detections({"masks": masks.to(dev), "boxes": boxes.float().to(dev)})
    assert det.boxes.dtype == torch.int64
    keep = O.small_detection_keep(boxes, masks, Cfg.min_box_size, Cfg.min_mask_size)
    assert 0 < int(keep.sum()) < N
    det.remove_very_small_detections(Cfg)
    assert torch.equal(det.boxes.cpu(), boxes[keep]) and torch.equal(det.masks.cpu(), masks[keep])
    b2, m2 = boxes[keep], masks[keep]
    n2 = len(b2)
    scores = torch.rand(n2, generator=g)
    oid = torch.randint(0, 4, (n2,), generator=g)
    det.add_attribute("scores", scores.to(dev))
    det.add_attribute("object_ids", oid.to(dev))
    det.check_size()
    kidx = O.nms_per_object_id(b2.float(), scores, oid, 0.25)
    det.apply_nms_per_object_id(nms_thresh=0.25)
    assert len(det) == len(kidx) < n2
    assert torch.equal(det.scores.cpu(), scores[kidx]) and torch.equal(det.object_ids.cpu(), oid[kidx])
    assert torch.equal(det.boxes.cpu(), b2[kidx]) and torch.equal(det.masks.cpu(), m2[kidx])
    c = det.clone()
    sel = torch.tensor([3, 0, -1, 3])
    c.filter(sel.to(dev))
    assert torch.equal(c.scores.cpu(), scores[kidx][sel]) and torch.equal(c.masks.cpu(), m2[kidx][sel])
    assert len(det) == len(kidx)  # the clone was filtered, not the original
    c.apply_nms(0.5)
    assert torch.equal(c.scores.cpu(), scores[kidx][sel][O.nms(b2[kidx][sel].float(), scores[kidx][sel], 0.5)])
    det.to_numpy()
    assert isinstance(det.masks, np.ndarray)


def test_take_rows_dtypes(dev):
    from sam6d_hip import ism
    g = torch.Generator().manual_seed(1)
    idx = torch.tensor([5, 0, 0, 7, -2])
    for t in (torch.rand(8, 3, 5, generator=g), torch.randint(0, 9, (8,), generator=g), torch.rand(8, 4, generator=g) > 0.5,
              torch.randint(0, 255, (8, 7), generator=g).to(torch.uint8), torch.rand(8, 16, generator=g).double()):
        assert torch.equal(ism.take_rows(t.to(dev), idx.to(dev)).cpu(), t[idx])
        m = torch.tensor([1, 0, 0, 1, 1, 0, 0, 1], dtype=torch.bool)
        assert torch.equal(ism.take_rows(t.to(dev), m.to(dev)).cpu(), t[m])
    assert ism.take_rows(torch.rand(4, 2).to(dev), torch.zeros(0, dtype=torch.int64).to(dev)).shape == (0, 2)


# ---------------------------------------------------------------------------- get_test_data geometry (row f2)
def test_proposal_geometry_golden_bit_exact(dev):
    """mask & depth -> crop box -> ordered masked pixels -> radius filter -> chosen points and resized-crop indices, against the
    values composed from the reference's own helpers (tests/golden/test_data.npz)."""
    import hashlib
    from tests._util import golden
    from sam6d_hip import pem
    g = golden("test_data")
    masks = torch.from_numpy(g["masks"]).to(dev)
    depth = torch.from_numpy(g["depth"]).to(dev)
    geom = pem.proposal_geometry(masks, depth, g["K"], float(g["radius"]))
    kept = [int(i) for i in g["kept"]]
    count = geom["count"].cpu().numpy(); n_keep = geom["n_keep"].cpu().numpy()
    ours = [i for i in range(masks.shape[0]) if count[i] > 32 and n_keep[i] >= 4]
    assert ours == kept, (ours, kept)
    ns = int(g["ns"])
    sel = torch.zeros(masks.shape[0], ns, dtype=torch.int32)
    for i in kept:
        sel[i] = torch.from_numpy(g["p%d_sel" % i].astype(np.int32))
    pts, rc = pem.proposal_choose(geom, sel.to(dev), int(g["img_size"]))
    for i in kept:
        assert geom["bbox"][i].cpu().tolist() == [int(v) for v in g["p%d_bbox" % i]]
        k = int(g["p%d_n_keep" % i])
        assert int(n_keep[i]) == k
        assert np.array_equal(geom["center"][i].cpu().numpy(), g["p%d_center" % i])
        ch = np.ascontiguousarray(geom["choose"][i, :k].cpu().numpy())
        cl = np.ascontiguousarray(geom["cloud"][i, :k].cpu().numpy())
        assert hashlib.sha256(ch.tobytes()).hexdigest() == str(g["p%d_choose_sha" % i])
        assert hashlib.sha256(cl.tobytes()).hexdigest() == str(g["p%d_cloud_sha" % i])
        assert np.array_equal(pts[i].cpu().numpy(), g["p%d_pts" % i])
        assert np.array_equal(rc[i].cpu().numpy(), g["p%d_rgb_choose" % i])


def test_proposal_geometry_random_masks_vs_oracle(dev):
    from oracle import pem_oracle as O
    from sam6d_hip import pem
    g = np.random.default_rng(3)
    H, Wd, N = 96, 128, 12
    yy, xx = np.mgrid[0:H, 0:Wd]
    depth = (0.5 + 0.004 * xx + 0.3 * g.random((H, Wd))).astype(np.float32)
    depth[g.random((H, Wd)) < 0.1] = 0
    masks = np.zeros((N, H, Wd), np.uint8)
    for i in range(N):
        cy, cx, r = g.integers(0, H), g.integers(0, Wd), g.integers(3, 40)
        masks[i] = (((xx - cx) ** 2 + (yy - cy) ** 2) < r * r) & (g.random((H, Wd)) > 0.2)
    K = np.array([[150.0, 0, 64.0], [0, 150.0, 48.0], [0, 0, 1]], np.float32)
    geom = pem.proposal_geometry(torch.from_numpy(masks).to(dev), torch.from_numpy(depth).to(dev), K, 0.11)
    for i in range(N):
        o = O.proposal_geometry(masks[i], depth, K, np.float32(0.11))
        cnt, nk = int(geom["count"][i]), int(geom["n_keep"][i])
        assert (o is None) == (cnt <= 32 or nk < 4)
        if o is None:
            continue
        assert geom["bbox"][i].cpu().tolist() == o["bbox"] and nk == len(o["choose"])
        assert np.array_equal(geom["choose"][i, :nk].cpu().numpy(), o["choose"].astype(np.int32))
        assert np.array_equal(geom["cloud"][i, :nk].cpu().numpy(), o["cloud"])


# ------------------------------------------------------------------------------------------ detection_ism.json
def _blob_masks(g, N, H, W):
    m = torch.zeros(N, H, W)
    for i in range(N):
        x0 = int(torch.randint(0, max(1, W - 8), (1,), generator=g)); y0 = int(torch.randint(0, max(1, H - 8), (1,), generator=g))
        w = int(torch.randint(1, max(2, W // 2), (1,), generator=g)); h = int(torch.randint(1, max(2, H // 2), (1,), generator=g))
        m[i, y0:y0 + h, x0:x0 + w] = torch.rand(min(h, H - y0), min(w, W - x0), generator=g)  # soft values in (0, 1)
        m[i][torch.rand(H, W, generator=g) < 0.02] = 0
    return m


def test_mask_rle_golden_bit_exact(dev):
    """RLE kernels vs the counts the reference's own amg.mask_to_rle_pytorch produced (tests/golden/rle.npz)."""
    from sam6d_hip import ism
    from tests._util import golden
    g = golden("rle")
    masks = torch.from_numpy(g["masks"]).to(dev)
    counts, offs = ism.mask_rle_encode(masks)
    assert np.array_equal(offs.cpu().numpy(), g["offsets"]) and np.array_equal(counts.cpu().numpy(), g["counts"])
    rles = ism.mask_to_rle(masks)
    assert rles[6] == {"counts": [0, 60 * 80], "size": [60, 80]} and rles[7]["counts"] == [0, 1, 60 * 80 - 1]
    back = ism.rle_to_mask(rles, dev)
    assert back.dtype == torch.uint8 and np.array_equal(back.cpu().numpy(), g["masks"])


@pytest.mark.parametrize("N,H,W", [(24, 480, 640), (3, 7, 300), (2, 1, 1), (5, 33, 1), (4, 1, 77), (1, 257, 513), (0, 8, 8)])
def test_mask_rle_vs_oracle_and_round_trip(dev, N, H, W):
    from oracle import ism_oracle as O
    from sam6d_hip import ism
    g = torch.Generator().manual_seed(N * 1000 + H + W)
    masks = _blob_masks(g, N, H, W)
    if N > 1:
        masks[0] = 0
        masks[1] = (torch.rand(H, W, generator=g) < 0.5).float()  # noise: the most runs a mask can have
    rles = ism.mask_to_rle(masks.to(dev))
    assert len(rles) == N
    for i in range(N):
        assert rles[i] == O.mask_to_rle(O.force_binary_mask(masks[i].numpy())), "mask %d" % i
        assert sum(rles[i]["counts"]) == H * W
    if N:
        back = ism.rle_to_mask(rles, dev).cpu().numpy()
        assert np.array_equal(back, (masks.numpy() > 0).astype(np.uint8))
        # counts that stop short of H*W decode to zeros beyond the encoded runs (amg.rle_to_mask would leave garbage there)
        short = [{"counts": r["counts"][:-1], "size": r["size"]} for r in rles]
        dec = ism.rle_to_mask(short, dev).cpu().numpy()
        for i in range(N):
            n_enc = sum(short[i]["counts"])
            ref = (masks[i].numpy() > 0).transpose().reshape(-1).copy()  # column-major order
            ref[n_enc:] = False
            assert np.array_equal(dec[i].astype(bool), ref.reshape(W, H).transpose())


def test_mask_rle_rejects_cpu_and_bad_offsets(dev):
    from sam6d_hip import _lib, ism
    with pytest.raises(RuntimeError):
        ism.mask_rle_encode(torch.zeros(1, 4, 4))
    with pytest.raises(RuntimeError):
        ism.rle_to_mask([{"counts": "ab\x01", "size": [2, 2]}], dev)  # a character outside the RLE alphabet
    m = torch.zeros(1, 4, 4, device=dev); m[0, 1, 1] = 1
    counts = torch.full((8,), -5, dtype=torch.int32, device=dev)
    offs = torch.tensor([0, 5], dtype=torch.int64, device=dev)  # the mask has 3 runs, not 5: nothing may be written
    _lib.call("sam6d_mask_rle_encode", m.data_ptr(), 1, 4, 4, offs.data_ptr(), counts.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int((counts != -5).sum()) == 0


def test_detection_ism_json_seam(dev, tmp_path):
    """ISM/run_inference_custom.py:260-264: to_numpy -> save_to_file -> convert_npz_to_json -> save_json_bop23, then the PEM side
    (run_inference_custom_pytorch.py:282-317): json.load -> per-instance RLE -> mask.  Records equal the oracle's; masks survive."""
    import importlib
    import json
    import os
    import sys
    from oracle import ism_oracle as O
    from sam6d_hip import ism
    for k in [k for k in sys.modules if k == "model" or k.startswith("model.") or k == "utils" or k.startswith("utils.")]:
        del sys.modules[k]
    sys.path.insert(0, os.path.join(PKG, "ism"))
    U = importlib.import_module("model.utils")
    g = torch.Generator().manual_seed(9)
    N, H, W = 17, 120, 160
    masks = (_blob_masks(g, N, H, W) > 0).float()
    boxes = torch.randint(0, 100, (N, 4), generator=g)
    scores = torch.rand(N, generator=g)
    oids = torch.randint(0, 5, (N,), generator=g)
    det = U.Detections({"masks": masks.to(dev), "boxes": boxes.to(dev)})
    det.add_attribute("scores", scores.to(dev))
    det.add_attribute("object_ids", oids.to(dev))
    det.to_numpy()
    path = str(tmp_path / "detection_ism")
    det.save_to_file(0, 0, 0, path, "Custom", return_results=False)
    recs = U.convert_npz_to_json(idx=0, list_npz_paths=[path + ".npz"])
    U.save_json_bop23(path + ".json", recs)
    want = O.detections_to_records(oids.numpy(), scores.numpy(), boxes.numpy(), masks.numpy())
    assert recs == want
    with open(path + ".json") as f:
        loaded = json.load(f)
    assert loaded == want
    back = ism.rle_to_mask([r["segmentation"] for r in loaded], dev)
    assert torch.equal(back.cpu(), masks.to(torch.uint8))
    assert U.mask_to_rle(masks[3].numpy()) == want[3]["segmentation"]
    again = U.Detections(path + ".npz")  # load_from_file: xywh -> xyxy, ids back to 0-based
    assert np.array_equal(again.object_ids, oids.numpy()) and np.array_equal(np.asarray(again.masks), masks.numpy())


def test_compressed_rle_strings_decode_like_their_counts(dev):
    """pycocotools' COMPRESSED counts strings (the form PEM/run_inference_custom_pytorch.py:312-317 passes to cocomask.decode when
    frPyObjects refuses the object): the string of a mask's runs decodes to the same mask as the uncompressed runs, for seeded masks,
    an empty and a full mask; encode -> decode round trip of the strings.  PARITY UNPINNED (pycocotools is not installed; the codec is
    restated from cocoapi's maskApi.c and pinned by hand-worked strings in tests/test_oracle_golden.py)."""
    from sam6d_hip import ism
    gen = torch.Generator().manual_seed(12)
    masks = (torch.rand(6, 40, 56, generator=gen) > 0.55).float()
    masks[1, 5:30, 10:50] = 1.0
    masks[2] = 0.0
    masks[3] = 1.0
    rles = ism.mask_to_rle(masks.to(dev))
    strs = [{"counts": ism.rle_counts_to_string(r["counts"]), "size": r["size"]} for r in rles]
    for r, q in zip(rles, strs):
        assert ism.rle_string_to_counts(q["counts"]) == r["counts"]
        assert ism.rle_string_to_counts(q["counts"].encode("ascii")) == r["counts"]
    a = ism.rle_to_mask(rles, dev)
    b = ism.rle_to_mask(strs, dev)
    c = ism.rle_to_mask([rles[0], strs[1], strs[2], rles[3], strs[4], rles[5]], dev)  # both forms in one file
    assert torch.equal(a, b) and torch.equal(a, c)
    assert np.array_equal(a.cpu().numpy().astype(bool), masks.numpy() > 0)
    with pytest.raises(RuntimeError):
        ism.rle_string_to_counts("T")  # continuation bit set on the last character
