"""N>1 path on CPU: world_size-2 gloo processes shard proposals round-robin, all-gather the 13-float pose blocks and
reassemble them in global order.  The per-rank 'compute' here is a deterministic stand-in; the GPU kernels
are covered by the -m gpu tests."""
import os
import sys

import torch
import torch.multiprocessing as mp

from tests._util import ROOT, PKG  # noqa: F401


def _fake_pose(ids):
    g = ids.float().unsqueeze(1)
    R = (torch.arange(9).float().unsqueeze(0) + 100 * g).reshape(-1, 3, 3)
    t = torch.arange(3).float().unsqueeze(0) - g
    s = g.squeeze(1) * 0.5
    return R, t, s


def _worker(rank, world, port, n_total, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from sam6d_hip import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids, n_valid = parallel.shard_indices(n_total, rank, world)
    R, t, s = _fake_pose(ids)
    gR, gt, gs = parallel.gather_poses(R, t, s, dist)
    allR = parallel.unshard(gR.reshape(-1, 9), n_total, world).reshape(-1, 3, 3)
    allt = parallel.unshard(gt, n_total, world)
    alls = parallel.unshard(gs.reshape(-1, 1), n_total, world).reshape(-1)
    wR, wt, ws = _fake_pose(torch.arange(n_total))
    ok = torch.equal(allR, wR) and torch.equal(allt, wt) and torch.equal(alls, ws)
    ret[rank] = bool(ok) and n_valid == len(range(rank, n_total, world))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, n_total, port):
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_total, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_gather_poses_world2_even():
    _run(2, 8, 29611)


def test_gather_poses_world2_ragged():
    _run(2, 7, 29612)  # 7 proposals over 2 ranks: padding on rank 1


def test_shard_indices_cover():
    from sam6d_hip import parallel
    for n, w in ((200, 8), (7, 2), (1, 4), (32, 1)):
        seen = []
        for r in range(w):
            ids, nv = parallel.shard_indices(n, r, w)
            assert len(ids) == (n + w - 1) // w
            seen += ids[:nv].tolist()
        assert sorted(seen) == list(range(n))


# ------------------------------------------------------------------------------------------- ISM proposals over the ranks
def _ism_scene(quirk):
    """200 proposals of the config-3 generator with one mask / box per proposal; quirk: one proposal's box is moved away from its
    projected template so that its overlap is non-positive (the image-wide scalar-0.0 IoU case, ISM/utils/bbox_utils.py:214-220)."""
    from tests.test_oracle_golden import ism_inputs, ism_masks
    d = ism_inputs(0)
    masks, _ = ism_masks(d["gen"], 200)
    boxes = torch.tensor([0, 0, 640, 480]).repeat(200, 1)  # proposal boxes that certainly overlap the projected template: the whole image
    if quirk:
        boxes[7] = torch.tensor([0, 0, 1, 1])  # (a selected proposal of rank 1's shard)
    return d, masks, boxes


def _ism_score_fn(d, masks, boxes):
    from oracle import ism_oracle as IO

    def score_fn(ids):
        sel, obj, sem, best = IO.semantic_score(d["q"][ids], d["ref"])
        qa = d["q_appe"][ids][sel]
        appe, ref_sel = IO.appearance_score(best, obj, qa, d["r_appe"])
        bx = boxes[ids][sel]
        vu = IO.project_template_to_image(best, obj, d["poses"], d["pc"], masks[ids][sel], d["depth"], d["K"], d["depth_scale"])
        iou, vis = IO.geometric_score(vu, bx, qa, ref_sel)
        return dict(sel=sel, sem=sem, appe=appe, iou=iou, vis=vis, all_positive=torch.is_tensor(iou), object_ids=obj, boxes=bx)
    return score_fn


def _ism_final(sem, appe, geo, vis):
    from oracle import ism_oracle as IO
    return IO.final_score(sem, appe, geo if geo is not None else 0.0, vis)


def _ism_nms(boxes, scores, thresh, object_ids):
    from oracle import ism_oracle as IO
    return IO.nms_per_object_id(boxes, scores, object_ids, thresh)


def _ism_worker(rank, world, port, quirk, ret):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    from sam6d_hip import parallel
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d, masks, boxes = _ism_scene(quirk)
    out = parallel.ism_sharded_detections(_ism_score_fn(d, masks, boxes), 200, _ism_final, _ism_nms, dist, nms_thresh=0.25)
    ret[rank] = {k: v.clone() for k, v in out.items()}
    dist.barrier()
    dist.destroy_process_group()


def _ism_unsharded(quirk):
    """What the unsharded Detections flow holds after apply_nms_per_object_id (ISM/model/detector.py:355-390)."""
    from oracle import ism_oracle as IO
    d, masks, boxes = _ism_scene(quirk)
    r = _ism_score_fn(d, masks, boxes)(torch.arange(200))
    fin = IO.final_score(r["sem"], r["appe"], r["iou"], r["vis"])
    keep = IO.nms_per_object_id(r["boxes"].float(), fin, r["object_ids"], 0.25)
    return dict(scores=fin[keep], object_ids=r["object_ids"][keep], boxes=r["boxes"][keep], proposal_ids=r["sel"][keep],
                all_positive=r["all_positive"], n_selected=len(r["sel"]))


def _check_ism(quirk, port):
    want = _ism_unsharded(quirk)
    assert want["all_positive"] == (not quirk), "the scene does not exercise the intended IoU case"
    assert 20 < want["n_selected"] < 200 and 0 < len(want["proposal_ids"]) < want["n_selected"], "selection / NMS must both drop something"
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_ism_worker, args=(2, port, quirk, ret), nprocs=2, join=True)
    for r in range(2):
        got = ret[r]
        assert torch.equal(got["proposal_ids"], want["proposal_ids"]), "rank %d: kept proposals differ from the unsharded run" % r
        assert torch.equal(got["object_ids"], want["object_ids"]) and torch.equal(got["boxes"], want["boxes"])
        assert torch.equal(got["scores"], want["scores"]), "rank %d: final scores differ" % r


def test_ism_proposals_sharded_world2_equals_unsharded():
    """World-2 gloo run of the ISM scoring (per-rank compute = the CPU oracle): one all-gather of 12 floats per proposal, final score
    and per-object NMS on the merged records -- the same detections as the unsharded flow (ids, boxes, scores bit for bit)."""
    _check_ism(False, 29621)


def test_ism_proposals_sharded_world2_iou_quirk_is_image_wide():
    """A single proposal with a non-positive overlap zeroes the geometric term of EVERY proposal of the image
    (ISM/utils/bbox_utils.py:214-220) -- also of the proposals scored on the other rank."""
    _check_ism(True, 29622)
