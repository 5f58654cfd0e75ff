"""Synthetic weights and inputs for the PEM/ISM matching path (no checkpoints exist offline).

`pem_param_shapes()` lists the reference's state_dict keys/shapes for the hot-path sub-modules
(verified against the reference modules by oracle/gen_golden.py), `make_pem_weights()` fills them
deterministically from a seed with non-trivial values for every tensor (biases, LayerNorm/BatchNorm affine and
running statistics included, so parity tests exercise them), and `config2_inputs()` is BASELINE.md's headline
workload generator.
"""
import math

import numpy as np
import torch

C = 256


def _attn_layer(p, rpe):
    out = []
    names = ["proj_q", "proj_k", "proj_v"] + (["proj_p"] if rpe else [])
    for n in names:
        out += [(f"{p}.attention.attention.{n}.weight", (C, C)), (f"{p}.attention.attention.{n}.bias", (C,))]
    out += [(f"{p}.attention.linear.weight", (C, C)), (f"{p}.attention.linear.bias", (C,)),
            (f"{p}.attention.norm.weight", (C,)), (f"{p}.attention.norm.bias", (C,))]
    out += _ffn(p)
    return out


def _ffn(p):
    return [(f"{p}.output.expand.weight", (2 * C, C)), (f"{p}.output.expand.bias", (2 * C,)),
            (f"{p}.output.squeeze.weight", (C, 2 * C)), (f"{p}.output.squeeze.bias", (C,)),
            (f"{p}.output.norm.weight", (C,)), (f"{p}.output.norm.bias", (C,))]


def _geo_transformer(p):
    return _attn_layer(p + ".layers.0", True) + _attn_layer(p + ".layers.1", False)


def pem_param_shapes(nblock=3):
    """(key, shape) for geo_embedding / coarse_point_matching / fine_point_matching, reference naming."""
    s = [("geo_embedding.embedding.div_term", (C // 2,)),
         ("geo_embedding.proj_d.weight", (C, C)), ("geo_embedding.proj_d.bias", (C,)),
         ("geo_embedding.proj_a.weight", (C, C)), ("geo_embedding.proj_a.bias", (C,))]
    for m in ("coarse_point_matching", "fine_point_matching"):
        s += [(f"{m}.bg_token", (1, 1, C)), (f"{m}.in_proj.weight", (C, C)), (f"{m}.in_proj.bias", (C,)),
              (f"{m}.out_proj.weight", (C, C)), (f"{m}.out_proj.bias", (C,))]
    for i in range(nblock):
        s += _geo_transformer(f"coarse_point_matching.transformers.{i}")
    for k in (1, 2):
        dims = [6, 32, 64, 128]
        for l in range(3):
            q = f"fine_point_matching.PE.mlp{k}.layer{l}"
            s += [(q + ".conv.weight", (dims[l + 1], dims[l], 1, 1))]
            for n in ("weight", "bias", "running_mean", "running_var"):
                s += [(f"{q}.normlayer.bn.{n}", (dims[l + 1],))]
            s += [(f"{q}.normlayer.bn.num_batches_tracked", ())]
    s += [("fine_point_matching.PE.mlp3.conv.weight", (C, C, 1)), ("fine_point_matching.PE.mlp3.conv.bias", (C,))]
    for i in range(nblock):
        t = f"fine_point_matching.transformers.{i}"
        s += _geo_transformer(t + ".sparse_layer")
        d = t + ".dense_layer"
        s += [(f"{d}.attention.attention.scale", (1, 1, C))]
        for n in ("proj_q", "proj_k", "proj_v"):
            s += [(f"{d}.attention.attention.{n}.weight", (C, C)), (f"{d}.attention.attention.{n}.bias", (C,))]
        s += [(f"{d}.attention.linear.weight", (C, C)), (f"{d}.attention.linear.bias", (C,)),
              (f"{d}.attention.norm.weight", (C,)), (f"{d}.attention.norm.bias", (C,))]
        s += _ffn(d)
    return s


def div_term(d_model=C):
    """PEM/model/transformer.py:264-266 (same torch ops, so the buffer is bit-identical)."""
    return torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model))


def make_pem_weights(seed=1, nblock=3):
    """Deterministic CPU float32 weights for every key of pem_param_shapes()."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in pem_param_shapes(nblock):
        leaf = key.rsplit(".", 1)[-1]
        if key.endswith("div_term"):
            t = div_term()
        elif leaf == "num_batches_tracked":
            t = torch.tensor(1, dtype=torch.int64)
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "bg_token":
            t = 0.02 * torch.randn(shape, generator=g)
        elif leaf == "scale":
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == "weight" and len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        elif leaf == "weight":  # LayerNorm / BatchNorm gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:  # biases, norm betas
            t = 0.05 * torch.randn(shape, generator=g)
        sd[key] = t.float() if t.dtype != torch.int64 else t
    return sd


def config2_inputs(B=32, seed=1, n_dense=2048, n_model=1024):
    """BASELINE.md / SURVEY 8d config 2: synthetic (B,2048,3) clouds; CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g) - 0.5
    dense_pm = u(B, n_dense, 3) + torch.tensor([0.0, 0.0, 8.0])
    dense_po = u(B, n_dense, 3)
    dense_fm = torch.randn(B, n_dense, C, generator=g)
    dense_fo = torch.randn(B, n_dense, C, generator=g)
    model = u(B, n_model, 3)
    radius = torch.ones(B)
    rand = torch.rand(B, 18000, generator=g)
    return dict(dense_pm=dense_pm, dense_fm=dense_fm, dense_po=dense_po, dense_fo=dense_fo, radius=radius,
                model=model, rand=rand)


def config4_inputs(B=200, n_obj=8, seed=4, n_dense=2048, n_model=1024):
    """an LM-O-style scene -- B proposals of n_obj objects (object of proposal b: b % n_obj, i.e. ~B / n_obj each),
    from the config-2 generator.  Template-side tensors are returned UNIQUE, one per object -- dense_po (n_obj,N,3), dense_fo
    (n_obj,N,256), model_obj (n_obj,P,3) -- with template_ids (B,) int64; `model` (B,P,3) is the per-proposal CAD sample the reference's
    caller passes (PEM/run_inference_custom_pytorch.py:447-454).  repeated(d) gives the per-instance repeated form."""
    g = torch.Generator().manual_seed(seed)
    u = lambda *s: torch.rand(*s, generator=g) - 0.5
    dense_pm = u(B, n_dense, 3) + torch.tensor([0.0, 0.0, 8.0])
    dense_fm = torch.randn(B, n_dense, C, generator=g)
    dense_po = u(n_obj, n_dense, 3)
    dense_fo = torch.randn(n_obj, n_dense, C, generator=g)
    model_obj = u(n_obj, n_model, 3)
    ids = torch.arange(B, dtype=torch.int64) % n_obj
    rand = torch.rand(B, 18000, generator=g)
    return dict(dense_pm=dense_pm, dense_fm=dense_fm, dense_po=dense_po, dense_fo=dense_fo, radius=torch.ones(B), model=model_obj[ids].contiguous(),
                rand=rand, template_ids=ids)


def repeated(d):
    """The per-instance repeated form of a config4_inputs dict (what the reference's caller builds with .repeat / indexing)."""
    ids = d["template_ids"].to(d["dense_po"].device)
    out = {k: v for k, v in d.items() if k != "template_ids"}
    out["dense_po"] = d["dense_po"][ids].contiguous()
    out["dense_fo"] = d["dense_fo"][ids].contiguous()
    return out


def config3_inputs(seed=0, Nq=200, Nt=42, D=1024, Pn=256, H=480, W=640):
    """SURVEY 8d config 3 (ISM template scoring): 200 proposals x 42 templates, 1024-d class tokens, 256 x 1024 patch descriptors with
    30 % of the patches masked out, box masks on a 480 x 640 depth image, 2048 CAD points, 42 template poses; CPU tensors.  Descriptors
    of 150 proposals are correlated with the templates so that the 0.2 confidence threshold selects about three quarters of them."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(Nq, D, generator=g)
    base = torch.randn(D, generator=g)
    ref = base + 0.8 * torch.randn(1, Nt, D, generator=g)
    q[:150] = base + 0.5 * ref[0, torch.randint(0, Nt, (150,), generator=g)] + 0.9 * q[:150]
    q_appe = torch.nn.functional.normalize(torch.randn(Nq, Pn, D, generator=g), dim=-1)
    r_appe = torch.nn.functional.normalize(torch.randn(1, Nt, Pn, D, generator=g), dim=-1)
    q_appe = q_appe * (torch.rand(Nq, Pn, 1, generator=g) > 0.3)
    r_appe = r_appe * (torch.rand(1, Nt, Pn, 1, generator=g) > 0.3)
    poses = torch.eye(4).repeat(Nt, 1, 1)
    for i in range(Nt):
        poses[i, :3, :3] = random_rotation(g)
    poses[:, :3, 3] = torch.randn(Nt, 3, generator=g) * 0.4
    pc = (torch.rand(1, 2048, 3, generator=g) - 0.5) * 0.2
    K = torch.tensor([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=torch.float64)
    depth = (800 + 200 * torch.rand(H, W, generator=g)).to(torch.int32)
    depth[torch.rand(H, W, generator=g) < 0.1] = 0
    x0 = torch.randint(0, W - 120, (Nq,), generator=g); y0 = torch.randint(0, H - 120, (Nq,), generator=g)
    bw = torch.randint(40, 120, (Nq,), generator=g); bh = torch.randint(40, 120, (Nq,), generator=g)
    yy = torch.arange(H).view(1, H, 1); xx = torch.arange(W).view(1, 1, W)
    masks = ((yy >= y0.view(-1, 1, 1)) & (yy < (y0 + bh).view(-1, 1, 1)) & (xx >= x0.view(-1, 1, 1)) & (xx < (x0 + bw).view(-1, 1, 1))).float()
    boxes = torch.stack([x0, y0, x0 + bw, y0 + bh], 1)
    return dict(q=q, ref=ref, q_appe=q_appe, r_appe=r_appe, poses=poses, pc=pc, K=K, depth_scale=torch.tensor(1.0, dtype=torch.float64),
                depth=depth, masks=masks, boxes=boxes)


def random_rotation(g):
    q = torch.randn(4, generator=g)
    q = q / q.norm()
    w, x, y, z = q.tolist()
    return torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=torch.float32)


def kat_inputs(B=2, seed=3, n_dense=2048, n_model=1024, noise=0.0):
    """Known-answer scene (SURVEY 8c KAT, extended to the full path): the observed cloud is the template
    cloud under a known rigid motion (p_obs = p_tmpl @ R_gt^T + t_gt, permuted), and corresponding points carry
    the same feature vector, so the matching is peaky and the discrete hypothesis selection is stable."""
    g = torch.Generator().manual_seed(seed)
    po = torch.rand(B, n_dense, 3, generator=g) - 0.5
    fo = torch.randn(B, n_dense, C, generator=g)
    Rg = torch.stack([random_rotation(g) for _ in range(B)])
    tg = torch.stack([torch.tensor([0.1, -0.2, 2.0]) + 0.1 * torch.randn(3, generator=g) for _ in range(B)])
    perm = torch.stack([torch.randperm(n_dense, generator=g) for _ in range(B)])
    pm = torch.gather(po, 1, perm.unsqueeze(2).expand(B, n_dense, 3)) @ Rg.transpose(1, 2) + tg.unsqueeze(1)
    fm = torch.gather(fo, 1, perm.unsqueeze(2).expand(B, n_dense, C))
    if noise > 0:
        pm = pm + noise * torch.randn(pm.shape, generator=g)
    sub = torch.stack([torch.randperm(n_dense, generator=g)[:n_model] for _ in range(B)])
    model = torch.gather(po, 1, sub.unsqueeze(2).expand(B, n_model, 3)).contiguous()
    rand = torch.rand(B, 18000, generator=g)
    return dict(dense_pm=pm.contiguous(), dense_fm=fm.contiguous(), dense_po=po, dense_fo=fo,
                radius=torch.ones(B), model=model, rand=rand, R_gt=Rg, t_gt=tg)


class AttrDict(dict):
    """attribute-style config object like the one PEM/run_inference_custom_pytorch.py:117-125 builds from the yaml"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


def default_model_cfg():
    """The `model:` section of PEM/config/base.yaml:16-54 (values restated here; the yaml itself is not shipped)."""
    A = AttrDict
    return A(coarse_npoint=196, fine_npoint=2048,
             feature_extraction=A(vit_type="vit_base", up_type="linear", embed_dim=768, out_dim=256,
                                  use_pyramid_feat=True, pretrained=False),
             geo_embedding=A(sigma_d=0.2, sigma_a=15, angle_k=3, reduction_a="max", hidden_dim=256),
             coarse_point_matching=A(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, temp=0.1, sim_type="cosine",
                                     normalize_feat=True, loss_dis_thres=0.15, nproposal1=6000, nproposal2=300),
             fine_point_matching=A(nblock=3, input_dim=256, hidden_dim=256, out_dim=256, pe_radius1=0.1, pe_radius2=0.2,
                                   focusing_factor=3, temp=0.1, sim_type="cosine", normalize_feat=True,
                                   loss_dis_thres=0.15))
