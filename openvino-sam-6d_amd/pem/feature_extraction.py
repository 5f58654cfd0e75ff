"""Drop-in for PEM/model/feature_extraction.py.  The ViT backbone is OUTSIDE the matching hot path:
it stays plain PyTorch-ROCm.  timm is used when importable; otherwise a local ViT-B/16 with timm-compatible state_dict
keys (patch_embed.proj, cls_token, pos_embed, blocks.N.{norm1,attn.qkv,attn.proj,norm2,mlp.fc1,mlp.fc2}, norm) so the
released checkpoint still loads.  No network access: a missing MAE checkpoint is skipped with a message instead of
being downloaded (feature_extraction.py:78-97)."""
import os
from functools import partial

import torch
import torch.nn as nn
from torch.nn import functional as F

from model_utils import get_chosen_pixel_feats, sample_pts_feats
from sam6d_hip import pem as _pem


class _Attn(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, Cc = x.shape
        q, k, v = self.qkv(x).reshape(B, N, 3, self.num_heads, Cc // self.num_heads).permute(2, 0, 3, 1, 4)
        x = F.scaled_dot_product_attention(q, k, v)
        return self.proj(x.transpose(1, 2).reshape(B, N, Cc))


class _Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))


class _Block(nn.Module):
    def __init__(self, dim, heads, mlp_ratio, norm_layer):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attn(dim, heads)
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _PatchEmbed(nn.Module):
    def __init__(self, patch, dim):
        super().__init__()
        self.proj = nn.Conv2d(3, dim, patch, patch)
        self.num_patches = (224 // patch) ** 2

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class ViT(nn.Module):
    """feature_extraction.py:17-35: returns the normalised token maps of 4 evenly spaced blocks."""

    def __init__(self, patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, qkv_bias=True, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        self.patch_embed = _PatchEmbed(patch_size, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * .02)
        self.blocks = nn.ModuleList([_Block(embed_dim, num_heads, mlp_ratio, norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1) + self.pos_embed
        d = len(self.blocks)
        n = d // 4
        want = [d - 1, d - n - 1, d - 2 * n - 1, d - 3 * n - 1]
        out = []
        for i, blk in enumerate(self.blocks):
            x = blk(x)
            if i in want:
                out.append(self.norm(x))
        return out


class ViT_AE(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.embed_dim, self.out_dim = cfg.embed_dim, cfg.out_dim
        self.use_pyramid_feat = cfg.use_pyramid_feat
        if cfg.vit_type == 'vit_base':
            depth, heads = 12, 12
        elif cfg.vit_type == 'vit_large':
            depth, heads = 24, 16
        else:
            raise ValueError(cfg.vit_type)
        if cfg.up_type != 'linear':
            raise NotImplementedError("only up_type='linear' (PEM/config/base.yaml:20)")
        self.vit = ViT(patch_size=16, embed_dim=self.embed_dim, depth=depth, num_heads=heads, mlp_ratio=4)
        nblock = 4 if self.use_pyramid_feat else 1
        self.output_upscaling = nn.Linear(self.embed_dim * nblock, 16 * self.out_dim, bias=True)
        if getattr(cfg, "pretrained", False):
            ck = os.path.join('checkpoints', 'mae_pretrain_' + cfg.vit_type + '.pth')
            if os.path.isfile(ck):
                sd = torch.load(ck, map_location='cpu', weights_only=True)
                self.vit.load_state_dict(sd.get('model', sd), strict=False)
            else:
                print("[feature_extraction] %s not found and no network access: ViT left at its initial weights" % ck)

    def forward(self, x):
        B, _, Hh, Ww = x.size()
        outs = self.vit(x)
        cls_tokens = outs[-1][:, 0, :].contiguous()
        outs = [l[:, 1:, :].contiguous() for l in outs]
        x = torch.cat(outs, dim=2) if self.use_pyramid_feat else outs[-1]
        x = self.output_upscaling(x).reshape(B, 14, 14, 4, 4, self.out_dim).permute(0, 5, 1, 3, 2, 4).contiguous()
        x = F.interpolate(x.reshape(B, -1, 56, 56), (Hh, Ww), mode="bilinear", align_corners=False)
        return x, cls_tokens


class ViTEncoder(nn.Module):
    """feature_extraction.py:122-172."""

    def __init__(self, cfg, npoint=2048):
        super().__init__()
        self.npoint = npoint
        self.rgb_net = ViT_AE(cfg)

    def forward(self, pts, rgb, rgb_choose, dense_po, dense_fo):
        if dense_po is None or dense_fo is None:
            raise ValueError('dense_po and dense_fo must be provided for export/inference')
        dense_fm = self.get_img_feats(rgb, rgb_choose)
        dense_pm, dense_po, radius = _pem.radius_normalize(pts, dense_po)  # feature_extraction.py:133-137, on the GPU
        return dense_pm, dense_fm, dense_po, dense_fo, radius

    def get_img_feats(self, img, choose):
        return get_chosen_pixel_feats(self.rgb_net(img)[0], choose)

    def get_obj_feats(self, tem_rgb_list, tem_pts_list, tem_choose_list, npoint=None):
        npoint = npoint or self.npoint
        if isinstance(tem_rgb_list, list):
            feats = [self.get_img_feats(t, c) for t, c in zip(tem_rgb_list, tem_choose_list)]
            return sample_pts_feats(torch.cat(tem_pts_list, dim=1), torch.cat(feats, dim=1), npoint)
        B, T = tem_rgb_list.shape[:2]
        f = self.get_img_feats(tem_rgb_list.view(B * T, *tem_rgb_list.shape[2:]), tem_choose_list.view(B * T, -1))
        Fd = f.shape[-1]
        return sample_pts_feats(tem_pts_list.view(B, -1, 3), f.view(B, -1, Fd), npoint)
