"""Drop-in for PEM/model/coarse_point_matching.py (inference)."""
import torch
import torch.nn as nn

from transformer import GeometricTransformer, _Packed
from sam6d_hip import pem as _pem


def _cfg_dict(cfg, extra=None):
    d = dict(_pem.DEFAULT_CFG)
    for k in ("temp", "nproposal1", "nproposal2", "pe_radius1", "pe_radius2"):
        if hasattr(cfg, k):
            d[k] = getattr(cfg, k)
    d.update(extra or {})
    return d


class CoarsePointMatching(_Packed):
    """coarse_point_matching.py:12-63.  forward(p1, f1, geo1, p2, f2, geo2, radius, model) -> (init_R, init_t).
    `hypothesis_rand` (B, 3*nproposal1) may be set on the module to inject the sampling uniforms (parity tests); otherwise
    they are drawn with torch.rand like the reference (model_utils.py:292)."""

    def __init__(self, cfg, return_feat=False):
        super().__init__()
        self.cfg, self.return_feat, self.nblock = cfg, return_feat, cfg.nblock
        if cfg.sim_type != 'cosine' or not cfg.normalize_feat:
            raise ValueError("only sim_type='cosine', normalize_feat=True (PEM/config/base.yaml:39-40)")
        self.in_proj = nn.Linear(cfg.input_dim, cfg.hidden_dim)
        self.out_proj = nn.Linear(cfg.hidden_dim, cfg.out_dim)
        self.bg_token = nn.Parameter(torch.randn(1, 1, cfg.hidden_dim) * .02)
        self.transformers = nn.ModuleList([
            GeometricTransformer(blocks=['self', 'cross'], d_model=cfg.hidden_dim, num_heads=4, dropout=None,
                                 activation_fn='ReLU', return_attention_scores=False) for _ in range(self.nblock)])
        self.hypothesis_rand = None

    def _build(self, sd, dev):
        sd = {"coarse_point_matching." + k: v for k, v in sd.items()}
        W = _pem.PemWeights.__new__(_pem.PemWeights)
        g = _pem._getter(sd, dev)
        W.coarse = _pem.PemWeights._matching(g, "coarse_point_matching")
        W.coarse["blocks"] = [_pem.pack_geo_transformer(sd, dev, "coarse_point_matching.transformers.%d" % i)
                              for i in range(self.nblock)]
        return W

    def forward(self, p1, f1, geo1, p2, f2, geo2, radius, model):
        if self.training:
            raise RuntimeError("inference only: call .eval() (the reference fork is inference-only too, README.md:78-84)")
        W = self._packed(self._build)
        B = p1.shape[0]
        rand = self.hypothesis_rand
        if rand is None:
            rand = torch.rand(B, self.cfg.nproposal1 * 3, device=p1.device)
        sp = torch.cat([p1, p2], 0).contiguous()
        sf = torch.cat([f1, f2], 0).contiguous()
        E = torch.cat([geo1, geo2], 0).contiguous()
        return _pem.coarse_point_matching(sp, sf, E, radius.reshape(-1).contiguous(), model.contiguous(), W,
                                          rand.contiguous(), _cfg_dict(self.cfg))
