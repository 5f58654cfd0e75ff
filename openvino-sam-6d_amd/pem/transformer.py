"""Drop-in for PEM/model/transformer.py: same class names, constructor arguments, forward signatures and state_dict
keys.  The modules are parameter containers; forward() packs the weights once (cached until the
parameters change) and issues the HIP launches of sam6d_hip.pem.  Inference only.

The three modules Net uses (GeometricStructureEmbedding, GeometricTransformer, SparseToDenseTransformer) run the fused kernels.  The
sub-modules (SinusoidalPositionalEmbedding, MultiHeadAttention, AttentionLayer, AttentionOutput, TransformerLayer, RPE*, Linear*) have
forwards too, for callers that use a layer directly: those are launch-per-op sequences of the same library (projection GEMMs, a
softmax kernel, batched GEMMs) that DO return the attention probabilities, as the reference's modules do."""
import numpy as np
import torch
import torch.nn as nn

from sam6d_hip import pem as _pem


def _sig(mod):
    return tuple((t.data_ptr(), t._version, t.device) for t in list(mod.parameters()) + list(mod.buffers()))


class _Packed(nn.Module):
    """caches a packed-weights object built from state_dict(); rebuilt when any tensor is replaced or modified"""

    def _packed(self, build):
        sig = _sig(self)
        if getattr(self, "_pack_sig", None) != sig:
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("%s: parameters must live on a HIP device (model.to('cuda')); there is no CPU path"
                                   % type(self).__name__)
            object.__setattr__(self, "_pack", build({k: v for k, v in self.state_dict().items()}, dev))
            object.__setattr__(self, "_pack_sig", sig)
        return self._pack


def _lin(m):
    """nn.Linear -> the library's weight record (pure views: nothing is copied for contiguous fp32 parameters)"""
    return _pem.Linear(m.weight.detach().float(), m.bias.detach().float() if m.bias is not None else None)


def _norm(m):
    return (m.weight.detach().float().contiguous(), m.bias.detach().float().contiguous())


def _no_masks(**kw):
    for k, v in kw.items():
        if v is not None:
            raise NotImplementedError("%s is not used by PEM inference and is not implemented" % k)


def _need_cuda(t, what):
    if not (torch.is_tensor(t) and t.is_cuda):
        raise RuntimeError("%s: needs HIP device tensors (this build has no CPU path)" % what)


class SinusoidalPositionalEmbedding(nn.Module):
    """transformer.py:259-285.  forward(emb_indices (*)) -> (*, d_model): [sin(x w_0), cos(x w_0), sin(x w_1), ...]."""

    def __init__(self, d_model):
        super().__init__()
        if d_model % 2 != 0:
            raise ValueError(f'Sinusoidal positional encoding with odd d_model: {d_model}')
        self.d_model = d_model
        self.register_buffer('div_term', torch.exp(torch.arange(0, d_model, 2).float() * (-np.log(10000.0) / d_model)))

    def forward(self, emb_indices):
        _need_cuda(emb_indices, "SinusoidalPositionalEmbedding")
        x = emb_indices.detach().float().contiguous()
        return _pem.sinusoid_embedding(x.reshape(-1), self.div_term, self.d_model).reshape(*emb_indices.shape, self.d_model)


class GeometricStructureEmbedding(_Packed):
    """transformer.py:288-363.  forward(points (B,N,3)) -> (B,N,N,hidden_dim)."""

    def __init__(self, cfg):
        super().__init__()
        self.sigma_d, self.sigma_a, self.angle_k = cfg.sigma_d, cfg.sigma_a, cfg.angle_k
        self.embedding = SinusoidalPositionalEmbedding(cfg.hidden_dim)
        self.proj_d = nn.Linear(cfg.hidden_dim, cfg.hidden_dim)
        self.proj_a = nn.Linear(cfg.hidden_dim, cfg.hidden_dim)
        self.reduction_a = cfg.reduction_a
        if self.reduction_a != 'max':
            raise ValueError(f'Unsupported reduction mode: {self.reduction_a} (only "max", PEM/config/base.yaml:30).')

    def forward(self, points):
        W = self._packed(lambda sd, dev: _pem.pack_geo({"g." + k: v for k, v in sd.items()}, dev, "g"))
        return _pem.geo_embedding(points.contiguous(), W, self.sigma_d, self.sigma_a, self.angle_k)


class _MHA(nn.Module):
    def __init__(self, d_model, num_heads, rpe):
        super().__init__()
        if d_model % num_heads != 0:
            raise ValueError('`d_model` ({}) must be a multiple of `num_heads` ({}).'.format(d_model, num_heads))
        self.d_model, self.num_heads = d_model, num_heads
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)
        self.proj_v = nn.Linear(d_model, d_model)
        if rpe:
            self.proj_p = nn.Linear(d_model, d_model)


class MultiHeadAttention(_MHA):
    """transformer.py:95-150.  forward(input_q (B,N,C), input_k (B,M,C), input_v (B,M,C)) -> hidden_states (B,N,C),
    attention_scores (B,H,N,M) (the softmax probabilities, as the reference returns them)."""

    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__(d_model, num_heads, False)

    def forward(self, input_q, input_k, input_v, key_weights=None, key_masks=None, attention_factors=None, attention_masks=None):
        _no_masks(key_weights=key_weights, key_masks=key_masks, attention_factors=attention_factors, attention_masks=attention_masks)
        _need_cuda(input_q, "MultiHeadAttention")
        return _pem.mha_forward(input_q.contiguous(), input_k.contiguous(), input_v.contiguous(), _lin(self.proj_q), _lin(self.proj_k),
                                _lin(self.proj_v), self.num_heads)


class RPEMultiHeadAttention(_MHA):
    """transformer.py:366-420.  forward(input_q, input_k, input_v, embed_qk (B,N,M,C)) -> hidden_states, attention_scores."""

    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__(d_model, num_heads, True)

    def forward(self, input_q, input_k, input_v, embed_qk, key_weights=None, key_masks=None, attention_factors=None):
        _no_masks(key_weights=key_weights, key_masks=key_masks, attention_factors=attention_factors)
        _need_cuda(input_q, "RPEMultiHeadAttention")
        return _pem.mha_forward(input_q.contiguous(), input_k.contiguous(), input_v.contiguous(), _lin(self.proj_q), _lin(self.proj_k),
                                _lin(self.proj_v), self.num_heads, embed_qk=embed_qk.contiguous(), proj_p=_lin(self.proj_p))


class _AttnLayer(nn.Module):
    def __init__(self, d_model, num_heads, rpe):
        super().__init__()
        self.attention = (RPEMultiHeadAttention if rpe else MultiHeadAttention)(d_model, num_heads)
        self.linear = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)


def _add_norm(hidden, lin, residual, norm):
    """LayerNorm(linear(hidden) + residual)   (transformer.py:158-160, 442-444, 608-610)"""
    shp = residual.shape
    h2 = hidden.reshape(-1, shp[-1]).contiguous()
    return _pem.layernorm(_pem.linear(h2, _lin(lin), residual=residual.reshape(-1, shp[-1]).contiguous()), _norm(norm)).reshape(shp)


class AttentionLayer(_AttnLayer):
    """transformer.py:152-181.  forward(input_states, memory_states) -> output_states, attention_scores."""

    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__(d_model, num_heads, False)

    def forward(self, input_states, memory_states, memory_weights=None, memory_masks=None, attention_factors=None, attention_masks=None):
        hidden, scores = self.attention(input_states, memory_states, memory_states, key_weights=memory_weights, key_masks=memory_masks,
                                        attention_factors=attention_factors, attention_masks=attention_masks)
        return _add_norm(hidden, self.linear, input_states, self.norm), scores


class RPEAttentionLayer(_AttnLayer):
    """transformer.py:423-458.  forward(input_states, memory_states, position_states) -> output_states, attention_scores."""

    def __init__(self, d_model, num_heads, dropout=None):
        super().__init__(d_model, num_heads, True)

    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None, attention_factors=None):
        hidden, scores = self.attention(input_states, memory_states, memory_states, position_states, key_weights=memory_weights,
                                        key_masks=memory_masks, attention_factors=attention_factors)
        return _add_norm(hidden, self.linear, input_states, self.norm), scores


class AttentionOutput(nn.Module):
    def __init__(self, d_model, dropout=None, activation_fn='ReLU'):
        super().__init__()
        if activation_fn != 'ReLU':
            raise ValueError("only ReLU is implemented (the PEM configuration)")
        self.expand = nn.Linear(d_model, d_model * 2)
        self.squeeze = nn.Linear(d_model * 2, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states):
        """transformer.py:193-199: LayerNorm(input + squeeze(relu(expand(input))))."""
        _need_cuda(input_states, "AttentionOutput")
        shp = input_states.shape
        x2 = input_states.reshape(-1, shp[-1]).contiguous()
        h = _pem.linear(x2, _lin(self.expand), act=1)
        return _pem.layernorm(_pem.linear(h, _lin(self.squeeze), residual=x2), _norm(self.norm)).reshape(shp)


class TransformerLayer(nn.Module):
    """transformer.py:202-226.  forward(input_states, memory_states) -> output_states, attention_scores."""

    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU'):
        super().__init__()
        self.attention = AttentionLayer(d_model, num_heads)
        self.output = AttentionOutput(d_model, activation_fn=activation_fn)

    def forward(self, input_states, memory_states, memory_weights=None, memory_masks=None, attention_factors=None, attention_masks=None):
        hidden, scores = self.attention(input_states, memory_states, memory_weights=memory_weights, memory_masks=memory_masks,
                                        attention_factors=attention_factors, attention_masks=attention_masks)
        return self.output(hidden), scores


class RPETransformerLayer(nn.Module):
    """transformer.py:461-479.  forward(input_states, memory_states, position_states) -> output_states, attention_scores."""

    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU'):
        super().__init__()
        self.attention = RPEAttentionLayer(d_model, num_heads)
        self.output = AttentionOutput(d_model, activation_fn=activation_fn)

    def forward(self, input_states, memory_states, position_states, memory_weights=None, memory_masks=None, attention_factors=None):
        hidden, scores = self.attention(input_states, memory_states, position_states, memory_weights=memory_weights,
                                        memory_masks=memory_masks, attention_factors=attention_factors)
        return self.output(hidden), scores


def _check(d_model, num_heads, blocks, dropout, parallel):
    if d_model != 256 or num_heads != 4:
        raise ValueError("kernels are specialised for d_model=256, num_heads=4 (PEM/config/base.yaml)")
    if list(blocks) != ['self', 'cross'] or parallel:
        raise ValueError("only blocks=['self','cross'] with sequential cross attention is implemented")
    if dropout:
        raise ValueError("inference only: dropout must be None")


class GeometricTransformer(_Packed):
    """transformer.py:483-527.  forward(feats0, embeddings0, feats1, embeddings1) -> (feats0, feats1)."""

    def __init__(self, blocks, d_model, num_heads, dropout=None, activation_fn='ReLU', return_attention_scores=False,
                 parallel=False):
        super().__init__()
        _check(d_model, num_heads, blocks, dropout, parallel)
        if return_attention_scores:
            raise ValueError("return_attention_scores is not supported (attention probabilities never leave the kernel)")
        self.blocks = blocks
        self.layers = nn.ModuleList([RPETransformerLayer(d_model, num_heads, activation_fn=activation_fn),
                                     TransformerLayer(d_model, num_heads, activation_fn=activation_fn)])

    def forward(self, feats0, embeddings0, feats1, embeddings1, masks0=None, masks1=None):
        if masks0 is not None or masks1 is not None:
            raise NotImplementedError("key masks are not used by PEM inference")
        T = self._packed(lambda sd, dev: _pem.pack_geo_transformer({"t." + k: v for k, v in sd.items()}, dev, "t"))
        B = feats0.shape[0]
        S = torch.cat([feats0, feats1], 0).contiguous()
        E = torch.cat([embeddings0, embeddings1], 0).contiguous()
        out = _pem.geometric_transformer(S, E, T)
        return out[:B], out[B:]


class LinearAttention(nn.Module):
    def __init__(self, d_model, num_heads, focusing_factor=3):
        super().__init__()
        if focusing_factor != 3:
            raise ValueError("focusing_factor must be 3 (PEM/config/base.yaml:49)")
        self.proj_q = nn.Linear(d_model, d_model)
        self.proj_k = nn.Linear(d_model, d_model)
        self.proj_v = nn.Linear(d_model, d_model)
        self.scale = nn.Parameter(torch.zeros(size=(1, 1, d_model)))
        self.num_heads = num_heads

    def forward(self, input_q, input_k, input_v):
        """transformer.py:548-578 (focused linear attention, the kv contraction order)."""
        _need_cuda(input_q, "LinearAttention")
        return _pem.linear_attention_forward(input_q.contiguous(), input_k.contiguous(), input_v.contiguous(), _lin(self.proj_q),
                                             _lin(self.proj_k), _lin(self.proj_v), self.scale.detach().float().reshape(-1).contiguous(),
                                             self.num_heads)


class LinearAttentionLayer(nn.Module):
    """transformer.py:581-609.  forward(input_states, memory_states) -> output_states."""

    def __init__(self, d_model, num_heads, dropout=False, focusing_factor=3):
        super().__init__()
        self.attention = LinearAttention(d_model, num_heads, focusing_factor=focusing_factor)
        self.linear = nn.Linear(d_model, d_model)
        self.norm = nn.LayerNorm(d_model)

    def forward(self, input_states, memory_states):
        hidden = self.attention(input_states, memory_states, memory_states)
        return _add_norm(hidden, self.linear, input_states, self.norm)


class LinearTransformerLayer(nn.Module):
    """transformer.py:612-622.  forward(input_states, memory_states) -> output_states."""

    def __init__(self, d_model, num_heads, dropout=None, activation_fn='ReLU', focusing_factor=3):
        super().__init__()
        self.attention = LinearAttentionLayer(d_model, num_heads, focusing_factor=focusing_factor)
        self.output = AttentionOutput(d_model, activation_fn=activation_fn)

    def forward(self, input_states, memory_states):
        return self.output(self.attention(input_states, memory_states))


class SparseToDenseTransformer(_Packed):
    """transformer.py:627-720.  forward(dense_feats0, embeddings0, fps_idx0, dense_feats1, embeddings1, fps_idx1)."""

    def __init__(self, d_model, sparse_blocks, num_heads=4, dropout=None, activation_fn='ReLU', parallel=False,
                 focusing_factor=3, with_bg_token=True, replace_bg_token=True):
        super().__init__()
        _check(d_model, num_heads, sparse_blocks, dropout, parallel)
        if not (with_bg_token and replace_bg_token):
            raise ValueError("only with_bg_token=True, replace_bg_token=True is implemented (fine_point_matching.py:37-38)")
        self.sparse_layer = GeometricTransformer(sparse_blocks, d_model, num_heads, activation_fn=activation_fn)
        self.dense_layer = LinearTransformerLayer(d_model, num_heads, focusing_factor=focusing_factor)

    def forward(self, dense_feats0, embeddings0, fps_idx0, dense_feats1, embeddings1, fps_idx1, masks0=None, masks1=None):
        T = self._packed(lambda sd, dev: _pem.pack_sparse_to_dense({"t." + k: v for k, v in sd.items()}, dev, "t"))
        B = dense_feats0.shape[0]
        D = torch.cat([dense_feats0, dense_feats1], 0).contiguous()
        E = torch.cat([embeddings0, embeddings1], 0).contiguous()
        idx = torch.cat([fps_idx0, fps_idx1], 0).contiguous()
        out = _pem.sparse_to_dense_transformer(D, E, idx, T)
        return out[:B], out[B:]
