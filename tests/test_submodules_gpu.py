"""The drop-in sub-modules of pem/transformer.py called DIRECTLY, as a user of the reference's transformer.py can call them
(PEM/model/transformer.py: SinusoidalPositionalEmbedding :259-285, MultiHeadAttention :95-150, AttentionLayer :152-181, AttentionOutput
:184-199, TransformerLayer :202-226, RPEMultiHeadAttention :366-420, RPEAttentionLayer :423-458, RPETransformerLayer :461-479,
LinearAttention :532-578, LinearAttentionLayer :581-609, LinearTransformerLayer :612-622) against the reference's own outputs:
tests/golden/submodules.npz, transformer.npz, sparse_to_dense.npz (written by oracle/gen_golden.py from the reference modules with
the same seeded inputs and the weights of synth.make_pem_weights(1))."""
import numpy as np
import pytest
import torch

from tests._util import golden

pytestmark = pytest.mark.gpu


def _close(got, want, atol, what):
    got = got.detach().float().cpu().numpy()
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.isfinite(got).all(), what
    d = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert d <= atol, "%s: max abs diff %.3e > %.1e" % (what, d, atol)


def _layer_inputs(seed, B=1, n=197):  # the generator of oracle/gen_golden.py
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, n, 256, generator=g)
    y = torch.randn(B, n, 256, generator=g)
    e0 = 0.5 * torch.randn(B, n, n, 256, generator=g)
    e1 = 0.5 * torch.randn(B, n, n, 256, generator=g)
    return x, y, e0, e1


def _sub(sd, prefix):
    return {k[len(prefix) + 1:]: v for k, v in sd.items() if k.startswith(prefix + ".")}


@pytest.fixture(scope="module")
def sd():
    from sam6d_hip import synth
    return synth.make_pem_weights(1)


@pytest.fixture(scope="module")
def geo_tr(sd, dev):
    import transformer as T
    m = T.GeometricTransformer(blocks=["self", "cross"], d_model=256, num_heads=4, dropout=None, activation_fn="ReLU",
                               return_attention_scores=False).eval()
    m.load_state_dict(_sub(sd, "coarse_point_matching.transformers.0"), strict=True)
    return m.to(dev)


def test_sinusoidal_positional_embedding(dev):
    import transformer as T
    g = golden("submodules")
    emb = T.SinusoidalPositionalEmbedding(256)
    # div_term is a registered buffer (it travels with the checkpoint): take the reference's bits -- torch.exp on another host CPU
    # differs in the last bit for some entries, which moves sin(866 w) by 2e-5
    emb.load_state_dict({"div_term": torch.from_numpy(g["sin_div_term"])}, strict=True)
    emb = emb.to(dev)
    out = emb(torch.from_numpy(g["sin_idx"]).to(dev))
    _close(out, g["sin"], 2e-7, "sinusoid embedding (arguments up to 866)")


def test_multi_head_attention_and_layer(dev, geo_tr):
    g = golden("submodules")
    x, y, _, _ = _layer_inputs(int(g["seed"]))
    x, y = x.to(dev), y.to(dev)
    lay = geo_tr.layers[1]
    hid, sc = lay.attention.attention(x, y, y)
    assert sc.shape == (1, 4, 197, 197)
    _close(hid[:, ::4], g["mha_hidden"], 2e-5, "MultiHeadAttention hidden_states")
    _close(sc[:, :, ::8], g["mha_scores"], 2e-6, "MultiHeadAttention attention_scores")
    out, sc2 = lay.attention(x, y)
    _close(out[:, ::4], g["attn_out"], 5e-5, "AttentionLayer output_states")
    assert torch.equal(sc, sc2)
    full, _ = lay(x, y)
    _close(full, golden("transformer")["cross"], 5e-5, "TransformerLayer output_states")
    with pytest.raises(NotImplementedError):
        lay(x, y, memory_masks=torch.zeros(1, 197, dtype=torch.bool, device=dev))


def test_rpe_attention_modules(dev, geo_tr):
    g = golden("submodules")
    x, _, e0, _ = _layer_inputs(int(g["seed"]))
    x, e0 = x.to(dev), e0.to(dev)
    lay = geo_tr.layers[0]
    hid, sc = lay.attention.attention(x, x, x, e0)
    _close(hid[:, ::4], g["rpe_hidden"], 3e-5, "RPEMultiHeadAttention hidden_states")
    _close(sc[:, :, ::8], g["rpe_scores"], 3e-6, "RPEMultiHeadAttention attention_scores")
    out, _ = lay.attention(x, x, e0)
    _close(out[:, ::4], g["rpe_attn_out"], 5e-5, "RPEAttentionLayer output_states")
    _close(lay.output(x)[:, ::4], g["ffn_out"], 5e-5, "AttentionOutput")
    full, _ = lay(x, x, e0)
    _close(full, golden("transformer")["rpe"], 5e-5, "RPETransformerLayer output_states")


def test_linear_attention_modules(dev, sd):
    import transformer as T
    g = golden("submodules")
    m = T.SparseToDenseTransformer(256, num_heads=4, sparse_blocks=["self", "cross"], dropout=None, activation_fn="ReLU",
                                   focusing_factor=3, with_bg_token=True, replace_bg_token=True).eval()
    m.load_state_dict(_sub(sd, "fine_point_matching.transformers.0"), strict=True)
    m = m.to(dev)
    gen = torch.Generator().manual_seed(int(g["seed_dense"]))
    d0 = torch.randn(1, 2049, 256, generator=gen)
    d1 = torch.randn(1, 2049, 256, generator=gen)
    q_in, m_in = d0[:, 1:].contiguous().to(dev), d1[:, 1:197].contiguous().to(dev)
    la = m.dense_layer.attention.attention(q_in, m_in, m_in)
    _close(la[:, ::16], g["linattn"], 2e-5, "LinearAttention")
    lal = m.dense_layer.attention(q_in, m_in)
    _close(lal[:, ::16], g["linattn_layer"], 5e-5, "LinearAttentionLayer")
    lin = m.dense_layer(q_in, m_in)
    _close(lin[:, ::8], golden("sparse_to_dense")["lin_rows"], 5e-5, "LinearTransformerLayer")
