"""Host-side mathematics behind the GPU path, checked on the CPU: the Chebyshev reformulation of the geometric embedding
(sam6d_hip.pem.cheb_coefficients) against the sinusoid projection it replaces (PEM/model/transformer.py:259-285, 343-363)."""
import numpy as np
import torch

from tests._util import PKG  # noqa: F401  (sys.path)
from sam6d_hip import pem, synth


def _direct(weight, div_term, x):
    ph = x[:, None] * div_term[None, :]
    emb = np.stack([np.sin(ph), np.cos(ph)], -1).reshape(len(x), -1)
    return emb @ weight.T


def _cheb_eval(c, x, xmax, dtype):
    u = (x.astype(dtype) * dtype(2.0 / xmax) - dtype(1.0)).astype(dtype)
    T = [np.ones_like(u), u]
    for _ in range(2, c.shape[1]):
        T.append((dtype(2) * u * T[-1] - T[-2]).astype(dtype))
    return np.stack(T, 1).astype(np.float64) @ c.T


def test_chebyshev_expansion_reproduces_the_sinusoid_projection():
    sd = synth.make_pem_weights(1)
    div = sd["geo_embedding.embedding.div_term"].double().numpy()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(4000) * pem.GEO_XMAX, [0.0, pem.GEO_XMAX, 1e-6, pem.GEO_XMAX - 1e-6]])
    for name in ("proj_d", "proj_a"):
        w = sd["geo_embedding.%s.weight" % name]
        c = pem.cheb_coefficients(w, sd["geo_embedding.embedding.div_term"])
        assert c.shape == (256, pem.GEO_CHEB_K)
        want = _direct(w.double().numpy(), div, x)
        scale = np.abs(want).max()
        err64 = np.abs(_cheb_eval(c, x, pem.GEO_XMAX, np.float64) - want).max()
        assert err64 < 1e-9 * max(scale, 1.0), "%s: float64 recurrence error %.3e" % (name, err64)
        # the fp32 recurrence rpe_score_kernel runs (indices arrive as fp32)
        x32 = x.astype(np.float32)
        want32 = _direct(w.double().numpy(), div, x32.astype(np.float64))
        err32 = np.abs(_cheb_eval(c, x32, pem.GEO_XMAX, np.float32) - want32).max()
        # the reference's own fp32 evaluation of the same projection (sin/cos of an fp32 product) for comparison
        ph = (x32[:, None] * div.astype(np.float32)[None, :]).astype(np.float32)
        emb = np.stack([np.sin(ph), np.cos(ph)], -1).reshape(len(x32), -1).astype(np.float32)
        ref32 = np.abs(emb @ w.numpy().T - want32).max()
        assert err32 < 4e-6 * max(scale, 1.0), "%s: fp32 recurrence error %.3e" % (name, err32)
        assert err32 < 2.0 * ref32 + 1e-6, "%s: fp32 recurrence error %.3e vs the reference's own fp32 error %.3e" % (name, err32, ref32)


def test_chebyshev_range_covers_normalised_clouds():
    """d_idx = dist / 0.2 and a_idx = angle_deg / 15 of radius-normalised clouds stay below GEO_XMAX (the bg token does not:
    its pairs take the sinusoid kernel)."""
    assert 180.0 / 15.0 <= pem.GEO_XMAX
    inp = synth.config2_inputs(B=2, seed=3)
    for k in ("dense_pm", "dense_po"):
        p = inp[k]
        assert float(torch.cdist(p, p).amax()) / 0.2 < pem.GEO_XMAX


def test_options_object_resolves_env_and_gates_routes(monkeypatch):
    """pem.Options: the explicit replacement of the process-global matmul mode + SAM6D_* switches.  Environment switches resolve into it
    once; routes that exist only in the split-precision arithmetic are gated by the mode; the thread-local mode of the library follows
    an explicit override and falls back to the process default."""
    from sam6d_hip import _lib
    lib = _lib.load()
    assert lib.sam6d_get_thread_matmul_mode() == -1
    base = lib.sam6d_get_matmul_mode()
    o = pem.Options()
    assert o.mode == base and o.fused_block == (base >= 1) and o.microbatch == 1 and o.rpe_products == 0
    monkeypatch.setenv("SAM6D_FUSED_BLOCK", "0")
    monkeypatch.setenv("SAM6D_RPE_PRODUCTS", "3")
    monkeypatch.setenv("SAM6D_MICROBATCH", "2")
    e = pem.Options.from_env()
    assert not e.fused_block and not e.fused_front and not e.rows_linear and e.rpe_products == 3 and e.microbatch == 2
    assert pem.Options.from_env(fused_block=True).fused_block == (base >= 1)
    x = pem.Options(matmul_mode=0)
    assert x.mode == 0 and not (x.w16 or x.fused_block or x.fused_rpe or x.fused_out)
    assert x.replace(matmul_mode=1).mode == 1
    try:
        pem.Options(no_such_switch=1)
        assert False, "unknown field accepted"
    except TypeError:
        pass
    # thread override: visible through sam6d_get_matmul_mode, gone after -1
    assert lib.sam6d_set_thread_matmul_mode(0) == 0
    assert lib.sam6d_get_matmul_mode() == 0 and pem.Options().mode == 0
    assert lib.sam6d_set_thread_matmul_mode(-1) == 0
    assert lib.sam6d_get_matmul_mode() == base
    assert lib.sam6d_set_thread_matmul_mode(7) != 0
