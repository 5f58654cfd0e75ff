"""smoke(): one small pass of the whole PEM matching path on cuda:0 (B=1) checked against the CPU oracle."""
import torch


def run():
    from oracle import pem_oracle as O
    from sam6d_hip import pem, synth
    dev = torch.device("cuda:0")
    sd = synth.make_pem_weights(1)
    W = pem.PemWeights(sd, dev)
    inp = synth.kat_inputs(B=1, seed=5)
    d = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in inp.items()}
    R, t, s, aux = pem.pem_match(d["dense_pm"], d["dense_fm"], d["dense_po"], d["dense_fo"], d["radius"], d["model"], W,
                                 d["rand"], return_aux=True)
    torch.cuda.synchronize()
    oR, ot, os_, oaux = O.pem_match(inp["dense_pm"], inp["dense_fm"], inp["dense_po"], inp["dense_fo"], inp["radius"],
                                    inp["model"], sd, inp["rand"], return_aux=True)
    assert torch.equal(aux["fps_idx_m"].cpu(), oaux["fps_idx_m"]) and torch.equal(aux["fps_idx_o"].cpu(), oaux["fps_idx_o"])
    dR, dt, ds = (R.cpu() - oR).abs().max().item(), (t.cpu() - ot).abs().max().item(), (s.cpu() - os_).abs().max().item()
    print("[smoke] PEM path vs oracle: max|dR| %.2e  max|dt| %.2e  |dscore| %.2e" % (dR, dt, ds))
    assert dR < 1e-4 and dt < 1e-4 and ds < 1e-4, "PEM path deviates from the CPU oracle beyond the 1e-4 contract"
