"""Time sam6d_pe_mlp_max alone at the bench shape (32 clouds x 2048 points, nsample 32 / 64)."""
import sys, os, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "openvino-sam-6d_amd"))
from sam6d_hip import _lib, pem, synth
dev = torch.device("cuda:0")
W = pem.PemWeights(synth.make_pem_weights(1), dev)
B, N = 32, 2048
g = torch.Generator().manual_seed(0)
pts = (torch.rand(B, N, 3, generator=g) - 0.5).to(dev)
for S in (32, 64):
    idx = torch.randint(0, N, (B, N, S), generator=g, dtype=torch.int32).to(dev)
    L = W.pe["mlp"][0 if S == 32 else 1]
    out = torch.zeros(B * N, 256, device=dev)
    def run():
        _lib.call("sam6d_pe_mlp_max", pts.data_ptr(), idx.data_ptr(), B, N, S, L[0]["w"].data_ptr(), L[0]["scale"].data_ptr(),
                  L[0]["shift"].data_ptr(), L[1]["w"].data_ptr(), L[1]["scale"].data_ptr(), L[1]["shift"].data_ptr(),
                  L[2]["w"].data_ptr(), L[2]["scale"].data_ptr(), L[2]["shift"].data_ptr(), out.data_ptr(), 256, 0,
                  torch.cuda.current_stream().cuda_stream)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tiles = B * N * S / 32
    print("S=%d: %.1f us/launch; %.0f matrix-pipe-bound us (2112 cyc/tile, 1024 SIMDs, 2.4 GHz) -> %.2f of it" %
          (S, us, tiles * 2112 / 1024 / 2400, tiles * 2112 / 1024 / 2400 / us))
