"""Manual A/B of the 128-row and 256-row GEMM kernels on the denoiser's layer shapes (device time via torch events)."""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

M = 65536
shapes = [(256, [512], 1), (256, [256, 512], 3), (512, [256], 1), (512, [512, 256], 3), (1024, [512], 1), (1024, [1024, 512], 3),
          (2048, [1024], 1), (2048, [2048, 1024], 3), (2048, [2048], 1), (2048, [2048], 2), (1024, [2048, 2048], 1),
          (1024, [1024, 2048, 2048], 2), (512, [1024, 1024], 1), (512, [512, 1024, 1024], 2), (256, [512, 512], 1),
          (256, [256, 512, 512], 2), (256, [256, 256], 1), (256, [256, 256, 256], 2), (512, [256], 0)]
bf = torch.bfloat16
for N, Ks, nout in shapes:
    As = [torch.randn(M, k, device="cuda").to(bf) for k in Ks]
    W = (torch.randn(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    gn = [(gam, bet, max(8, N // 32))] * min(nout, 2)
    res = {}
    for hint in (128, 256):
        kw = dict(bias=bias, gn=gn, out_f32=(nout == 0), out_op=(nout == 3), dtype="bf16", tile_hint=hint)
        for _ in range(2):
            ops.gemm(As, W, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.gemm(As, W, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[hint] = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * M * N * sum(Ks)
    print(f"N={N:5d} K={sum(Ks):5d} outs={nout}: 128-tile {res[128]:8.1f} us ({fl/res[128]/1e6:7.1f} TF)   256-tile {res[256]:8.1f} us ({fl/res[256]/1e6:7.1f} TF)", flush=True)
