"""Manual probe of one GEMM shape: python tools/gemm_probe.py N K1[,K2..] nout hint iters  (nout: -1 = operand-type output only)."""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

N, Ks, nout, hint, iters = int(sys.argv[1]), [int(k) for k in sys.argv[2].split(",")], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
M = int(sys.argv[6]) if len(sys.argv) > 6 else 65536
bf = torch.bfloat16
As = [torch.randn(M, k, device="cuda").to(bf) for k in Ks]
W = (torch.randn(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
bias = torch.randn(N, device="cuda")
gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
gn = [(gam, bet, max(8, N // 32))] * min(max(nout, 0), 2)
kw = dict(bias=bias, gn=gn, out_f32=(nout == 0), out_op=(nout == 3 or nout == -1), dtype="bf16", tile_hint=hint)
for _ in range(3):
    ops.gemm(As, W, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.gemm(As, W, **kw)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
fl = 2.0 * M * N * sum(Ks)
print(f"M={M} N={N} K={sum(Ks)} nout={nout} hint={hint}: {us:.1f} us  {fl/us/1e6:.1f} TF", flush=True)
