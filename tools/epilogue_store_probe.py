"""Four-wave GEMM at M = 65536, N = 2048, K = 2048 with the stores of all but the first row tile switched off (m_valid = 256):
what is left of the epilogue is its arithmetic and LDS traffic.  Meant for the library built with -DADCDIFF_STAMPS."""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

bf = torch.bfloat16
M, N, K = 65536, 2048, 2048
gam, bet, bias = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(bf)
A = torch.randn(M, K, device="cuda").to(bf)
W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
variants = {
    "operand output": dict(out_f32=False, out_op=True),
    "one GroupNorm output": dict(out_f32=False, gn=[(gam, bet, 64)]),
    "bias + residual + operand + GroupNorm": dict(out_f32=False, out_op=True, bias=bias, residual=res, gn=[(gam, bet, 64)]),
}
for name, kw in variants.items():
    for mv in (None, 256):
        for _ in range(3):
            ops._raw_gemm([A], W, dtype="bf16", m_valid=mv, **kw)
        torch.cuda.synchronize()
        with ops.timing(64) as tm:
            for _ in range(10):
                ops._raw_gemm([A], W, dtype="bf16", m_valid=mv, **kw)
            torch.cuda.synchronize()
        ms = sorted(r["ms"] for r in tm.records[1:-1])
        print("%-40s m_valid %-6s %7.0f us" % (name, mv, ms[len(ms) // 2] * 1e3), flush=True)
