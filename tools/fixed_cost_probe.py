"""Fixed cost (prologue + epilogue + launch) and cost per K tile of the four-wave GEMM kernel at M = 65536, N = 2048, for several
epilogues: a line fit of the device time over K = 1024, 2048, 4096.  python tools/fixed_cost_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

bf = torch.bfloat16
M, N = 65536, 2048
gam, bet, bias = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(bf)
variants = {
    "operand-type output only": dict(out_f32=False, out_op=True),
    "one GroupNorm output (gs 64)": dict(out_f32=False, gn=[(gam, bet, 64)]),
    "bias + residual + operand output + one GroupNorm": dict(out_f32=False, out_op=True, bias=bias, residual=res, gn=[(gam, bet, 64)]),
    "operand output + two GroupNorm outputs (64, 128)": dict(out_f32=False, out_op=True, bias=bias, gn=[(gam, bet, 64), (gam, bet, 128)]),
}
for name, kw in variants.items():
    ts = {}
    for K in (1024, 2048, 4096):
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        for _ in range(3):
            ops._raw_gemm([A], W, dtype="bf16", **kw)
        torch.cuda.synchronize()
        with ops.timing(64) as tm:
            for _ in range(10):
                ops._raw_gemm([A], W, dtype="bf16", **kw)
            torch.cuda.synchronize()
        ms = sorted(r["ms"] for r in tm.records[1:-1])
        ts[K] = ms[len(ms) // 2] * 1e3
        del A, W
    per_tile = (ts[4096] - ts[1024]) / 48          # us per 64 of K per launch
    fixed = ts[1024] - 16 * per_tile
    print("%-52s K=1024 %.0f us  2048 %.0f us  4096 %.0f us   per K tile %.2f us  fixed %.0f us (%.0f %% of the K=2048 launch)" %
          (name, ts[1024], ts[2048], ts[4096], per_tile, fixed, 100 * fixed / ts[2048]))
