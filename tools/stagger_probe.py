"""Device time of the four-wave GEMM at M = 65536, N = 2048, K = 2048 / 4096 for several epilogues, against the start delay per
XCD index (`gemm_stagger`, 10 ns ticks).  python tools/stagger_probe.py [ticks ...]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops, _lib

bf = torch.bfloat16
M, N = 65536, 2048
ticks = [int(a) for a in sys.argv[1:]] or [0, 100, 200, 300, 400, 600, 800]
gam, bet, bias = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
res = torch.randn(M, N, device="cuda").to(bf)
variants = {
    "operand output": dict(out_f32=False, out_op=True),
    "one GroupNorm output": dict(out_f32=False, gn=[(gam, bet, 64)]),
    "bias + residual + operand + GroupNorm": dict(out_f32=False, out_op=True, bias=bias, residual=res, gn=[(gam, bet, 64)]),
    "operand + two GroupNorm outputs": dict(out_f32=False, out_op=True, bias=bias, gn=[(gam, bet, 64), (gam, bet, 128)]),
}
print("%-40s %5s " % ("epilogue", "K") + " ".join("%7d" % t for t in ticks))
for name, kw in variants.items():
    for K in (2048, 4096):
        A = torch.randn(M, K, device="cuda").to(bf)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        row = []
        for tk in ticks:
            _lib.debug_set("gemm_stagger", tk)
            for _ in range(3):
                ops._raw_gemm([A], W, dtype="bf16", **kw)
            torch.cuda.synchronize()
            with ops.timing(64) as tm:
                for _ in range(10):
                    ops._raw_gemm([A], W, dtype="bf16", **kw)
                torch.cuda.synchronize()
            ms = sorted(r["ms"] for r in tm.records[1:-1])
            row.append(ms[len(ms) // 2] * 1e3)
        _lib.debug_set("gemm_stagger", 0)
        print("%-40s %5d " % (name, K) + " ".join("%7.0f" % v for v in row), flush=True)
        del A, W
