"""Repeats one-wave-kernel GEMM launches of the training forward's shapes (bf16, float32 output) and compares every result bit for bit
with the 128-row tile kernel's (same fragments, K order and epilogue).  Between the launches a large unrelated GEMM runs, as other
kernels do in a forward pass.  python tools/w1_repro.py [repetitions]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bf = torch.bfloat16
torch.manual_seed(5)
X = torch.randn(4096, 1024, device="cuda").to(bf)
Y = torch.randn(2048, 1024, device="cuda").to(bf)
for M in (1024,):
    for N, Ks, res in ((2048, [2048], True), (2048, [2048, 1024], False), (1024, [1024, 2048, 1024], False), (2048, [1024], False), (512, [512, 256], False),
                       (256, [256, 512], False), (512, [512], True)):
        As = [torch.randn(M, k, device="cuda").to(bf) for k in Ks]
        W = (torch.randn(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
        kw = dict(bias=torch.randn(N, device="cuda"), dtype="bf16", m_valid=M)
        if res:
            kw["residual"] = torch.randn(M, N, device="cuda").to(bf)
        else:
            kw["rowbias"] = torch.randn(1000, N, device="cuda")
            kw["rowbias_idx"] = torch.randint(0, 1000, (M,), device="cuda")
        want = ops._raw_gemm(As, W, tile_hint=128, **kw).clone()
        bad = 0
        rows = set()
        for r in range(reps):
            ops._raw_gemm([X], Y, dtype="bf16")
            out = ops._raw_gemm(As, W, **kw)
            if not torch.equal(out, want):
                bad += 1
                rows.update((out != want).any(1).nonzero().flatten().tolist())
        print("M %4d N %4d K %-18s residual %-5s: %3d of %d launches differ from the tile kernel%s" %
              (M, N, Ks, res, bad, reps, ("; rows " + str(sorted(rows)[:8]) + " ... " + str(sorted(rows)[-3:])) if rows else ""), flush=True)
