"""Manual race screen for a GEMM kernel variant: python tools/gemm_race_screen.py HINT [iters]
Runs every layer shape `iters` times and compares each run's outputs bit for bit with the single-stage 128-row kernel's.
HINT 32 (the one-wave kernel) runs 128 padded rows with 32 and 17 live ones and compares the live rows."""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

hint = int(sys.argv[1])
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
shapes = [(256, [256, 512], 3), (512, [512, 256], 3), (1024, [1024, 512], 3), (2048, [1024], 1), (2048, [2048, 1024], 3), (2048, [2048], 1),
          (1024, [2048, 2048], 1), (1024, [1024, 2048, 2048], 2), (512, [512, 1024, 1024], 2), (256, [256, 512, 512], 2), (256, [64], 1),
          (256, [128, 64], 1), (512, [64, 64, 64], 2)]
bf = torch.bfloat16
bad = 0
for M, live in ([(128, 32), (128, 17), (128, 64), (256, 200)] if hint in (32, 325) else [(65536, 0), (4096, 0), (256, 0)]):
    for N, Ks, nout in shapes:
        As = [torch.randn(M, k, device="cuda").to(bf) for k in Ks]
        W = (torch.randn(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
        bias = torch.randn(N, device="cuda")
        gam, bet = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
        gn = [(gam, bet, max(8, N // 32))] * min(nout, 2)
        kw = dict(bias=bias, gn=gn, out_op=(nout == 3), dtype="bf16", m_valid=live)
        rows = live if live else M
        def flat(r):
            xs = r if isinstance(r, (tuple, list)) else [r]
            fx = []
            for x in xs:
                if isinstance(x, (tuple, list)): fx += list(x)
                elif x is not None: fx.append(x)
            return fx
        ref = [x[:rows].clone() for x in flat(ops.gemm(As, W, tile_hint=1, **kw))]
        n_bad = 0
        for it in range(iters if M == 65536 else 4 * iters):
            out = flat(ops.gemm(As, W, tile_hint=hint, **kw))
            if not all(torch.equal(a, b[:rows]) for a, b in zip(ref, out)):
                n_bad += 1
        bad += n_bad
        print(f"M={M} live={rows} N={N} K={Ks}: {n_bad} mismatching runs", flush=True)
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
