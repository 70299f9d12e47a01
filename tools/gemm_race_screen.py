"""Manual race screen for a GEMM kernel variant: python tools/gemm_race_screen.py HINT [iters]
Runs every layer shape `iters` times and compares each run's outputs bit for bit with the single-stage 128-row kernel's.
HINT 32 (the one-wave kernel) runs 128 padded rows with 32 and 17 live ones and compares the live rows.
HINT 327 (the few-row kernel that splits K over workgroups) is compared with its OWN first run: its results are reproducible bit for
bit (fixed split-order reduction) but differ from the tile kernels' in float32 summation order; the arrival counters must end at zero."""
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
scratch = torch.zeros(4096 + 16 * 64 * 2048 * 4, device="cuda", dtype=torch.uint8) if hint == 327 else None
def run(As, W, h, **kw):
    if hint == 327 and h == 327:
        return ops._raw_gemm(As, W, tile_hint=327, splitk_scratch=scratch, **kw)
    return ops.gemm(As, W, tile_hint=h, **kw)
for M, live in ([(128, 32), (128, 17), (128, 64)] if hint == 327 else [(128, 32), (128, 17), (128, 64), (256, 200)] if hint in (32, 325) else [(65536, 0), (4096, 0), (256, 0)]):
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
        ref = [x[:rows].clone() for x in flat(run(As, W, 327 if hint == 327 else 1, **kw))]
        n_bad = 0
        for it in range(iters if M == 65536 else 4 * iters):
            out = flat(run(As, W, hint, **kw))
            if not all(torch.equal(a, b[:rows]) for a, b in zip(ref, out)):
                n_bad += 1
        bad += n_bad
        print(f"M={M} live={rows} N={N} K={Ks}: {n_bad} mismatching runs", flush=True)
if scratch is not None:
    left = int(scratch[:4096].view(torch.int32).abs().sum())
    print("arrival counters left non-zero:", left)
    bad += left
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
