"""Manual probe (GPU): random shapes through the few-row kernels (tile_hint 32 and 325) against the 128-row tile kernel, bit for bit.
python tools/w1_fuzz.py [CASES] [SEED] [big|fp32|mid]   big: 256-row multiples through the 256 x 256 kernels (hints 256, 2561, 2564) instead;
fp32: the default choice among the float32 tile kernels (64 x 64, 64 x 128, rings) against the single-stage 128-row kernel (hint 1)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
torch.manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bf = torch.bfloat16
bad = 0
big = len(sys.argv) > 3 and sys.argv[3] == "big"
f32 = len(sys.argv) > 3 and sys.argv[3] == "fp32"
mid = len(sys.argv) > 3 and sys.argv[3] == "mid"   # around the 1024-row limit of the default choice
if f32:
    bf = torch.float32
for c in range(cases):
    N = int(rng.choice([256, 512, 768, 1024] if big else [64, 128, 192, 256, 384, 512, 1024, 2048]))
    nseg = int(rng.integers(1, 4))
    Ks = [int(rng.choice([64, 128, 192, 256, 512, 1024, 2048])) for _ in range(nseg)]
    if f32:
        Ks = [k // 2 for k in Ks]   # K tiles of 32
    gss = [g for g in (4, 8, 16, 32, 64, 128) if N % g == 0 and (g <= 64 or N % 128 == 0)]
    n_gn = int(rng.integers(0, 3))
    gs = [int(rng.choice(gss)) for _ in range(n_gn)]
    M = int(rng.choice([256, 512, 768] if big else [1024, 1152, 2048] if mid else [128, 256]))
    live = int(rng.integers(1, M + 1)) if rng.random() < 0.5 or big or mid else int(rng.integers(1, 65))
    As = [torch.randn(M, k, device="cuda").to(bf) for k in Ks]
    W = (torch.randn(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
    kw = dict(dtype="fp32" if f32 else "bf16", m_valid=live)
    if rng.random() < 0.7: kw["bias"] = torch.randn(N, device="cuda")
    if rng.random() < 0.4:
        kw["rowbias"] = torch.randn(9, N, device="cuda"); kw["rowbias_idx"] = torch.randint(0, 9, (M,), device="cuda")
    elif rng.random() < 0.3:
        kw["rowbias"] = torch.randn(N, device="cuda")
    if rng.random() < 0.5: kw["residual"] = torch.randn(M, N, device="cuda").to(bf)
    if rng.random() < 0.5 or n_gn == 0: kw["out_op"] = True
    kw["gn"] = [(torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda"), g) for g in gs]
    def flat(r):
        return list(r) if isinstance(r, (tuple, list)) else [r]
    ref = flat(ops.gemm(As, W, tile_hint=1 if f32 else 128, **kw))
    for hint in ((256, 2561, 2564, 0) if big else (0, 3, 128) if f32 else (32, 325, 0)):
        out = flat(ops.gemm(As, W, tile_hint=hint, **kw))
        ok = all(torch.equal(a[:live], b[:live]) for a, b in zip(ref, out))
        if not ok:
            bad += 1
            for i, (a, b) in enumerate(zip(ref, out)):
                d = (a[:live].float() - b[:live].float()).abs()
                if d.max() > 0:
                    nz = d.nonzero()
                    print("  output", i, "max", float(d.max()), "n", len(nz), "rows", sorted(set(nz[:, 0].tolist()))[:8], "cols", sorted(set(nz[:, 1].tolist()))[:8])
            print("MISMATCH", dict(N=N, Ks=Ks, gs=gs, M=M, live=live, hint=hint, opts=sorted(k for k in kw if k not in ("dtype", "m_valid", "gn"))), flush=True)
torch.cuda.synchronize()
print("cases", cases, "mismatches", bad)
sys.exit(1 if bad else 0)
