"""Manual probe (GPU): the one-wave GEMM kernel (tile_hint 32) against the 128-row tile kernel for every combination of the
epilogue options on one small shape (N = 64, K = 64, 32 live rows); prints the largest difference per output (0.0 = bit-identical).
It was how the stale parameter vectors from a wrapped vmcnt counter were narrowed down to bias + GroupNorm."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops
rng = np.random.default_rng(0)
M, N, K, live = 128, 64, 64, 32
bf = torch.bfloat16
A = torch.tensor(rng.standard_normal((M, K)).astype(np.float32), device="cuda").to(bf)
W = torch.tensor(rng.standard_normal((N, K)).astype(np.float32), device="cuda").to(bf)
bias = torch.tensor(rng.standard_normal(N).astype(np.float32), device="cuda")
for kw in (dict(), dict(bias=bias)):
    o = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=32, **kw)
    t = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=128, **kw)
    d = (o - t)[:live].cpu().numpy()
    print(sorted(kw), "max diff", np.abs(d).max(), "rows differing", np.unique(np.nonzero(d)[0])[:40], "cols", np.unique(np.nonzero(d)[1])[:70])
    if "bias" in kw and np.abs(d).max() > 0:
        b = bias.cpu().numpy()
        print("diff row0[:8]", d[0, :8], "bias[:8]", b[:8])
table = torch.tensor(rng.standard_normal((7, N)).astype(np.float32), device="cuda")
idx = torch.tensor(rng.integers(0, 7, M).astype(np.int64), device="cuda")
res = torch.tensor(rng.standard_normal((M, N)).astype(np.float32), device="cuda").to(bf)
gam = torch.rand(N, device="cuda") + 0.5
bet = torch.randn(N, device="cuda")
for name, kw in (("rowbias", dict(rowbias=table, rowbias_idx=idx)), ("residual", dict(residual=res)), ("out_op", dict(out_op=True)),
                 ("gn4", dict(gn=[(gam, bet, 4)])), ("gn64", dict(gn=[(gam, bet, 64)])), ("gn4+64", dict(gn=[(gam, bet, 4), (gam, bet, 64)])),
                 ("all", dict(bias=bias, rowbias=table, rowbias_idx=idx, residual=res, out_op=True, gn=[(gam, bet, 4), (gam, bet, 64)]))):
    o = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=32, **kw)
    t = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=128, **kw)
    if not isinstance(o, tuple): o, t = (o,), (t,)
    for i, (a_, b_) in enumerate(zip(o, t)):
        d = (a_.float() - b_.float())[:live].cpu().numpy()
        print(name, i, "max diff", np.abs(d).max(), "rows", np.unique(np.nonzero(d)[0])[:40], "ncols", len(np.unique(np.nonzero(d)[1])))
print("---- pairs")
opts = dict(bias=dict(bias=bias), rowbias=dict(rowbias=table, rowbias_idx=idx), residual=dict(residual=res), out_op=dict(out_op=True),
            gn4=dict(gn=[(gam, bet, 4)]), gn2=dict(gn=[(gam, bet, 4), (gam, bet, 64)]))
import itertools
for r in (2, 3, 4, 5):
    for combo in itertools.combinations([k for k in opts if k != "gn4"], r):
        kw = {}
        for c in combo: kw.update(opts[c])
        o = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=32, **kw)
        t = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=128, **kw)
        if not isinstance(o, tuple): o, t = (o,), (t,)
        print(combo, [float((a_.float() - b_.float())[:live].abs().max()) for a_, b_ in zip(o, t)])
kw = {}
for c in opts:
    if c != "gn4": kw.update(opts[c])
o = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=32, **kw)[0]
t = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=128, **kw)[0]
plain = ops.gemm([A], W, dtype="bf16", m_valid=live, tile_hint=128)
print("o - plain", (o - plain)[0, :6].cpu().numpy(), "\nt - plain", (t - plain)[0, :6].cpu().numpy())
print("bias", bias[:6].cpu().numpy(), "\ntable[idx0]", table[idx[0], :6].cpu().numpy(), "\nres", res[0, :6].float().cpu().numpy())
