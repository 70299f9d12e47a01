"""Manual comparison of the GEMM kernels (tile hints 128 / 256 / 2561) on the denoiser's layer shapes at M = 65536,
interleaved rounds in one process; prints median microseconds and the largest output difference against hint 128."""
import statistics
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
hints = [int(h) for h in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2560, 2561]
zeros = len(sys.argv) > 3 and sys.argv[3] == "zeros"   # zero operands: shows how much of the time is the clock the chip holds on real data
shapes = [(256, [512], 1), (256, [256, 512], 3), (512, [256], 1), (512, [512, 256], 3), (1024, [512], 1), (1024, [1024, 512], 3),
          (2048, [1024], 1), (2048, [2048, 1024], 3), (2048, [2048], 1), (2048, [2048], 2), (1024, [2048, 2048], 1),
          (1024, [1024, 2048, 2048], 2), (512, [1024, 1024], 1), (512, [512, 1024, 1024], 2), (256, [512, 512], 1),
          (256, [256, 512, 512], 2), (256, [256, 256], 1), (256, [256, 256, 256], 1), (512, [256], 0)]
bf = torch.bfloat16
tot = {h: 0.0 for h in hints}
for N, Ks, nout in shapes:
    As = [(torch.zeros if zeros else torch.randn)(M, k, device="cuda").to(bf) for k in Ks]
    W = ((torch.zeros if zeros else torch.randn)(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    gam, bet = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
    gn = [(gam, bet, max(8, N // 32))] * min(max(nout, 0), 2)
    t = {h: [] for h in hints}
    outs = {}
    for rnd in range(6):
        for h in hints[rnd % len(hints):] + hints[:rnd % len(hints)]:   # rotate the order: position in the round matters
            kw = dict(bias=bias, gn=gn, out_f32=(nout == 0), out_op=(nout == 3), dtype="bf16", tile_hint=h)
            r = ops.gemm(As, W, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                r = ops.gemm(As, W, **kw)
            e1.record()
            torch.cuda.synchronize()
            t[h].append(e0.elapsed_time(e1) / 5 * 1e3)
            if h not in outs:
                outs[h] = r
    def flat(r):
        xs = r if isinstance(r, (tuple, list)) else [r]
        fx = []
        for x in xs:
            if isinstance(x, (tuple, list)): fx += list(x)
            elif x is not None: fx.append(x)
        return fx
    fl = 2.0 * M * N * sum(Ks)
    line = f"N={N:5d} K={sum(Ks):5d} nout={nout:2d}:"
    for h in hints:
        m = statistics.median(t[h]); tot[h] += m
        d = max(float((a.float() - b.float()).abs().max()) for a, b in zip(flat(outs[hints[0]]), flat(outs[h])))
        line += f"  [{h}] {m:7.1f} us {fl/m/1e6:6.0f} TF d={d:.3g}"
    print(line, flush=True)
print("sum of medians:", {h: round(v, 1) for h, v in tot.items()})
