"""Manual A/B of two builds of the library on GEMM shapes, interleaved rounds in one process:
python tools/gemm_ab.py libA.so libB.so [hint]   (prints median microseconds per shape and build)."""
import statistics
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import _lib, ops

def bind(path):
    _lib._lib = None
    _lib.LIB_PATH = path
    return _lib.load()

libs = [bind(sys.argv[1]), bind(sys.argv[2])]
hint = int(sys.argv[3]) if len(sys.argv) > 3 else 0
M = 65536
shapes = [(2048, [2048], -1), (2048, [2048], 1), (2048, [2048, 1024], 3), (2048, [1024], 1), (1024, [2048, 2048], 1),
          (1024, [1024, 1536], 3), (512, [512, 256], 3), (256, [512], 1), (512, [256], 0)]
bf = torch.bfloat16
for N, Ks, nout in shapes:
    As = [torch.randn(M, k, device="cuda").to(bf) for k in Ks]
    W = (torch.randn(N, sum(Ks), device="cuda") / sum(Ks) ** 0.5).to(bf)
    bias = torch.randn(N, device="cuda")
    gam, bet = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    gn = [(gam, bet, max(8, N // 32))] * min(max(nout, 0), 2)
    kw = dict(bias=bias, gn=gn, out_f32=(nout == 0), out_op=(nout == 3 or nout == -1), dtype="bf16", tile_hint=hint)
    t = [[], []]
    outs = []
    for rnd in range(7):
        for which in (0, 1):
            _lib._lib = libs[which]
            r = ops.gemm(As, W, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                r = ops.gemm(As, W, **kw)
            e1.record()
            torch.cuda.synchronize()
            t[which].append(e0.elapsed_time(e1) / 5 * 1e3)
            if rnd == 0:
                outs.append(r)
    def flat(r):
        xs = r if isinstance(r, (tuple, list)) else [r]
        flatx = []
        for x in xs:
            if isinstance(x, (tuple, list)): flatx += list(x)
            elif x is not None: flatx.append(x)
        return flatx
    diff = max(float((a.float() - b.float()).abs().max()) for a, b in zip(flat(outs[0]), flat(outs[1])))
    fl = 2.0 * M * N * sum(Ks)
    a, b = statistics.median(t[0]), statistics.median(t[1])
    print(f"N={N:5d} K={sum(Ks):5d} nout={nout:2d}: A {a:8.1f} us ({fl/a/1e6:7.1f} TF)  B {b:8.1f} us ({fl/b/1e6:7.1f} TF)  B/A time {b/a:.3f}  maxdiff {diff:.3g}", flush=True)
