"""Per-layer-shape timing of the few-row GEMM kernels at a batch of 32: the default one-wave / loader-wave kernels against the
split-K kernel (tile_hint 327), 300 launches each, median of the library's per-launch records.  python tools/splitk_probe.py [rows]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for kv in sys.argv[3:]:   # debug switches of the library, key=value
    adcdiff._lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
bf = torch.bfloat16
DEV = "cuda"
# (K segments, N, GroupNorm group sizes of the outputs) of the sampling step's layers at emb 512 / hidden 256
SHAPES = [([512], 256, (8,)), ([256, 512], 256, (8, 16)), ([256], 512, (16,)), ([512, 256], 512, (16, 16)), ([512], 1024, (32,)), ([1024, 512], 1024, (32, 32)),
          ([1024], 2048, (64,)), ([2048, 1024], 2048, (64, 128)), ([2048], 2048, (64,)), ([2048], 2048, (64, 64)), ([2048, 2048], 1024, (32,)),
          ([1024, 2048, 2048], 1024, (64,)), ([1024, 1024], 512, (16,)), ([512, 1024, 1024], 512, (32,)), ([512, 512], 256, (8,)), ([256, 512, 512], 256, (16,)),
          ([256, 256], 256, (8,)), ([256, 256, 256], 256, (8,)), ([256], 512, ())]
if len(sys.argv) > 2 and sys.argv[2] == "wide":   # what the 128-channel group of encoder.3's second GEMM costs
    SHAPES = [([2048, 1024], 2048, (64, 128)), ([2048, 1024], 2048, (64, 64)), ([2048, 1024], 2048, (128,)), ([2048, 1024], 2048, (64,)),
              ([2048, 1024], 2048, ()), ([1024, 2048, 2048], 1024, (64,)), ([1024, 2048, 2048], 1024, (32,)), ([1024, 2048, 2048], 1024, ())]
scratch = torch.zeros(4096 + 16 * 64 * 2048 * 4, device=DEV, dtype=torch.uint8)
tot = {}
for Ks, N, gs in SHAPES:
    As = [torch.randn(128, k, device=DEV).to(bf) for k in Ks]
    W = (torch.randn(N, sum(Ks), device=DEV) / sum(Ks) ** 0.5).to(bf)
    gn = [(torch.ones(N, device=DEV), torch.zeros(N, device=DEV), g) for g in gs]
    kw = dict(bias=torch.randn(N, device=DEV), out_f32=not gs, out_op=bool(gs), gn=gn, m_valid=rows, dtype="bf16")
    line = "K %-18s N %4d gs %-10s" % (Ks, N, gs)
    for name, hint, sc in (("default", 0, None), ("split", 327, scratch)):
        for _ in range(20):
            ops._raw_gemm(As, W, tile_hint=hint, splitk_scratch=sc, **kw)
        torch.cuda.synchronize()
        with ops.timing(512) as tm:       # an event in front of every launch: a record spans to the next launch's event
            for _ in range(300):
                ops._raw_gemm(As, W, tile_hint=hint, splitk_scratch=sc, **kw)
            torch.cuda.synchronize()
        ms = sorted(r["ms"] for r in tm.records[10:-1])
        us = ms[len(ms) // 2] * 1e3       # median: the host's launch rate bounds the spans from below (a few of them are long)
        tot[name] = tot.get(name, 0) + us
        line += "  %s %6.1f us" % (name, us)
    print(line)
print("sum", {k: round(v, 1) for k, v in tot.items()})
