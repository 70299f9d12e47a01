"""Does a few-row GEMM launch find its weights in L2 when the launch before it read the same weights?  Per layer shape at a batch of
32: 300 launches on ONE weight matrix against 300 launches that rotate through copies of it (160 MB of copies: more than the 32 MB
of L2, less than the Infinity Cache; 640 MB: more than both).  Median of the library's per-launch records.
python tools/w_residency_probe.py"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

bf = torch.bfloat16
DEV = "cuda"
SHAPES = [([512], 256, (8,)), ([512], 1024, (32,)), ([1024], 2048, (64,)), ([2048], 2048, (64,)), ([2048, 2048], 1024, (32,)), ([1024, 1024], 512, (16,))]
scratch = torch.zeros(4096 + 16 * 64 * 2048 * 4, device=DEV, dtype=torch.uint8)
for Ks, N, gs in SHAPES:
    As = [torch.randn(128, k, device=DEV).to(bf) for k in Ks]
    wbytes = N * sum(Ks) * 2
    gn = [(torch.ones(N, device=DEV), torch.zeros(N, device=DEV), g) for g in gs]
    kw = dict(bias=torch.randn(N, device=DEV), out_f32=not gs, out_op=bool(gs), gn=gn, m_valid=32, dtype="bf16")
    line = "K %-14s N %4d W %5.1f MB" % (Ks, N, wbytes / 1e6)
    for name, total in (("same", 0), ("160MB", 160e6), ("640MB", 640e6)):
        ncopy = max(1, int(total // wbytes))
        Ws = [(torch.randn(N, sum(Ks), device=DEV) / sum(Ks) ** 0.5).to(bf) for _ in range(ncopy)]
        for i in range(20):
            ops._raw_gemm(As, Ws[i % ncopy], splitk_scratch=scratch, **kw)
        torch.cuda.synchronize()
        with ops.timing(512) as tm:
            for i in range(300):
                ops._raw_gemm(As, Ws[i % ncopy], splitk_scratch=scratch, **kw)
            torch.cuda.synchronize()
        ms = sorted(r["ms"] for r in tm.records[10:-1])
        line += "  %s %6.1f us" % (name, ms[len(ms) // 2] * 1e3)
        del Ws
    print(line, flush=True)
