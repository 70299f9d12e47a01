"""Main-phase time of the four-wave GEMM kernel against the number of K tiles (M = 65536, N = 2048, one GroupNorm output), for the
library built with -DADCDIFF_STAMPS (it prints the per-round means to stderr for every tenth launch): is the time in front of the
steady loop a constant, or do the first tiles cost more?  python tools/main_phase_curve.py 2> stamps.txt"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops

bf = torch.bfloat16
M, N = 65536, 2048
gam, bet = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
# settle the clock first
A = torch.randn(M, 2048, device="cuda").to(bf)
W = (torch.randn(N, 2048, device="cuda") / 45).to(bf)
for _ in range(40):
    ops._raw_gemm([A], W, dtype="bf16", out_f32=False, gn=[(gam, bet, 64)])
torch.cuda.synchronize()
del A, W
for tiles in (2, 3, 4, 5, 6, 8, 10, 12, 16, 24, 32, 48):
    K = 64 * tiles
    A = torch.randn(M, K, device="cuda").to(bf)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
    for _ in range(10):   # the stamps build dissects every tenth launch of the process
        ops._raw_gemm([A], W, dtype="bf16", out_f32=False, gn=[(gam, bet, 64)], tile_hint=2564)
    torch.cuda.synchronize()
    del A, W
