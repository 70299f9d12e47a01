"""Manual probe (run under rocprofv3 --kernel-trace --stats): 300 launches of ONE GEMM shape at a small batch (M = 128 padded
rows, 32 live), so that the kernel's average duration in the stats is that shape's.
python tools/small_m_probe.py N K GROUP_SIZE [HINT [ROW_PAD [LIVE_ROWS]]]   (ROW_PAD: extra elements between the rows of W in memory)"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops
N, K, gs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hint = int(sys.argv[4]) if len(sys.argv) > 4 else 0
pad = int(sys.argv[5]) if len(sys.argv) > 5 else 0
live = int(sys.argv[6]) if len(sys.argv) > 6 else 32
bf = torch.bfloat16
A = torch.randn(128, K, device="cuda").to(bf)
W = (torch.randn(N, K + pad, device="cuda") / K ** 0.5).to(bf)[:, :K]
bias = torch.randn(N, device="cuda")
gam, bet = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
for _ in range(300):
    ops.gemm([A], W, bias=bias, dtype="bf16", m_valid=live, gn=[(gam, bet, gs)], tile_hint=hint)
torch.cuda.synchronize()
