"""Manual yardstick: the library GEMM (torch.matmul -> hipBLASLt/rocBLAS, bf16) on the denoiser's shapes next to adcdiff_gemm with
a bare epilogue (one operand-type output), on random and on zero operands.  Says how far the hand-written main loop is from what
the vendor library reaches on the same chip under the same clocks; it is not a parity test.
    python tools/gemm_yardstick.py [M] [iters]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff  # noqa: F401
from adcdiff import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bf = torch.bfloat16
SHAPES = [(2048, 2048), (2048, 1024), (1024, 4096), (1024, 1024), (512, 2048), (512, 512), (256, 512), (2048, 4096)]


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for zeros in (False, True):
    for N, K in SHAPES:
        if zeros:
            A = torch.zeros(M, K, device="cuda", dtype=bf)
            W = torch.zeros(N, K, device="cuda", dtype=bf)
        else:
            A = torch.randn(M, K, device="cuda").to(bf)
            W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(bf)
        Wt = W.t()
        out = torch.empty(M, N, device="cuda", dtype=bf)
        fl = 2.0 * M * N * K
        t_lib = timed(lambda: torch.matmul(A, Wt, out=out))
        t_own = timed(lambda: ops.gemm([A], W, out_f32=False, out_op=True, dtype="bf16"))
        print(f"{'zero' if zeros else 'rand'} M={M} N={N} K={K}: library {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TF | "
              f"adcdiff {t_own:8.1f} us {fl / t_own / 1e6:7.1f} TF | ratio {t_lib / t_own:.3f}", flush=True)
