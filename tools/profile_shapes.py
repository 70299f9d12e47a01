"""Manual profiling aid: per-launch device time of every kernel class and GEMM shape in one denoising step."""
import sys, collections
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops
from oracle.weights import make_weights

def main(dtype="bf16", B=65536, steps=4):
    cfg = dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype=dtype)
    m = adcdiff.DiffusionModel(cfg)
    w = make_weights(1234, 512, 256, 512)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=997)
    torch.cuda.synchronize()
    with ops.timing(100 * steps) as tm:
        m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=1000 - steps)
        torch.cuda.synchronize()
    agg = collections.OrderedDict()
    for r in tm.records:
        key = (r["kind"], r["M"], r["N"], r["K"])
        a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]; a[3] += r["bytes"]
    tot = collections.defaultdict(float)
    print(f"{dtype} B={B}")
    for (kind, M, N, K), (n, ms, fl, by) in agg.items():
        tot[kind] += ms / steps
        print(f"kind={kind} M={M} N={N:5d} K={K:5d} x{n//steps:2d}/step  {ms/n*1e3:9.1f} us  {fl/ms/1e9 if fl else 0:8.1f} TF/s  {by/ms/1e6:8.1f} GB/s")
    print({k: round(v, 3) for k, v in tot.items()}, "ms per denoise step by kind; total", round(sum(tot.values()), 3))

if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "bf16", int(sys.argv[2]) if len(sys.argv) > 2 else 65536)
