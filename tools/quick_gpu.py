"""Manual first-contact script for a GPU box: timing of one denoiser step per dtype/batch."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import adcdiff
from oracle.weights import make_weights

def run(dtype, B, steps):
    cfg = dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype=dtype)
    m = adcdiff.DiffusionModel(cfg)
    w = make_weights(1234, 512, 256, 512)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=996)
    torch.cuda.synchronize()
    t0 = time.time()
    out = m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=1000 - steps)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    fl = 81.92e6 * B / dt / 1e12
    print(f"{dtype} B={B}: {dt*1e3:.3f} ms/step, {fl:.1f} TFLOP/s algorithmic, finite={bool(torch.isfinite(out).all())}", flush=True)

if __name__ == "__main__":
    run("fp32", 4096, 20)
    run("bf16", 4096, 20)
    run("bf16", 65536, 10)
