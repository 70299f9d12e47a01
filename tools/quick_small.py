"""Manual timing of small-batch sampling (the generator's default: batch 32, DDIM 50 steps; and the T = 1000 loop)."""
import sys, time
import torch
sys.path.insert(0, ".")
import adcdiff
from oracle.weights import make_weights

w = make_weights(1234, 512, 256, 512)
for dtype in ("bf16", "fp32"):
    m = adcdiff.DiffusionModel(dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype=dtype, memory_monitor_enabled=False))
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    for B in (32, 256, 1024):
        m.sample(B, use_ddim=True, ddim_steps=50, seed=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            m.sample(B, use_ddim=True, ddim_steps=50, seed=1)
        torch.cuda.synchronize()
        ddim = (time.perf_counter() - t0) / 5
        m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=900)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.p_sample_loop((B, 512), seed=1)
        torch.cuda.synchronize()
        ddpm = time.perf_counter() - t0
        print(f"{dtype} B={B}: DDIM-50 {ddim*1e3:.1f} ms ({ddim/50*1e6:.0f} us/step), DDPM-1000 {ddpm*1e3:.0f} ms ({ddpm*1e3:.0f} us/step), {B/ddpm:.0f} samples/s", flush=True)
