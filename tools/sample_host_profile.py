"""Manual probe (GPU): where the host time of one DDIM-50 sample() call at batch 32 goes (cProfile, cumulative)."""
import cProfile
import pstats
import sys
import time
import torch
sys.path.insert(0, ".")
import adcdiff
from oracle.weights import make_weights
m = adcdiff.DiffusionModel(dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype="bf16", memory_monitor_enabled=False))
m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, 512, 256, 512).items()})
for _ in range(3):
    m.sample(32, use_ddim=True, ddim_steps=50, seed=1)
torch.cuda.synchronize()
t0 = time.perf_counter(); x = m.sample(32, use_ddim=True, ddim_steps=50, seed=1); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"sample() returned after {(t1 - t0) * 1e3:.2f} ms, device idle after {(t2 - t0) * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    m.sample(32, use_ddim=True, ddim_steps=50, seed=1)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
