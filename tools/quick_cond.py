"""Manual timing of conditional vs unconditional sampling steps (bf16, batch 65536)."""
import sys, time
import torch
sys.path.insert(0, ".")
import adcdiff
from oracle.weights import make_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
m = adcdiff.DiffusionModel(dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype="bf16", memory_monitor_enabled=False))
m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, 512, 256, 512).items()})
cond = torch.randn(B, 512, device="cuda")
for name, c in (("unconditional", None), ("conditional", cond), ("conditional 1-row", cond[:1].expand(B, -1).contiguous())):
    m.p_sample_loop((B, 512), condition=c, seed=1, t_start=999, t_stop=995)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.p_sample_loop((B, 512), condition=c, seed=1, t_start=999, t_stop=960)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40
    print(f"{name}: {dt*1e3:.3f} ms/step, {B/dt/1000:.0f} samples/s at T=1000", flush=True)
t0 = time.perf_counter()
out = m.sample(B, condition=cond, use_ddim=True, ddim_steps=50, seed=1)
torch.cuda.synchronize()
print(f"DDIM-50 conditional: {(time.perf_counter()-t0)*1e3:.0f} ms for {B} samples")
