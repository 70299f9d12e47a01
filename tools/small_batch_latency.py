"""Wall time of the generator's real call at its default batch: python tools/small_batch_latency.py [BATCH] [DTYPE]

MoleculeGenerator.generate_molecules samples 32 latents per call with DDIM-50 and a one-row condition
(molecule_generator.py:238-262).  Prints one JSON line with the median wall time of sample() for DDIM-50 and DDPM-1000,
conditional and unconditional, so that a change to the launch path (fewer launches, a captured graph) has a before/after."""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import adcdiff  # noqa: E402
from oracle.weights import make_weights  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
m = adcdiff.DiffusionModel(dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype=dtype, memory_monitor_enabled=False))
m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, 512, 256, 512).items()})
cond = torch.randn(1, 512, device="cuda")


def wall(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 3)


out = {"batch": B, "dtype": dtype}
out["ddim50_cond_ms"] = wall(lambda: m.sample(B, condition=cond, use_ddim=True, ddim_steps=50, seed=1), 9)
out["ddim50_uncond_ms"] = wall(lambda: m.sample(B, use_ddim=True, ddim_steps=50, seed=1), 9)
out["ddpm1000_cond_ms"] = wall(lambda: m.sample(B, condition=cond, use_ddim=False, seed=1), 3)
out["ddpm1000_uncond_ms"] = wall(lambda: m.sample(B, use_ddim=False, seed=1), 3)
print(json.dumps(out))
