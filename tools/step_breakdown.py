"""Manual per-launch breakdown of one denoising step: python tools/step_breakdown.py DTYPE BATCH"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import ops, _lib
from oracle.weights import make_weights

dtype, B = sys.argv[1], int(sys.argv[2])
m = adcdiff.DiffusionModel(dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype=dtype, memory_monitor_enabled=False))
m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, 512, 256, 512).items()})
m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=990)
with ops.timing(4096) as tm:
    m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=992)
    torch.cuda.synchronize()
recs = tm.records
per = len(recs) // 8
tot = 0.0
for i in range(per):
    rs = [recs[s * per + i] for s in range(2, 8)]
    ms = sorted(r["ms"] for r in rs)[len(rs) // 2]
    r = rs[0]
    tot += ms
    tf = r["flops"] / (ms * 1e-3) / 1e12 if r["flops"] else 0.0
    gbs = r["bytes"] / (ms * 1e-3) / 1e9
    print(f"kind {r['kind']} M={r['M']:6d} N={r['N']:5d} K={r['K']:5d}  {ms*1e3:8.1f} us  {tf:7.1f} TF  {gbs:7.0f} GB/s")
print(f"sum {tot*1e3:.0f} us per step")
