import sys, torch
sys.path.insert(0, ".")
import bench
bench.load_runtime()
dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev)
B = 65536
for name, fn in (("ddpm", lambda: model.p_sample_loop((B, 512), None, seed=1, t_start=999, t_stop=992)),
                 ("ddim eta 0", lambda: model.ddim_sample_loop((B, 512), None, 8, 0.0, seed=1)),
                 ("ddim eta 0.5", lambda: model.ddim_sample_loop((B, 512), None, 8, 0.5, seed=1))):
    recs, clean_ms, over = bench.timed_window(fn)
    per = len(recs) // 8
    for i in range(per):
        rs = [recs[s * per + i] for s in range(1, 8)]
        if rs[0]["N"] == 512 and rs[0]["K"] == 256:
            print(name, i, "kind", rs[0]["kind"], "%.1f us" % (sum(r["ms"] for r in rs) / len(rs) * 1e3))
