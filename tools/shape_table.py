"""Every GEMM launch of one denoising step at batch 65536 (bf16), by layer shape: device time, share of the step, fraction of the bf16 peak.
python tools/shape_table.py"""
import sys
import torch
sys.path.insert(0, ".")
import bench

bench.load_runtime()
dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev)
B = 65536
fn = lambda: model.p_sample_loop((B, 512), None, seed=1, t_start=999, t_stop=992)
recs, clean_ms, over = bench.timed_window(fn)
tot = sum(r["ms"] for r in recs)
shapes = {}
for r in recs:
    k = (r["kind"], r["N"], r["K"])
    a = shapes.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
print("window %.3f ms for 8 steps; per step %.3f ms" % (tot, tot / 8))
for k, a in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
    print("kind %d N %4d K %4d  launches/step %.1f  avg %.1f us  share %.3f  frac of peak %.3f" % (k[0], k[1], k[2], a[0] / 8, a[1] / a[0] * 1e3, a[1] / tot,
                                                                                             a[2] / (a[1] * 1e-3) / 2.5e15 if a[1] else 0))
