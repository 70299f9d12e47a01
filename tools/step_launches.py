"""The launches of one denoising step at batch 65536 (bf16) in launch order: kind, shape, device time (mean over 8 steps).
python tools/step_launches.py [debug_key=value ...]"""
import sys
import torch
sys.path.insert(0, ".")
import bench

bench.load_runtime()
for kv in sys.argv[1:]:
    from adcdiff import _lib
    _lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev)
B = 65536
fn = lambda: model.p_sample_loop((B, 512), None, seed=1, t_start=999, t_stop=992)
recs, clean_ms, over = bench.timed_window(fn)
per = len(recs) // 8
tot = 0.0
for i in range(per):
    rs = [recs[s * per + i] for s in range(1, 8)]
    assert all(r["N"] == rs[0]["N"] and r["K"] == rs[0]["K"] for r in rs)
    ms = sum(r["ms"] for r in rs) / len(rs)
    tot += ms
    fl = rs[0]["flops"]
    print("%2d kind %d M %6d N %4d K %4d  %7.1f us  %5.3f of peak" % (i, rs[0]["kind"], rs[0]["M"], rs[0]["N"], rs[0]["K"], ms * 1e3, fl / (ms * 1e-3) / 2.5e15 if fl else 0))
print("step %.3f ms" % tot)
