"""Repeats a few reverse steps of the sampling loop (bf16, fixed seed) at several batch sizes and compares the latents bit for bit
with the first repetition: every GEMM kernel family of the sampling path under back-to-back launches.
python tools/sample_repro.py [repetitions]"""
import sys
import torch
sys.path.insert(0, ".")
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bench.load_runtime()
dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev)
for B, steps in ((32, 40), (64, 40), (200, 30), (512, 20), (1024, 20), (2048, 12), (4096, 10), (65536, 6)):
    first, bad = None, 0
    for r in range(reps):
        out = model.p_sample_loop((B, 512), None, seed=7, t_start=999, t_stop=999 - steps + 1)
        torch.cuda.synchronize()
        if first is None:
            first = out.clone()
        elif not torch.equal(out, first):
            bad += 1
    print("batch %6d, %2d steps: %d of %d repetitions differ from the first" % (B, steps, bad, reps - 1), flush=True)
