"""Is the batch-32 sampling call bound by the host's launch rate?  Host time until adcdiff_ddim_sample_loop returns (everything enqueued)
against the time until the GPU has finished.  python tools/small_host_rate.py [batch]"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench
bench.load_runtime()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = bench.build_model("bf16", torch.device("cuda", 0))
for _ in range(3):
    model.ddim_sample_loop((B, 512), None, 50, 0.0, seed=1)
torch.cuda.synchronize()
for _ in range(5):
    t0 = time.perf_counter()
    model.ddim_sample_loop((B, 512), None, 50, 0.0, seed=1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch %d: host returns after %.2f ms, GPU done after %.2f ms" % (B, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
