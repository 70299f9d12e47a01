"""Manual: board power and shader clock while one GEMM variant runs back to back for a few seconds.
    python tools/power_probe.py own|lib rand|zero [N K hint]"""
import subprocess
import sys
import threading
import time
import torch
sys.path.insert(0, ".")
import adcdiff  # noqa: F401
from adcdiff import ops

which, data = sys.argv[1], sys.argv[2]
N, K = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2048, 2048)
hint = int(sys.argv[5]) if len(sys.argv) > 5 else 0
M = 65536
bf = torch.bfloat16
A = (torch.zeros if data == "zero" else torch.randn)(M, K, device="cuda").to(bf)
W = ((torch.zeros if data == "zero" else torch.randn)(N, K, device="cuda") / K ** 0.5).to(bf)
out = torch.empty(M, N, device="cuda", dtype=bf)
fn = (lambda: torch.matmul(A, W.t(), out=out)) if which == "lib" else (lambda: ops.gemm([A], W, out_f32=False, out_op=True, dtype="bf16", tile_hint=hint))
samples = []
stop = False


def sampler():
    while not stop:
        try:
            r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append(r.strip().splitlines()[-1])
        except Exception as e:
            samples.append(str(e))
        time.sleep(0.3)


th = threading.Thread(target=sampler)
th.start()
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    n += 1
    last = e0.elapsed_time(e1) / 50 * 1e3
stop = True
th.join()
print(f"{which} {data} N={N} K={K} hint={hint}: {last:.1f} us {2.0*M*N*K/last/1e6:.0f} TF (after {n} bursts)")
for s in samples[2:8]:
    print("   ", s[:200])
