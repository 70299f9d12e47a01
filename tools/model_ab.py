"""Manual A/B of two builds of the library on the whole denoising step (bf16, batch 65536), interleaved rounds in one
process: python tools/model_ab.py libA.so libB.so"""
import statistics
import sys
import time
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff import _lib
from oracle.weights import make_weights

def bind(path):
    _lib._lib = None
    _lib.LIB_PATH = path
    return _lib.load()

libs = [bind(sys.argv[1]), bind(sys.argv[2])]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
w = make_weights(1234, 512, 256, 512)
models = []
for lib in libs:
    _lib._lib = lib
    m = adcdiff.DiffusionModel(dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype="bf16", memory_monitor_enabled=False))
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    m.p_sample_loop((B, 512), seed=1, t_start=999, t_stop=996)
    models.append(m)
t = [[], []]
outs = [None, None]
for rnd in range(6):
    for i in (0, 1):
        _lib._lib = libs[i]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs[i] = models[i].p_sample_loop((B, 512), seed=1, t_start=999, t_stop=970)
        torch.cuda.synchronize()
        t[i].append((time.perf_counter() - t0) / 30 * 1e3)
a, b = statistics.median(t[0]), statistics.median(t[1])
print(f"ms per denoising step: A {a:.3f} (min {min(t[0]):.3f})  B {b:.3f} (min {min(t[1]):.3f})  B/A {b/a:.4f}  max|A-B| {float((outs[0]-outs[1]).abs().max()):.3g}")
