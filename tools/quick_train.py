"""Manual timing of the training step (config 5 shape: emb 512 / hidden 256, per-GPU batch 4096)."""
import sys, time
import torch
sys.path.insert(0, ".")
import adcdiff
from oracle.weights import make_weights

def run(dtype, B=4096, steps=5):
    cfg = dict(input_dim=512, hidden_dim=256, output_dim=512, compute_dtype=dtype, max_batch_size=B, learning_rate=1e-4)
    m = adcdiff.DiffusionModel(cfg)
    w = make_weights(1234, 512, 256, 512)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    x = torch.randn(B, 512, device="cuda")
    losses = [m.train_step(x, 0)["loss"] for _ in range(2)]
    torch.cuda.synchronize()
    t0 = time.time()
    for s in range(steps):
        losses.append(m.train_step(x, s)["loss"])
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    print(f"{dtype} B={B}: {dt*1e3:.2f} ms/step, {B/dt:.0f} samples/s, {3*88.08e6*B/dt/1e12:.1f} TFLOP/s (3x forward-as-executed), losses {losses[0]:.4f} -> {losses[-1]:.4f}", flush=True)

if __name__ == "__main__":
    dts = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "fp32"]
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    for d in dts:
        run(d, B)
