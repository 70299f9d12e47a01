"""Repeats the taped training forward (bf16, dropout on, one seed) and compares the predicted noise and the loss gradient bit for bit
with the first repetition.  python tools/train_forward_repro.py [repetitions] [batch] [conditional 0|1] [debug_key=value ...]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff
from adcdiff.models import diffusion_model as dm

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
conditional = (sys.argv[3] if len(sys.argv) > 3 else "1") == "1"
dims = (512, 256, 512)
for kv in sys.argv[4:]:
    adcdiff._lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
m = adcdiff.DiffusionModel(dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype="bf16", max_batch_size=4096))
m.model.train()
gen = torch.Generator(device="cpu").manual_seed(11)
x = torch.randn(B, dims[0], generator=gen).cuda()
nz = torch.randn(B, dims[0], generator=gen).cuda()
t = torch.randint(0, 1000, (B,), generator=gen).cuda()
cond = torch.randn(B, dims[0], generator=gen).cuda().contiguous() if conditional else None
dev = m.model._device()
eng = m.model.engine()
handle = eng.ensure(dev)
eng.train_workspace_for(B, dev)


def fwd():
    eps = torch.ops.adcdiff.unet_train_forward(handle.value, x, t, cond, 0.1, 12345, m.model.output_dim, eng.workspace)
    loss, d_eps = torch.ops.adcdiff.mse_loss_grad(eps, nz, 1.0)
    torch.cuda.synchronize()
    return eps.clone(), d_eps.clone()


e0, d0 = fwd()
bad_e = bad_d = 0
worst = 0.0
for r in range(reps):
    e, d = fwd()
    if not torch.equal(e, e0):
        bad_e += 1
        worst = max(worst, float((e - e0).abs().max()))
        if bad_e <= 3:
            diff = (e != e0)
            rows = diff.any(1).nonzero().flatten()
            print("  repetition %d: %d elements differ in %d rows (first rows %s), max |diff| %.3e" %
                  (r, int(diff.sum()), rows.numel(), rows[:8].tolist(), float((e - e0).abs().max())), flush=True)
    if not torch.equal(d, d0):
        bad_d += 1
print("batch %d conditional %s: %d repetitions, predicted noise differed %d times (max |diff| %.3e), loss gradient %d times" %
      (B, conditional, reps, bad_e, worst, bad_d), flush=True)
