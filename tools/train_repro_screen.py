"""Repeats one conditional training forward + backward (bf16, dropout on, fixed seeds) and compares every weight gradient bit for bit
with the first repetition: products over the batch rows must reproduce (bias and GroupNorm gradients are summed with atomics and
may differ in the last bits).  python tools/train_repro_screen.py [repetitions] [batch] [modes, e.g. 01]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dims = (512, 256, 512)
m = adcdiff.DiffusionModel(dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype="bf16", max_batch_size=4096))
m.model.train()
gen = torch.Generator(device="cpu").manual_seed(11)
x = torch.randn(B, dims[0], generator=gen).cuda()
nz = torch.randn(B, dims[0], generator=gen).cuda()
t = torch.randint(0, 1000, (B,), generator=gen).cuda()
cond = torch.randn(B, dims[0], generator=gen).cuda()


def grads():
    m.model.zero_grad(set_to_none=True)
    torch.manual_seed(3)
    loss = m.p_losses(x, t, cond, noise=nz)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}


for mode in [int(c) for c in (sys.argv[3] if len(sys.argv) > 3 else "01")]:
    adcdiff._lib.debug_set("train_transpose", mode)
    adcdiff._lib.debug_set("train_group_dw", 0)
    loss0, first = grads()
    bad = {}
    for r in range(reps):
        loss, g = grads()
        if loss != loss0:
            bad["<loss>"] = bad.get("<loss>", 0) + 1
        for k in first:
            if k.endswith("weight") and "norm" not in k and not torch.equal(g[k], first[k]):
                bad[k] = bad.get(k, 0) + 1
    bad = {k: v for k, v in bad.items() if not k.startswith("time_embed") and k != "<loss>"}
    print("train_transpose %d: %d repetitions, %d keys differed from the first repetition at least once (beyond the loss and the time "
          "embedding, whose sums use atomics); counts: %s" % (mode, reps, len(bad), sorted(set(bad.values())) or "-"), flush=True)
adcdiff._lib.debug_set("train_transpose", 0)
adcdiff._lib.debug_set("train_group_dw", 1)
