"""Runs the taped training forward repeatedly and, when a repetition's workspace differs from the first one's, prints the byte ranges
that differ (the workspace is carved in launch order, so the first range is the first tensor that went wrong).
python tools/train_forward_ws_diff.py [repetitions] [batch]"""
import sys
import torch
sys.path.insert(0, ".")
import adcdiff

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dims = (512, 256, 512)
m = adcdiff.DiffusionModel(dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype="bf16", max_batch_size=4096))
m.model.train()
gen = torch.Generator(device="cpu").manual_seed(11)
x = torch.randn(B, dims[0], generator=gen).cuda()
t = torch.randint(0, 1000, (B,), generator=gen).cuda()
dev = m.model._device()
eng = m.model.engine()
handle = eng.ensure(dev)
eng.train_workspace_for(B, dev)
ws = eng.workspace
print("workspace %d bytes" % ws.numel())


def fwd():
    ws.zero_()
    eps = torch.ops.adcdiff.unet_train_forward(handle.value, x, t, None, 0.1, 12345, m.model.output_dim, ws)
    torch.cuda.synchronize()
    return ws.clone()


w0 = fwd()
shown = 0
for r in range(reps):
    w = fwd()
    if not torch.equal(w, w0):
        d = (w != w0).nonzero().flatten()
        # coalesce into ranges separated by more than 64 KiB
        starts = [int(d[0])]
        ends = []
        gaps = (d[1:] - d[:-1] > 65536).nonzero().flatten()
        for g in gaps.tolist():
            ends.append(int(d[g])); starts.append(int(d[g + 1]))
        ends.append(int(d[-1]))
        print("repetition %d: %d bytes differ in %d ranges:" % (r, d.numel(), len(starts)), ", ".join("[%d, %d]" % se for se in list(zip(starts, ends))[:12]), flush=True)
        first = d[(d >= starts[0]) & (d <= ends[0])]
        for pitch in (8192, 4096, 2048, 1024):
            cols = (first - starts[0] + (starts[0] % pitch)) % pitch
            print("    first range: %d differing bytes; modulo a pitch of %d bytes they lie in [%d, %d]" % (first.numel(), pitch, int(cols.min()), int(cols.max())))
        shown += 1
        if shown >= 4:
            break
print("done")
