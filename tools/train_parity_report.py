"""Training parity at the BASELINE sizes against the reference's captures (tests/golden/train_a.npz, train_b.npz, tiny_cond.npz):
prints, per dtype, the loss error and the worst tensor-relative gradient / post-step weight errors.  Run on the GPU box:
    python tools/train_parity_report.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import adcdiff  # noqa: E402
from conftest import load_golden  # noqa: E402
from oracle import philox  # noqa: E402
from oracle.weights import make_weights  # noqa: E402

SEED = 2025


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda").to(dtype)


def sample_of(g, key, got):
    got = got.detach().float().cpu().numpy()
    if key in g.files:
        return got.ravel(), g[key].ravel()
    return got.ravel()[::int(g[key + ".stride"])], g[key + ".sample"]


def names(g, prefix):
    """parameter names stored under `prefix` (whole, or as .sample / .stride / .norm)"""
    out = set()
    for k in g.files:
        if k.startswith(prefix):
            k = k[len(prefix):]
            for suf in (".sample", ".stride", ".norm"):
                if k.endswith(suf):
                    k = k[:-len(suf)]
            out.add(k)
    return sorted(out)


def build(dims, dtype, **extra):
    m = adcdiff.DiffusionModel(dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype=dtype, max_batch_size=4096,
                                    use_mixed_precision=(dtype == "bf16"), **extra))
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, *dims).items()})
    return m


def report(name, dims, dtype, conditional):
    g = load_golden(name)
    B, D = g["x0"].shape
    nz = cu(philox.normal(SEED, philox.STREAM_QNOISE, 0, 0, B, D))
    res = {}
    m = build(dims, dtype)
    m.model.eval()
    loss = m.p_losses(cu(g["x0"]), cu(g["t"], torch.int64), noise=nz)
    res["loss_rel"] = abs(float(loss.detach()) - float(g["p_losses"])) / float(g["p_losses"])
    loss.backward()
    grads = {k: p.grad for k, p in m.model.named_parameters() if p.grad is not None}
    worst, l2 = {}, {}
    for key in names(g, "grad."):
        got, ref = sample_of(g, "grad." + key, grads[key])
        worst[key] = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))
        l2[key] = float(np.linalg.norm(got.astype(np.float64) - ref) / max(np.linalg.norm(ref.astype(np.float64)), 1e-30))
    res["grad_worst"] = max(worst.values()); res["grad_worst_key"] = max(worst, key=worst.get)
    res["grad_l2_worst"] = max(l2.values()); res["grad_l2_worst_key"] = max(l2, key=l2.get)
    sq = sum(float((p.double() ** 2).sum()) for p in grads.values())
    res["gnorm_rel"] = abs(np.sqrt(sq) - float(g["grad_global_norm"])) / float(g["grad_global_norm"])
    if conditional:
        m.model.zero_grad(set_to_none=True)
        loss = m.p_losses(cu(g["x0"]), cu(g["t"], torch.int64), cu(g["cond"]), noise=nz)
        res["cond_loss_rel"] = abs(float(loss.detach()) - float(g["cond.p_losses"])) / float(g["cond.p_losses"])
        loss.backward()
        grads = {k: p.grad for k, p in m.model.named_parameters() if p.grad is not None}
        worst = {}
        for key in names(g, "cond.grad."):
            got, ref = sample_of(g, "cond.grad." + key, grads[key])
            worst[key] = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12))
        res["cond_grad_worst"] = max(worst.values()); res["cond_grad_worst_key"] = max(worst, key=worst.get)
        res["cond_grads"] = worst
        sq = sum(float((p.double() ** 2).sum()) for p in grads.values())
        res["cond_gnorm_rel"] = abs(np.sqrt(sq) - float(g["cond.grad_global_norm"])) / float(g["cond.grad_global_norm"])
    # one optimiser step with the recorded timesteps and noise
    m = build(dims, dtype, max_batch_size=B) if False else build(dims, dtype)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    real_randint, real_q = torch.randint, adcdiff.ops.q_sample
    torch.randint = lambda *a, **k: cu(g["t"], torch.int64) if "device" in k else real_randint(*a, **k)
    adcdiff.ops.q_sample = lambda *a, **k: real_q(*a, **{**k, "noise": nz})
    try:
        out = m.train_step(cu(g["x0"]), 0)
    finally:
        torch.randint, adcdiff.ops.q_sample = real_randint, real_q
    res["step_loss_rel"] = abs(out["loss"] - float(g["train_step_loss"])) / float(g["train_step_loss"])
    sd = m.model.state_dict()
    w0 = make_weights(1234, *dims)
    worst, moved = {}, {}
    for key in names(g, "train_step_after."):
        got, ref = sample_of(g, "train_step_after." + key, sd[key])
        d = np.abs(got - ref)
        worst[key] = (float(d.max()), float((d < 4e-6).mean()))
    res["step_after"] = worst
    return res


def main():
    out = {}
    cases = [("tiny_cond", (128, 128, 128)), ("train_a", (512, 256, 512)), ("train_b", (512, 128, 512))]
    cond_ok = "--no-cond" not in sys.argv
    for name, dims in cases:
        for dtype in ("fp32", "bf16"):
            r = report(name, dims, dtype, cond_ok)
            out["%s.%s" % (name, dtype)] = r
            print(name, dtype, json.dumps({k: v for k, v in r.items() if k != "cond_grads"}))
    paths = [a for a in sys.argv[1:] if not a.startswith("--")]
    if paths:
        json.dump(out, open(paths[0], "w"), indent=1)


if __name__ == "__main__":
    main()
