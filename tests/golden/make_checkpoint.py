#!/usr/bin/env python3
"""Checkpoint fixtures written by the reference's own ``save_model`` / ``load_model``.

Run in the build container only (it needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_checkpoint.py

Two parts, both with the reference ``DiffusionModel`` imported unmodified from /root/reference, synthetic weights from
``oracle.weights.make_weights``, dropout probability 0 and injected counter-RNG draws (so everything is deterministic):

1. ``reference_checkpoint_h16.pth`` -- the FILE the reference writes (diffusion_model.py:521-532: model_state_dict,
   optimizer_state_dict, config) for a 16-wide model after ONE ``train_step`` (non-empty AdamW state).  16 is the
   smallest width that keeps the file small enough to commit (2 MB; at the 128-wide test config it is 132 MB).  The
   kernels do not run at that width, but the checkpoint format does not depend on it: the CPU test loads this file
   into the drop-in class.  The script also lets the REFERENCE load a checkpoint written by the drop-in class and
   records that every tensor came back identical (``tiny_ckpt.npz: ours_loaded_by_reference``).
2. At the 128-wide test config the reference trains one step, saves, a fresh reference model loads the file (kept in
   /tmp only) and the script records in ``tiny_ckpt.npz`` the restored model's forward, and the loss and a few weights
   of a SECOND ``train_step`` -- which exercises the restored AdamW moments and step count.  The GPU test replays the
   chain (step, save, load into a fresh model, step) with the drop-in class and compares.

Only numbers and the small file the reference wrote are stored; no reference source text.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import NOISE_SEED, TINY, WEIGHT_SEED, build_reference, compact, injected_noise  # noqa: E402  (imports the reference)
from oracle import philox  # noqa: E402
from oracle.weights import make_weights  # noqa: E402

CKPT_SMALL = os.path.join(HERE, "reference_checkpoint_h16.pth")
SMALL = dict(input_dim=16, hidden_dim=16, output_dim=16)
PROBE_KEYS = ("output_layer.weight", "middle.0.linear1.weight", "encoder.0.norm1.weight", "time_embed.linear1.bias",
              "decoder.2.time_proj.1.weight", "encoder.3.residual_proj.bias")


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def draws(step, B, D):
    x0 = philox.normal(NOISE_SEED, 52 + step, 0, 0, B, D)
    t = (philox.uniform_u32(NOISE_SEED, philox.STREAM_TDRAW, step, 0, B, 4)[:, 0] % 1000).astype(np.int64)
    nz = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, step, 0, B, D)
    return x0, t, nz


def small_file(out):
    B, D = 8, SMALL["input_dim"]
    x0, t, nz = draws(0, B, D)
    model, _ = build_reference(SMALL)
    no_dropout(model)
    with injected_noise([nz], randint_values=[t]):
        model.train_step(torch.from_numpy(x0), 0)
    model.save_model(CKPT_SMALL)
    ck = torch.load(CKPT_SMALL, map_location="cpu", weights_only=True)      # must not need unpickling of arbitrary objects
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "config"}, sorted(ck)
    st = ck["optimizer_state_dict"]["state"]
    out["h16.n_tensors"] = np.int64(len(ck["model_state_dict"]))
    out["h16.opt_n_state"] = np.int64(len(st))
    out["h16.opt_step"] = np.float64(float(st[0]["step"]))
    out["h16.param_checksum"] = np.float64(sum(float(v.double().sum()) for v in ck["model_state_dict"].values()))
    out["h16.exp_avg_checksum"] = np.float64(sum(float(s["exp_avg"].double().sum()) for s in st.values()))

    # the other direction: a checkpoint written by the drop-in class, read by the reference
    import adcdiff
    ours = adcdiff.DiffusionModel(dict(SMALL, strict_shapes=False, memory_monitor_enabled=False, use_mixed_precision=False))
    ours.model.load_state_dict(ck["model_state_dict"])
    ours.optimizer.load_state_dict(ck["optimizer_state_dict"])
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "ours.pth")
        ours.save_model(path)
        back, _ = build_reference(SMALL)
        back.load_model(path)
    same = all(torch.equal(back.model.state_dict()[k], ck["model_state_dict"][k]) for k in ck["model_state_dict"])
    bst = back.optimizer.state_dict()["state"]
    same = same and len(bst) == len(st) and all(torch.equal(bst[i]["exp_avg"], st[i]["exp_avg"]) and
                                                torch.equal(bst[i]["exp_avg_sq"], st[i]["exp_avg_sq"]) and
                                                float(bst[i]["step"]) == float(st[i]["step"]) for i in st)
    assert same, "the reference did not restore a checkpoint written by the drop-in class"
    out["ours_loaded_by_reference"] = np.bool_(same)
    print("small checkpoint:", os.path.getsize(CKPT_SMALL), "bytes")


def tiny_chain(out):
    B, D = 8, TINY["input_dim"]
    x0, t, nz = draws(0, B, D)
    model, _ = build_reference(TINY)
    no_dropout(model)
    with injected_noise([nz], randint_values=[t]):
        res = model.train_step(torch.from_numpy(x0), 0)
    out["step1_x0"], out["step1_t"] = x0, t
    out["step1_loss"] = np.float32(res["loss"])
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "tiny.pth")
        model.save_model(path)
        ck = torch.load(path, map_location="cpu", weights_only=True)
        restored, _ = build_reference(dict(TINY))
        # fresh (different) weights first, so that only the file can make the numbers below come out right
        restored.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in
                                        make_weights(WEIGHT_SEED + 1, TINY["input_dim"], TINY["hidden_dim"], TINY["output_dim"]).items()})
        no_dropout(restored)
        restored.load_model(path)
    st = ck["optimizer_state_dict"]["state"]
    out["opt_n_state"] = np.int64(len(st))
    out["opt_step"] = np.float64(float(st[0]["step"]))
    out["opt_exp_avg_norm"] = np.float64(np.sqrt(sum(float((s["exp_avg"].double() ** 2).sum()) for s in st.values())))
    out["opt_exp_avg_sq_norm"] = np.float64(np.sqrt(sum(float((s["exp_avg_sq"].double() ** 2).sum()) for s in st.values())))
    restored.eval()
    x = philox.normal(NOISE_SEED, 50, 0, 0, B, D)
    tt = np.array([999, 0, 1, 500, 250, 750, 17, 998], dtype=np.int64)
    with torch.no_grad():
        out["fwd_t"] = tt
        out["fwd_restored"] = restored.model(torch.from_numpy(x), torch.from_numpy(tt)).numpy()
    x1, t1, nz1 = draws(1, B, D)
    out["step2_x0"], out["step2_t"] = x1, t1
    with injected_noise([nz1], randint_values=[t1]):
        res = restored.train_step(torch.from_numpy(x1), 1)
    out["step2_loss"] = np.float32(res["loss"])
    sd = restored.model.state_dict()
    for k in PROBE_KEYS:
        compact(out, "step2_after." + k, sd[k].numpy().copy())
        compact(out, "loaded." + k, ck["model_state_dict"][k].numpy().copy())
    print("tiny chain: step-1 loss", float(out["step1_loss"]), "step-2 loss", float(out["step2_loss"]))


def main():
    out = {}
    small_file(out)
    tiny_chain(out)
    np.savez_compressed(os.path.join(HERE, "tiny_ckpt.npz"), **out)


if __name__ == "__main__":
    main()
