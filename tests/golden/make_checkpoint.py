#!/usr/bin/env python3
"""Write a checkpoint with the reference's own ``save_model`` and record what the reference does with it.

Run in the build container only (it needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_checkpoint.py

The reference ``DiffusionModel`` (imported unmodified from /root/reference) is built at the tiny test
widths, loaded with ``oracle.weights.make_weights``, trained for ONE ``train_step`` with injected
counter-RNG noise and dropout probability 0 (so the optimizer state is non-empty and deterministic)
and saved with ``save_model`` -> ``tiny_reference_checkpoint.pth`` (the dict of diffusion_model.py:521-532:
model_state_dict, optimizer_state_dict, config).  A second reference model then ``load_model``s that file
and the script records, in ``tiny_ckpt.npz``:

* the denoiser forward of the restored model on a fixed input (eval mode),
* the loss and a few post-step weights of a SECOND ``train_step`` on the restored model (so the restored
  AdamW moments and step count are exercised),
* the optimizer state's step count and the norms of its moment tensors.

Only numbers and the file the reference wrote are stored; no reference source text.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from make_golden import NOISE_SEED, TINY, build_reference, compact, injected_noise  # noqa: E402  (imports the reference)
from oracle import philox  # noqa: E402

CKPT = os.path.join(HERE, "tiny_reference_checkpoint.pth")
PROBE_KEYS = ("output_layer.weight", "middle.0.linear1.weight", "encoder.0.norm1.weight", "time_embed.linear1.bias",
              "decoder.2.time_proj.1.weight", "encoder.3.residual_proj.bias")


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0


def main():
    out = {}
    B, D = 8, TINY["input_dim"]
    x0 = philox.normal(NOISE_SEED, 52, 0, 0, B, D)
    tq = (philox.uniform_u32(NOISE_SEED, philox.STREAM_TDRAW, 0, 0, B, 4)[:, 0] % 1000).astype(np.int64)
    nz = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 0, 0, B, D)

    model, _ = build_reference(TINY)
    no_dropout(model)
    with injected_noise([nz], randint_values=[tq]):
        model.train_step(torch.from_numpy(x0), 0)
    model.save_model(CKPT)

    # the file must be readable without unpickling arbitrary objects
    ck = torch.load(CKPT, map_location="cpu", weights_only=True)
    assert set(ck) == {"model_state_dict", "optimizer_state_dict", "config"}, sorted(ck)
    out["ckpt_keys"] = np.array(sorted(ck["model_state_dict"].keys()))
    st = ck["optimizer_state_dict"]["state"]
    out["opt_n_state"] = np.int64(len(st))
    out["opt_step"] = np.float64(float(st[0]["step"]))
    out["opt_exp_avg_norm"] = np.float64(np.sqrt(sum(float((s["exp_avg"].double() ** 2).sum()) for s in st.values())))
    out["opt_exp_avg_sq_norm"] = np.float64(np.sqrt(sum(float((s["exp_avg_sq"].double() ** 2).sum()) for s in st.values())))

    restored, _ = build_reference(TINY)      # fresh weights, then overwritten by the file
    no_dropout(restored)
    restored.load_model(CKPT)
    restored.eval()
    x = philox.normal(NOISE_SEED, 50, 0, 0, B, D)
    t = np.array([999, 0, 1, 500, 250, 750, 17, 998], dtype=np.int64)
    with torch.no_grad():
        out["fwd_t"] = t
        out["fwd_restored"] = restored.model(torch.from_numpy(x), torch.from_numpy(t)).numpy()

    # a second step on the restored model: draws of its own
    x1 = philox.normal(NOISE_SEED, 53, 0, 0, B, D)
    tq1 = (philox.uniform_u32(NOISE_SEED, philox.STREAM_TDRAW, 1, 0, B, 4)[:, 0] % 1000).astype(np.int64)
    nz1 = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 1, 0, B, D)
    out["step2_x0"], out["step2_t"] = x1, tq1
    with injected_noise([nz1], randint_values=[tq1]):
        res = restored.train_step(torch.from_numpy(x1), 1)
    out["step2_loss"] = np.float32(res["loss"])
    sd = restored.model.state_dict()
    for k in PROBE_KEYS:
        compact(out, "step2_after." + k, sd[k].numpy().copy())
        compact(out, "loaded." + k, ck["model_state_dict"][k].numpy().copy())
    np.savez_compressed(os.path.join(HERE, "tiny_ckpt.npz"), **out)
    print("checkpoint:", os.path.getsize(CKPT), "bytes; step-2 loss", float(out["step2_loss"]))


if __name__ == "__main__":
    main()
