#!/usr/bin/env python3
"""Training captures from the reference itself at the BASELINE sizes, and the conditional loss.

Run in the build container only (needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_train.py

Same method as make_golden.py (whose helpers it uses): the reference's ``src/models/diffusion_model.py`` is imported
unmodified, synthetic weights are loaded with ``load_state_dict``, noise and timesteps are injected in call order.
Stored (numbers the reference computed, nothing of its source):

* ``train_a.npz`` / ``train_b.npz`` -- emb 512 / hidden 256 and hidden 128, batch 32, dropout off: ``p_losses`` value,
  strided samples + norms of 16 gradient tensors and the global gradient norm; one ``train_step`` (loss and the weights
  of four layers after the AdamW step); the conditional loss ``p_losses(x, t, condition)`` with the gradients of
  ``condition_embed`` and of a few layers the condition reaches through ``time_proj``.
* ``tiny_cond.npz`` -- the conditional loss and gradients at 128 / 128 / 128, batch 8, tensors stored whole.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import CONFIG_A, CONFIG_B, NOISE_SEED, TINY, build_reference, compact, injected_noise, philox  # noqa: E402

GRAD_KEYS = ("output_layer.weight", "output_layer.bias", "output_norm.weight", "decoder.3.linear2.weight",
             "decoder.0.linear1.weight", "decoder.0.residual_proj.weight", "decoder.1.norm1.weight", "middle.1.linear2.bias",
             "middle.0.norm1.weight", "middle.0.norm1.bias", "middle.1.linear1.weight", "encoder.0.linear1.weight",
             "encoder.3.time_proj.1.weight", "encoder.3.linear2.weight", "time_embed.linear1.weight", "time_embed.linear2.bias",
             "encoder.1.norm2.bias")
COND_KEYS = ("condition_embed.0.weight", "condition_embed.0.bias", "condition_embed.2.weight", "condition_embed.2.bias",
             "time_embed.linear2.weight", "time_embed.linear1.bias", "encoder.0.time_proj.1.weight", "middle.0.time_proj.1.weight",
             "decoder.3.time_proj.1.bias", "decoder.0.linear1.weight", "output_layer.weight", "encoder.2.norm2.weight")
STEP_KEYS = ("output_layer.weight", "middle.0.linear1.weight", "encoder.0.norm1.weight", "time_embed.linear1.bias",
             "decoder.2.time_proj.1.weight", "encoder.3.linear2.weight")


def inputs(B, D):
    x0 = philox.normal(NOISE_SEED, 52, 0, 0, B, D)
    tq = (philox.uniform_u32(NOISE_SEED, philox.STREAM_TDRAW, 0, 0, B, 4)[:, 0] % 1000).astype(np.int64)
    nz = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 0, 0, B, D)
    cond = philox.normal(NOISE_SEED, 51, 0, 0, B, D)
    return x0, tq, nz, cond


def loss_and_grads(dims, x0, tq, nz, cond):
    model, _ = build_reference(dims)     # eval(): dropout does not enter the gradient capture
    with injected_noise([nz]):
        loss = model.p_losses(torch.from_numpy(x0), torch.from_numpy(tq), None if cond is None else torch.from_numpy(cond))
    loss.backward()
    grads = {k: p.grad.numpy().copy() for k, p in model.model.named_parameters() if p.grad is not None}
    gnorm = np.float32(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))
    return np.float32(loss.item()), grads, gnorm


def capture(dims, B, whole=False):
    out = {}
    D = dims["input_dim"]
    x0, tq, nz, cond = inputs(B, D)
    out["x0"], out["t"], out["cond"] = x0, tq, cond          # noise = philox.normal(NOISE_SEED, STREAM_QNOISE, 0, 0, B, D)
    limit = 1 << 30 if whole else 4096
    loss, grads, gnorm = loss_and_grads(dims, x0, tq, nz, None)
    out["p_losses"], out["grad_global_norm"] = loss, gnorm
    for k in GRAD_KEYS:
        compact(out, "grad." + k, grads[k], limit)
    assert all(k.startswith("condition_embed") for k in set(n for n, _ in build_reference(dims)[0].model.named_parameters()) - set(grads))
    loss, grads, gnorm = loss_and_grads(dims, x0, tq, nz, cond)
    out["cond.p_losses"], out["cond.grad_global_norm"] = loss, gnorm
    for k in COND_KEYS:
        compact(out, "cond.grad." + k, grads[k], limit)
    # one optimiser step: train_step calls .train(), so dropout is switched off by its probability
    model, _ = build_reference(dims, max_batch_size=B)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    with injected_noise([nz], randint_values=[tq]):
        res = model.train_step(torch.from_numpy(x0), 0)
    out["train_step_loss"] = np.float32(res["loss"])
    sd = model.model.state_dict()
    for k in STEP_KEYS:
        compact(out, "train_step_after." + k, sd[k].numpy().copy(), limit)
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "tiny_cond.npz"), **capture(TINY, 8, whole=False))
    print("tiny_cond done")
    np.savez_compressed(os.path.join(HERE, "train_a.npz"), **capture(CONFIG_A, 32))
    print("train_a done")
    np.savez_compressed(os.path.join(HERE, "train_b.npz"), **capture(CONFIG_B, 32))
    print("train_b done")


if __name__ == "__main__":
    main()
