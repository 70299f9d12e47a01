#!/usr/bin/env python3
"""Capture golden vectors from the reference implementation itself.

Run in the build container only (it needs the read-only reference checkout):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports ``/root/reference/src/models/diffusion_model.py`` unmodified, loads synthetic weights
from ``oracle.weights.make_weights`` with ``load_state_dict``, puts the model in ``eval()`` (the
reference never does; dropout would otherwise make sampling non-deterministic), replaces
``torch.randn`` / ``torch.randn_like`` / ``torch.randint`` for the duration of each call by
functions that hand out the counter-RNG noise of ``oracle.philox`` in call order, and stores
inputs that cannot be regenerated plus the reference's outputs as small ``.npz`` files next to
this script.  Nothing from the reference's source is stored, only numbers it computed.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle import philox  # noqa: E402
from oracle.weights import make_weights  # noqa: E402
from src.models.diffusion_model import DiffusionModel  # noqa: E402  (the reference)

torch.set_num_threads(8)

TINY = dict(input_dim=128, hidden_dim=128, output_dim=128)  # smallest widths whose GroupNorm groups hold >= 4 channels
CONFIG_A = dict(input_dim=512, hidden_dim=256, output_dim=512)
CONFIG_B = dict(input_dim=512, hidden_dim=128, output_dim=512)
WEIGHT_SEED = 1234
NOISE_SEED = 2025


def build_reference(dims, num_timesteps=1000, **extra):
    cfg = dict(dims, num_timesteps=num_timesteps, beta_start=1e-4, beta_end=2e-2,
               memory_monitor_enabled=False, use_mixed_precision=False, **extra)
    model = DiffusionModel(cfg)
    w = make_weights(WEIGHT_SEED, dims["input_dim"], dims["hidden_dim"], dims["output_dim"])
    model.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    model.eval()
    return model, w


@contextlib.contextmanager
def injected_noise(draws, randint_values=None):
    """draws: list of float32 arrays handed to successive torch.randn / randn_like calls."""
    queue = list(draws)
    ints = list(randint_values or [])
    real = (torch.randn, torch.randn_like, torch.randint)

    def take(shape):
        arr = queue.pop(0)
        assert tuple(arr.shape) == tuple(shape), (arr.shape, shape)
        return torch.from_numpy(arr.copy())

    def fake_randn(*size, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        return take(size)

    def fake_randn_like(x, **kw):
        return take(tuple(x.shape))

    def fake_randint(low, high, size, **kw):
        arr = ints.pop(0)
        assert tuple(arr.shape) == tuple(size)
        return torch.from_numpy(arr.astype(np.int64))

    torch.randn, torch.randn_like, torch.randint = fake_randn, fake_randn_like, fake_randint
    try:
        yield
    finally:
        torch.randn, torch.randn_like, torch.randint = real
    assert not queue, "reference drew fewer noise tensors than expected"


def loop_draws(seed, b, d, T):
    """Draw order of p_sample_loop in eval mode: x_T, then one randn_like per i = T-1 .. 1."""
    draws = [philox.normal(seed, philox.STREAM_XT, 0, 0, b, d)]
    draws += [philox.normal(seed, philox.STREAM_STEP, i, 0, b, d) for i in range(T - 1, 0, -1)]
    return draws


def probes(x, n=64):
    """Compact summary of a [B, D] tensor: n fixed columns per row, row L2 norms, checksum."""
    d = x.shape[1]
    cols = (np.arange(n) * 7919) % d
    return dict(cols=cols.astype(np.int32), values=x[:, cols].astype(np.float32),
                row_norm=np.sqrt((x.astype(np.float64) ** 2).sum(1)).astype(np.float32),
                checksum=np.float64(x.astype(np.float64).sum()))


def compact(out, key, arr, limit=4096):
    """Store small tensors whole, large ones as an evenly strided sample plus their L2 norm."""
    flat = np.ascontiguousarray(arr, dtype=np.float32).ravel()
    if flat.size <= limit:
        out[key] = arr.astype(np.float32)
        return
    stride = (flat.size + limit - 1) // limit
    out[key + ".stride"] = np.int64(stride)
    out[key + ".sample"] = flat[::stride].copy()
    out[key + ".norm"] = np.float32(np.sqrt((flat.astype(np.float64) ** 2).sum()))


def tiny_fixtures():
    out = {}
    model, w = build_reference(TINY)
    B, D = 8, TINY["input_dim"]
    with torch.no_grad():
        # schedule tables, exact
        out["beta"] = model.beta_schedule.numpy()
        out["alpha"] = model.alpha.numpy()
        out["alpha_cumprod"] = model.alpha_cumprod.numpy()
        out["alpha_cumprod_prev"] = model.alpha_cumprod_prev.numpy()
        out["posterior_variance"] = model._posterior_variance().numpy()
        out["sqrt_recip_alpha"] = torch.sqrt(1.0 / model.alpha).numpy()
        out["sqrt_one_minus_alpha_cumprod"] = torch.sqrt(1.0 - model.alpha_cumprod).numpy()
        out["ddim_timesteps"] = model._make_ddim_timesteps(50).numpy()

        # sinusoid frequency tables exactly as the reference computes them (diffusion_model.py:812-814)
        import math
        for hidden in (64, 128, 256):
            half = hidden // 2
            out["sinusoid_freq_h%d" % hidden] = torch.exp(
                torch.arange(half) * -(math.log(10000) / (half - 1))).numpy()

        # time embedding
        t4 = torch.tensor([0, 1, 500, 999], dtype=torch.long)
        out["time_embed_t"] = t4.numpy()
        out["time_embed"] = model.model.time_embed(t4).numpy()

        # denoiser forward, with per-block outputs
        x = philox.normal(NOISE_SEED, 50, 0, 0, B, D)
        cond = philox.normal(NOISE_SEED, 51, 0, 0, B, D)
        t = np.array([999, 0, 1, 500, 250, 750, 17, 998], dtype=np.int64)
        out["fwd_t"] = t
        acts = {}
        hooks = []
        for name, mod in model.model.named_modules():
            if name.startswith(("encoder.", "middle.", "decoder.")) and name.count(".") == 1:
                hooks.append(mod.register_forward_hook(
                    lambda m, i, o, name=name: acts.__setitem__(name, o.numpy().copy())))
        out["fwd_uncond"] = model.model(torch.from_numpy(x), torch.from_numpy(t)).numpy()
        for k, v in acts.items():
            out["fwd_uncond_block." + k] = v
        for h in hooks:
            h.remove()
        out["fwd_cond"] = model.model(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(cond)).numpy()

        # q_sample with per-row t
        noise = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 0, 0, B, D)
        out["q_sample"] = model.q_sample(torch.from_numpy(x), torch.from_numpy(t), torch.from_numpy(noise)).numpy()

        # single reverse steps
        for i in (999, 1, 0):
            tt = torch.full((B,), i, dtype=torch.long)
            nz = philox.normal(NOISE_SEED, philox.STREAM_STEP, i, 0, B, D)
            with injected_noise([nz] if i > 0 else []):
                out["p_sample_t%d" % i] = model.p_sample(torch.from_numpy(x), tt).numpy()
        tt = torch.full((B,), 500, dtype=torch.long)
        nz = philox.normal(NOISE_SEED, philox.STREAM_STEP, 500, 0, B, D)
        with injected_noise([nz]):
            out["p_sample_cond_t500"] = model.p_sample(torch.from_numpy(x), tt, torch.from_numpy(cond)).numpy()

        # full T=1000 reverse loop through sample(use_ddim=False), unconditional and with condition
        with injected_noise(loop_draws(NOISE_SEED, B, D, 1000)):
            out["loop_T1000"] = model.sample(B, use_ddim=False).numpy()
        with injected_noise(loop_draws(NOISE_SEED + 1, B, D, 1000)):
            out["loop_T1000_cond"] = model.sample(B, condition=torch.from_numpy(cond), use_ddim=False).numpy()

        # DDIM, the generator's default (50 steps, eta = 0), and a stochastic variant
        with injected_noise([philox.normal(NOISE_SEED, philox.STREAM_XT, 0, 0, B, D)]):
            out["ddim50_eta0"] = model.sample(B, use_ddim=True, ddim_steps=50, eta=0.0).numpy()
        pairs = [(int(a), int(b)) for a, b in zip(model._make_ddim_timesteps(50).flip(0),
                                                  torch.cat([torch.tensor([0]), model._make_ddim_timesteps(50)[:-1]]).flip(0))]
        draws = [philox.normal(NOISE_SEED, philox.STREAM_XT, 0, 0, B, D)]
        draws += [philox.normal(NOISE_SEED, philox.STREAM_STEP, tcur, 0, B, D) for tcur, _ in pairs]
        with injected_noise(draws):
            out["ddim50_eta05_cond"] = model.sample(B, condition=torch.from_numpy(cond), use_ddim=True,
                                                    ddim_steps=50, eta=0.5).numpy()

    # short schedule (T = 50) loop
    model50, _ = build_reference(TINY, num_timesteps=50)
    with torch.no_grad():
        with injected_noise(loop_draws(NOISE_SEED, B, D, 50)):
            out["loop_T50"] = model50.sample(B, use_ddim=False).numpy()

    # loss, gradients and one optimiser step (fp32 branch of train_step, eval-mode forward for the
    # gradient capture so that dropout does not enter; train_step itself calls .train(), so the
    # model used for it has dropout probability set to 0 to stay deterministic)
    modelg, _ = build_reference(TINY)
    x0 = philox.normal(NOISE_SEED, 52, 0, 0, B, D)
    tq = (philox.uniform_u32(NOISE_SEED, philox.STREAM_TDRAW, 0, 0, B, 4)[:, 0] % 1000).astype(np.int64)
    nz = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 0, 0, B, D)
    out["train_x0"], out["train_t"] = x0, tq
    with injected_noise([nz]):
        loss = modelg.p_losses(torch.from_numpy(x0), torch.from_numpy(tq))
    loss.backward()
    out["p_losses"] = np.float32(loss.item())
    grads = {k: p.grad.numpy().copy() for k, p in modelg.model.named_parameters() if p.grad is not None}
    for k in ("output_layer.weight", "output_layer.bias", "output_norm.weight", "decoder.3.linear2.weight",
              "decoder.0.linear1.weight", "decoder.0.residual_proj.weight", "middle.1.linear2.bias",
              "middle.0.norm1.weight", "middle.0.norm1.bias", "encoder.0.linear1.weight",
              "encoder.3.time_proj.1.weight", "time_embed.linear1.weight", "time_embed.linear2.bias",
              "encoder.1.norm2.bias"):
        compact(out, "grad." + k, grads[k])
    out["grad_global_norm"] = np.float32(np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values())))

    modelt, _ = build_reference(TINY)
    for m in modelt.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    with injected_noise([nz], randint_values=[tq]):
        res = modelt.train_step(torch.from_numpy(x0), 0)
    out["train_step_loss"] = np.float32(res["loss"])
    sd = modelt.model.state_dict()
    for k in ("output_layer.weight", "middle.0.linear1.weight", "encoder.0.norm1.weight",
              "time_embed.linear1.bias", "decoder.2.time_proj.1.weight"):
        compact(out, "train_step_after." + k, sd[k].numpy().copy())
    return out


def big_fixtures(dims, tag):
    out = {}
    model, w = build_reference(dims)
    B, D = 32, dims["input_dim"]
    x = philox.normal(NOISE_SEED, 50, 0, 0, B, D)
    t = (np.arange(B) * 31 + 7) % 1000
    with torch.no_grad():
        y = model.model(torch.from_numpy(x), torch.from_numpy(t.astype(np.int64))).numpy()
        out["fwd_t"] = t.astype(np.int64)
        for k, v in probes(y).items():
            out["fwd." + k] = v
        y999 = model.model(torch.from_numpy(x), torch.full((B,), 999, dtype=torch.long)).numpy()
        for k, v in probes(y999).items():
            out["fwd_t999." + k] = v
        trace = {}
        with injected_noise(loop_draws(NOISE_SEED, B, D, 1000)):
            final = model.sample(B, use_ddim=False).numpy()
        for k, v in probes(final).items():
            out["loop_T1000." + k] = v
        with injected_noise([philox.normal(NOISE_SEED, philox.STREAM_XT, 0, 0, B, D)]):
            dd = model.sample(B, use_ddim=True, ddim_steps=50).numpy()
        for k, v in probes(dd).items():
            out["ddim50." + k] = v
    out["weights_checksum"] = np.float64(sum(float(v.astype(np.float64).sum()) for v in w.values()))
    out["state_dict_keys"] = np.array(list(model.model.state_dict().keys()))
    print(tag, "done; final |x| max", np.abs(final).max())
    return out


def main():
    np.savez_compressed(os.path.join(HERE, "tiny.npz"), **tiny_fixtures())
    print("tiny done")
    np.savez_compressed(os.path.join(HERE, "config_a.npz"), **big_fixtures(CONFIG_A, "config_a"))
    np.savez_compressed(os.path.join(HERE, "config_b.npz"), **big_fixtures(CONFIG_B, "config_b"))


if __name__ == "__main__":
    main()
