"""numpy float32 restatement of the reference's DDPM path -- TEST INFRASTRUCTURE ONLY.

Follows ``/root/reference/src/models/diffusion_model.py`` (cited per function as
``ref:LINE``).  Arithmetic is float32 in the reference's operation order wherever the order
is visible in its source; matrix products go through numpy (``x @ W.T``), whose summation
order differs from torch's, so agreement with the reference is to float32 rounding, not
bit-for-bit (measured floor recorded in tests/test_oracle_golden.py).

Weights are a dict ``key -> float32 ndarray`` with the reference's ``UNetModel.state_dict()``
keys.  Noise is never drawn here: callers pass it in (``oracle.philox.normal``).
"""
import math

import numpy as np

from . import philox

F32 = np.float32


# --------------------------------------------------------------------------- schedule
def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in float32 (ref:122).

    torch's CPU kernel computes ``step = (end - start) / (steps - 1)`` in float32 and fills
    the first half as ``fma(step, i, start)`` and the second half as
    ``fma(-step, steps-1-i, end)``: one rounding per element.  The product of two float32
    values is exact in float64, so evaluating in float64 and rounding once reproduces it
    (checked bit-for-bit against torch.linspace in tests/test_oracle_golden.py).
    """
    start, end = F32(start), F32(end)
    step = np.float64(F32((end - start) / F32(steps - 1)))
    i = np.arange(steps).astype(np.float64)
    lo = np.float64(start) + step * i
    hi = np.float64(end) - step * (steps - 1 - i)
    return np.where(np.arange(steps) < steps // 2, lo, hi).astype(F32)


def make_schedule(num_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
    """Linear-beta tables (ref:57-60, 110-122) and the per-step quantities the sampler
    derives from them on every call (ref:197-201, 509-519)."""
    beta = linspace_f32(beta_start, beta_end, num_timesteps)
    alpha = (F32(1.0) - beta).astype(F32)
    # torch.cumprod on CPU keeps its running product in double and rounds each output (ref:59)
    alpha_cumprod = np.cumprod(alpha.astype(np.float64)).astype(F32)
    alpha_cumprod_prev = np.concatenate([np.ones(1, F32), alpha_cumprod[:-1]])  # ref:60
    sched = dict(beta=beta, alpha=alpha, alpha_cumprod=alpha_cumprod,
                 alpha_cumprod_prev=alpha_cumprod_prev)
    sched["sqrt_alpha_cumprod"] = np.sqrt(alpha_cumprod)                          # ref:158
    sched["sqrt_one_minus_alpha_cumprod"] = np.sqrt(F32(1.0) - alpha_cumprod)     # ref:160, 199
    sched["sqrt_recip_alpha"] = np.sqrt(F32(1.0) / alpha)                         # ref:201
    pv = beta * (F32(1.0) - alpha_cumprod_prev) / (F32(1.0) - alpha_cumprod)       # ref:515-517
    sched["posterior_variance"] = np.maximum(pv, F32(1e-20)).astype(F32)          # ref:518
    return sched


def ddim_timesteps(num_timesteps, ddim_steps):
    """(t, t_prev) pairs in execution order (ref:315-316, 325, 375-376)."""
    c = num_timesteps // ddim_steps
    ts = np.arange(ddim_steps) * c + 1
    prev = np.concatenate([[0], ts[:-1]])
    return list(zip(ts[::-1].tolist(), prev[::-1].tolist()))


# --------------------------------------------------------------------------- layers
def num_groups(dim):
    """Largest divisor of dim that is <= 32 (ref:723-730, 840-847)."""
    if dim < 4:
        return 1
    for g in range(min(32, dim), 0, -1):
        if dim % g == 0:
            return g
    return 1


def silu(x):
    """F.silu: x * sigmoid(x) (ref:787, 820, 889, 897)."""
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def linear(x, w, b):
    """nn.Linear: x @ w.T + b with w [out, in] (ref:852-853)."""
    return (x @ w.T + b).astype(F32)


def group_norm(x, gamma, beta, eps=1e-5):
    """nn.GroupNorm(num_groups(C), C) on a [B, C] input: groups are contiguous channel
    ranges, biased variance, eps inside the square root (ref:849-850, 732)."""
    b, c = x.shape
    g = num_groups(c)
    xg = x.reshape(b, g, c // g)
    mean = xg.mean(axis=2, keepdims=True, dtype=F32)
    var = ((xg - mean) ** 2).mean(axis=2, keepdims=True, dtype=F32)
    y = (xg - mean) / np.sqrt(var + F32(eps))
    return (y.reshape(b, c) * gamma + beta).astype(F32)


def sinusoid_frequencies(hidden_dim):
    """exp(arange(half) * -(ln 10000 / (half - 1))) in float32 (ref:812-814)."""
    half = hidden_dim // 2
    scale = math.log(10000) / (half - 1)
    return np.exp(np.arange(half).astype(F32) * F32(-scale)).astype(F32)


FREQ_KEY = "__sinusoid_freq__"


def time_embedding(w, t, hidden_dim):
    """TimeEmbedding.forward (ref:802-823): sinusoid (sin first), Linear, SiLU, Linear.

    The frequencies are multiplied by timesteps up to 999, so a one-ulp difference in
    ``exp`` between math libraries moves the sinusoid by up to 6e-5.  torch's vectorised
    float32 ``exp`` is not correctly rounded, so when the weight dict carries the table the
    reference computed (key ``FREQ_KEY``, stored in the golden files) it is used as is.
    """
    freq = w[FREQ_KEY] if FREQ_KEY in w else sinusoid_frequencies(hidden_dim)
    arg = (np.asarray(t).astype(F32)[:, None] * freq[None, :]).astype(F32)
    emb = np.concatenate([np.sin(arg), np.cos(arg)], axis=-1).astype(F32)
    emb = linear(emb, w["time_embed.linear1.weight"], w["time_embed.linear1.bias"])
    emb = silu(emb)
    return linear(emb, w["time_embed.linear2.weight"], w["time_embed.linear2.bias"])


def condition_embedding(w, cond):
    """condition_embed: Linear, SiLU, Linear (ref:694-698, 765)."""
    h = linear(cond, w["condition_embed.0.weight"], w["condition_embed.0.bias"])
    return linear(silu(h), w["condition_embed.2.weight"], w["condition_embed.2.bias"])


def residual_block(w, name, x, t_emb, collect=None):
    """ResidualBlock.forward in eval mode, dropout = identity (ref:875-907)."""
    if name + ".residual_proj.weight" in w:
        residual = linear(x, w[name + ".residual_proj.weight"], w[name + ".residual_proj.bias"])
    else:
        residual = x                                                              # ref:873
    h = silu(group_norm(x, w[name + ".norm1.weight"], w[name + ".norm1.bias"]))     # ref:888-889
    h = linear(h, w[name + ".linear1.weight"], w[name + ".linear1.bias"])           # ref:890
    tp = linear(silu(t_emb), w[name + ".time_proj.1.weight"], w[name + ".time_proj.1.bias"])
    h = (h + tp).astype(F32)                                                        # ref:893
    h = silu(group_norm(h, w[name + ".norm2.weight"], w[name + ".norm2.bias"]))     # ref:896-897
    h = linear(h, w[name + ".linear2.weight"], w[name + ".linear2.bias"])           # ref:899
    out = (h + residual).astype(F32)                                                # ref:907
    if collect is not None:
        collect[name] = out
    return out


def unet_forward(w, x, t, cond=None, collect=None):
    """UNetModel.forward (ref:749-789).  x [B, in] f32, t [B] int, cond [B, in] f32 or None."""
    hidden = w["time_embed.linear2.weight"].shape[0]
    t_emb = time_embedding(w, t, hidden)
    if cond is not None:
        t_emb = (t_emb + condition_embedding(w, cond)).astype(F32)                  # ref:764-766
    if collect is not None:
        collect["t_emb"] = t_emb
    skips = []
    h = x.astype(F32)
    for i in range(4):
        h = residual_block(w, "encoder.%d" % i, h, t_emb, collect)
        skips.append(h)
    for i in range(2):
        h = residual_block(w, "middle.%d" % i, h, t_emb, collect)
    for i in range(4):
        h = np.concatenate([h, skips[3 - i]], axis=-1)                              # ref:781-782
        h = residual_block(w, "decoder.%d" % i, h, t_emb, collect)
    h = silu(group_norm(h, w["output_norm.weight"], w["output_norm.bias"]))         # ref:786-787
    return linear(h, w["output_layer.weight"], w["output_layer.bias"])               # ref:788


# --------------------------------------------------------------------------- diffusion
def q_sample(sched, x_start, t, noise):
    """ref:144-163."""
    a = sched["sqrt_alpha_cumprod"][t][:, None]
    s = sched["sqrt_one_minus_alpha_cumprod"][t][:, None]
    return (a * x_start + s * noise).astype(F32)


def p_sample_update(sched, x, eps, i, noise):
    """The elementwise part of p_sample for a batch-uniform timestep i (ref:197-216)."""
    beta_t = sched["beta"][i]
    s1m = sched["sqrt_one_minus_alpha_cumprod"][i]
    sra = sched["sqrt_recip_alpha"][i]
    mean = (sra * (x - beta_t * eps / s1m)).astype(F32)                              # ref:207-209
    if i == 0:
        return mean                                                                  # ref:211-212
    sigma = np.sqrt(sched["posterior_variance"][i])                                  # ref:214, 216
    return (mean + sigma * noise).astype(F32)


def p_sample(w, sched, x, i, cond, noise):
    """ref:186-216 with t = full((B,), i)."""
    t = np.full((x.shape[0],), i, dtype=np.int64)
    eps = unet_forward(w, x, t, cond)
    return p_sample_update(sched, x, eps, i, noise)


def p_sample_loop(w, sched, x_T, cond, noise_fn, t_start=None, t_stop=0, trace=None):
    """ref:253-261.  noise_fn(i) returns the [B, D] noise added at timestep i (i >= 1)."""
    T = len(sched["beta"])
    x = x_T.astype(F32)
    first = T - 1 if t_start is None else t_start
    for i in range(first, t_stop - 1, -1):
        x = p_sample(w, sched, x, i, cond, noise_fn(i) if i > 0 else None)
        if trace is not None:
            trace(i, x)
    return x


def ddim_update(sched, x, eps, t, t_prev, eta, noise):
    """Elementwise part of ddim_step (ref:350-369)."""
    a_t = sched["alpha_cumprod"][t]
    a_p = sched["alpha_cumprod"][t_prev]
    one = F32(1.0)
    pred_x0 = ((x - np.sqrt(one - a_t) * eps) / np.sqrt(a_t)).astype(F32)            # ref:355
    e2 = F32(eta ** 2)
    dir_coef = np.sqrt(one - a_p - e2 * (one - a_p) / (one - a_t) * (one - a_t / a_p))  # ref:358
    dir_xt = (dir_coef * eps).astype(F32)
    if eta > 0:
        sigma = F32(eta) * np.sqrt((one - a_p) / (one - a_t)) * np.sqrt(one - a_t / a_p)  # ref:363
        dir_xt = (dir_xt + sigma * noise).astype(F32)                                 # ref:364
    return (np.sqrt(a_p) * pred_x0 + dir_xt).astype(F32)                              # ref:367


def ddim_sample_loop(w, sched, x_T, cond, ddim_steps=50, eta=0.0, noise_fn=None):
    """ref:306-340 for batch <= max_batch_size."""
    T = len(sched["beta"])
    x = x_T.astype(F32)
    for t, t_prev in ddim_timesteps(T, ddim_steps):
        tt = np.full((x.shape[0],), t, dtype=np.int64)
        eps = unet_forward(w, x, tt, cond)
        x = ddim_update(sched, x, eps, t, t_prev, eta, noise_fn(t) if eta > 0 else None)
    return x


def p_losses(w, sched, x_start, t, noise, cond=None):
    """ref:165-184: mse_loss(noise, model(q_sample(x0, t, noise), t)) with mean reduction."""
    x_noisy = q_sample(sched, x_start, t, noise)
    pred = unet_forward(w, x_noisy, t, cond)
    d = (noise - pred).astype(F32)
    return F32(np.mean(d.astype(np.float64) ** 2))


def philox_noise_fn(seed, row0, nrows, ncols):
    """noise_fn for p_sample_loop: the shared counter-RNG definition of per-step noise."""
    return lambda i: philox.normal(seed, philox.STREAM_STEP, i, row0, nrows, ncols)


def philox_x_T(seed, row0, nrows, ncols):
    return philox.normal(seed, philox.STREAM_XT, 0, row0, nrows, ncols)
