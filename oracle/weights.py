"""Synthetic denoiser weights that need neither torch's RNG nor the reference to regenerate.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

No trained checkpoint ships with the reference, so parity and benchmarks run on synthetic
weights.  ``make_weights`` lists the tensors of the reference's ``UNetModel.state_dict()``
in its own order (diffusion_model.py:690-733: time_embed, condition_embed, encoder.0-3,
middle.0-1, decoder.0-3, output_norm, output_layer; inside a ResidualBlock :849-873:
norm1, norm2, linear1, linear2, time_proj.1, residual_proj) and fills tensor number i from
Philox stream 100+i:

* Linear weights: U(-a, a), a = sqrt(6 / (fan_in + fan_out)) -- the range of the reference's
  ``xavier_uniform_`` init (:742);
* mode "init": biases 0, GroupNorm weight 1 / bias 0 exactly as the reference initialises
  them (:743-747);
* mode "random" (default for parity tests): biases U(-0.1, 0.1), GroupNorm weight
  1 + U(-0.2, 0.2), GroupNorm bias U(-0.1, 0.1), so that a wrong bias / affine index in a
  kernel cannot hide behind zeros.
"""
from collections import OrderedDict

import numpy as np

from .philox import uniform_u32


def block_dims(input_dim, hidden_dim):
    """(in, out) of the ten ResidualBlocks, reference order (diffusion_model.py:701-720)."""
    h = hidden_dim
    enc = [(input_dim, h), (h, 2 * h), (2 * h, 4 * h), (4 * h, 8 * h)]
    mid = [(8 * h, 8 * h), (8 * h, 8 * h)]
    dec = [(16 * h, 4 * h), (8 * h, 2 * h), (4 * h, h), (2 * h, h)]
    names = ["encoder.%d" % i for i in range(4)] + ["middle.%d" % i for i in range(2)] + \
            ["decoder.%d" % i for i in range(4)]
    return list(zip(names, enc + mid + dec))


def state_dict_spec(input_dim, hidden_dim, output_dim):
    """[(key, shape, kind)] in the reference's state_dict order."""
    h = hidden_dim
    spec = [("time_embed.linear1.weight", (4 * h, h), "w"), ("time_embed.linear1.bias", (4 * h,), "b"),
            ("time_embed.linear2.weight", (h, 4 * h), "w"), ("time_embed.linear2.bias", (h,), "b"),
            ("condition_embed.0.weight", (h, input_dim), "w"), ("condition_embed.0.bias", (h,), "b"),
            ("condition_embed.2.weight", (h, h), "w"), ("condition_embed.2.bias", (h,), "b")]
    for name, (cin, cout) in block_dims(input_dim, hidden_dim):
        spec += [(name + ".norm1.weight", (cin,), "g"), (name + ".norm1.bias", (cin,), "gb"),
                 (name + ".norm2.weight", (cout,), "g"), (name + ".norm2.bias", (cout,), "gb"),
                 (name + ".linear1.weight", (cout, cin), "w"), (name + ".linear1.bias", (cout,), "b"),
                 (name + ".linear2.weight", (cout, cout), "w"), (name + ".linear2.bias", (cout,), "b"),
                 (name + ".time_proj.1.weight", (cout, h), "w"), (name + ".time_proj.1.bias", (cout,), "b")]
        if cin != cout:
            spec += [(name + ".residual_proj.weight", (cout, cin), "w"),
                     (name + ".residual_proj.bias", (cout,), "b")]
    spec += [("output_norm.weight", (h,), "g"), ("output_norm.bias", (h,), "gb"),
             ("output_layer.weight", (output_dim, h), "w"), ("output_layer.bias", (output_dim,), "b")]
    return spec


def _uniform(seed, idx, n):
    """n floats in [-1, 1) from Philox stream 100+idx (24-bit mantissas, exact in float32)."""
    cols = 1024
    rows = (n + cols - 1) // cols
    w = uniform_u32(seed, 100 + idx, 0, 0, rows, cols).reshape(-1)[:n]
    return ((w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -23)) - np.float32(1.0)


def make_weights(seed, input_dim, hidden_dim, output_dim, mode="random"):
    """OrderedDict key -> float32 ndarray with the reference's state_dict keys and shapes."""
    out = OrderedDict()
    for idx, (key, shape, kind) in enumerate(state_dict_spec(input_dim, hidden_dim, output_dim)):
        n = int(np.prod(shape))
        u = _uniform(seed, idx, n).reshape(shape)
        if kind == "w":
            a = np.float32(np.sqrt(6.0 / (shape[0] + shape[1])))
            val = u * a
        elif kind == "b":
            val = u * np.float32(0.1) if mode == "random" else np.zeros(shape, np.float32)
        elif kind == "g":
            val = np.float32(1.0) + u * np.float32(0.2) if mode == "random" else np.ones(shape, np.float32)
        else:
            val = u * np.float32(0.1) if mode == "random" else np.zeros(shape, np.float32)
        out[key] = np.ascontiguousarray(val, dtype=np.float32)
    return out
