"""Philox4x32-10 counter RNG and the N(0,1) noise definition shared with the HIP kernels.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

The reference draws its noise from torch's global mt19937 stream
(``torch.randn(shape)`` at diffusion_model.py:232, ``torch.randn_like(x)`` at :215, :176,
:362); that stream cannot be reproduced on a GPU, so both the oracle and the HIP kernels
define noise by a counter RNG instead and the golden vectors are captured with the
reference's ``torch.randn``/``randn_like`` patched to return this noise.

Algorithm: Philox4x32-10 as published in Salmon et al., "Parallel Random Numbers: As Easy
as 1, 2, 3" (SC'11) / the Random123 library, checked against that library's known-answer
vectors in ``tests/test_oracle_philox.py``.

Noise element (row r, column c) of stream ``stream`` at step ``step`` for seed ``seed``:
    counter = (c // 4, r, step, stream), key = (seed & 0xffffffff, seed >> 32)
    (w0, w1, w2, w3) = philox4x32_10(counter, key)
    pair p = (c % 4) // 2 uses (w[2p], w[2p+1]):
        u1 = ((w[2p] >> 8) + 1) * 2**-24          in (0, 1]
        u2 = (w[2p+1] >> 8) * 2**-24              in [0, 1)
        radius = sqrt(-2 ln u1)
        z = radius * cos(2 pi u2) if c % 2 == 0 else radius * sin(2 pi u2)
The oracle evaluates ln/cos/sin in float64 and rounds once to float32; the HIP kernel
uses the hardware transcendental units, so the two agree to a few float32 ulps, not
bit-for-bit (tolerance stated in tests/test_kernels_gpu.py).
"""
import numpy as np

M0 = np.uint64(0xD2511F53)
M1 = np.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)

STREAM_XT = 1      # initial latent x_T (step field = 0)
STREAM_STEP = 2    # reverse-process noise; step field = timestep i of the p_sample call
STREAM_QNOISE = 3  # forward-process noise for q_sample / p_losses; step field = train step
STREAM_TDRAW = 4   # timestep draws for training; step field = train step


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Ten Philox rounds on arrays of 32-bit words (held in uint64 for the 32x32 products)."""
    c0 = np.asarray(c0, dtype=np.uint64) & MASK32
    c1 = np.asarray(c1, dtype=np.uint64) & MASK32
    c2 = np.asarray(c2, dtype=np.uint64) & MASK32
    c3 = np.asarray(c3, dtype=np.uint64) & MASK32
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        n0 = hi1 ^ c1 ^ np.uint64(k0)
        n2 = hi0 ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, lo1, n2, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def normal(seed, stream, step, row0, nrows, ncols):
    """[nrows, ncols] float32 N(0,1) noise for global rows row0..row0+nrows-1."""
    assert ncols % 4 == 0, "noise columns come in groups of four"
    seed = int(seed)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    rows = (np.arange(nrows, dtype=np.uint64) + np.uint64(row0))[:, None]
    quads = np.arange(ncols // 4, dtype=np.uint64)[None, :]
    shape = (nrows, ncols // 4)
    w = philox4x32_10(np.broadcast_to(quads, shape), np.broadcast_to(rows, shape),
                      np.full(shape, step, dtype=np.uint64), np.full(shape, stream, dtype=np.uint64),
                      k0, k1)
    out = np.empty((nrows, ncols // 4, 4), dtype=np.float32)
    for p in range(2):
        u1 = ((w[2 * p] >> np.uint64(8)).astype(np.float64) + 1.0) * 2.0 ** -24
        u2 = (w[2 * p + 1] >> np.uint64(8)).astype(np.float64) * 2.0 ** -24
        radius = np.sqrt(-2.0 * np.log(u1))
        out[:, :, 2 * p] = (radius * np.cos(2.0 * np.pi * u2)).astype(np.float32)
        out[:, :, 2 * p + 1] = (radius * np.sin(2.0 * np.pi * u2)).astype(np.float32)
    return out.reshape(nrows, ncols)


def uniform_u32(seed, stream, step, row0, nrows, ncols):
    """[nrows, ncols] raw 32-bit words of the same counter layout (used for timestep draws)."""
    assert ncols % 4 == 0
    seed = int(seed)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    rows = (np.arange(nrows, dtype=np.uint64) + np.uint64(row0))[:, None]
    quads = np.arange(ncols // 4, dtype=np.uint64)[None, :]
    shape = (nrows, ncols // 4)
    w = philox4x32_10(np.broadcast_to(quads, shape), np.broadcast_to(rows, shape),
                      np.full(shape, step, dtype=np.uint64), np.full(shape, stream, dtype=np.uint64),
                      k0, k1)
    return np.stack(w, axis=-1).reshape(nrows, ncols).astype(np.uint32)
