"""Known-answer tests for the counter RNG that defines the noise (oracle side)."""
import numpy as np

from oracle import philox


def _kat(counter, key):
    out = philox.philox4x32_10(*[np.array([c]) for c in counter], key[0], key[1])
    return [int(o[0]) for o in out]


def test_random123_known_answers():
    # kat_vectors of the Random123 distribution, philox4x32 with 10 rounds
    assert _kat([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _kat([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _kat([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_normal_moments_and_row_independence():
    z = philox.normal(2025, philox.STREAM_STEP, 7, 0, 2048, 512)
    assert z.dtype == np.float32 and z.shape == (2048, 512)
    assert abs(float(z.mean())) < 5e-3
    assert abs(float(z.std()) - 1.0) < 5e-3
    assert abs(float((z.astype(np.float64) ** 4).mean()) - 3.0) < 5e-2
    assert np.isfinite(z).all()
    # a shard that starts at row 100 sees exactly the rows 100.. of the full draw
    part = philox.normal(2025, philox.STREAM_STEP, 7, 100, 16, 512)
    assert np.array_equal(part, z[100:116])
    # different step / stream / seed give different noise
    assert not np.array_equal(z[:4], philox.normal(2025, philox.STREAM_STEP, 8, 0, 4, 512))
    assert not np.array_equal(z[:4], philox.normal(2025, philox.STREAM_XT, 7, 0, 4, 512))
    assert not np.array_equal(z[:4], philox.normal(2026, philox.STREAM_STEP, 7, 0, 4, 512))
