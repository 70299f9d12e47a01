"""The entry point that allocates nothing (adcdiff_clip_adamw_step_noalloc: scratch in a file-scope device array of the
library) against the entry point it is a variant of (adcdiff_clip_adamw_step with a caller workspace), both through the C ABI
(ctypes): bit-identical parameters, moments and gradient norm on two calls in a row, the second on a smaller input, and a
clear error beyond the built-in limit.  Collected last on purpose (file name)."""
import ctypes as C
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

adcdiff = pytest.importorskip("adcdiff")
from adcdiff import _lib  # noqa: E402


def _table(params, grads, m, v, dev):
    rows = np.zeros((len(params), 6), dtype=np.int64)
    chunk = 0
    for i, (p, g, a, b) in enumerate(zip(params, grads, m, v)):
        rows[i] = (p.data_ptr(), g.data_ptr(), a.data_ptr(), b.data_ptr(), p.numel(), chunk)
        chunk += (p.numel() + _lib.OPT_CHUNK - 1) // _lib.OPT_CHUNK
    return torch.from_numpy(rows).to(dev), int(chunk)


def _run(sizes, seed, noalloc, steps=(1, 2)):
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device="cpu").manual_seed(seed)
    params = [torch.randn(n, generator=gen).to(dev) for n in sizes]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    norms = []
    for step in steps:
        grads = [(torch.randn(n, generator=gen) * (3.0 if step == 1 else 0.01)).to(dev) for n in sizes]
        table, chunks = _table(params, grads, m, v, dev)
        norm = torch.zeros(1, device=dev)
        args = _lib.AdamWArgs(1e-3, 0.9, 0.999, 1e-8, 1e-2, 1.0, step, 1)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if noalloc:
            gc.collect()   # garbage of earlier tests (models, workspaces) freed during the call would move the counter too
            before = torch.cuda.memory_allocated()
            _lib.check(lib.adcdiff_clip_adamw_step_noalloc(table.data_ptr(), len(params), chunks, C.byref(args), norm.data_ptr(), stream))
            assert torch.cuda.memory_allocated() == before
        else:
            ws = torch.empty(lib.adcdiff_clip_adamw_workspace_bytes(chunks), dtype=torch.uint8, device=dev)
            _lib.check(lib.adcdiff_clip_adamw_step(table.data_ptr(), len(params), chunks, C.byref(args), norm.data_ptr(), ws.data_ptr(),
                                                   ws.numel(), stream))
        torch.cuda.synchronize()
        norms.append(float(norm))
    return params, m, v, norms


def test_noalloc_clip_adamw_equals_workspace_version():
    big = [1, 7, 4096, 130 * 33, 1 << 20, 3 * 5 * 7, 64 * 1024 + 3]      # 275 chunks, ragged sizes
    small = [5, 4097, 1000]                                             # 4 chunks: the scratch keeps the first call's contents beyond them
    for sizes, seed in ((big, 11), (small, 12)):                        # two calls in a row on this GPU, the second smaller
        ref = _run(sizes, seed, noalloc=False)
        got = _run(sizes, seed, noalloc=True)
        assert ref[3] == got[3] and ref[3][0] > 1.0                     # gradient norms, and the clip acted on step 1
        for group_ref, group_got in zip(ref[:3], got[:3]):
            for a, b in zip(group_ref, group_got):
                assert torch.equal(a, b)
    # the state left by a full-size call must not leak into a smaller one that follows it directly
    _run(big, 13, noalloc=True, steps=(1,))
    assert _run(small, 12, noalloc=True)[3] == _run(small, 12, noalloc=False)[3]


def test_noalloc_rejects_sizes_beyond_its_scratch():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    p = torch.zeros(8, device=dev)
    table, _ = _table([p], [p], [p], [p], dev)
    args = _lib.AdamWArgs(1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0, 1, 1)
    rc = lib.adcdiff_clip_adamw_step_noalloc(table.data_ptr(), 1, _lib.OPT_NOALLOC_MAX_CHUNKS + 1, C.byref(args), None, None)
    assert rc == -4 and b"exceed the built-in scratch" in lib.adcdiff_last_error()
    assert torch.equal(p, torch.zeros_like(p))
