"""The few-row GEMM kernel that splits K over workgroups (at most 64 live rows, adcdiff_gemm_args.splitk_scratch) against the
128-row tile kernel, through the C ABI.  Its K order differs (S partial sums added in split order), so outputs are compared
within float32 summation error -- 2e-5 of the tensor's largest entry on the float32 output, one bf16 unit in the last place
(2^-7 relative to the entry, or of the tensor's scale for entries near zero) on the operand-type and SiLU(GroupNorm) outputs --
and must be reproducible bit for bit from launch to launch (fixed summation order), with the arrival counters left at zero."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

adcdiff = pytest.importorskip("adcdiff")
ops = adcdiff.ops
DEV = "cuda"
SCRATCH_BYTES = 4096 + 16 * 64 * 2048 * 4


def flat(r):
    return list(r) if isinstance(r, (tuple, list)) else [r]


def close(a, b, live):
    a, b = a[:live].float(), b[:live].float()
    if a.dtype == b.dtype and torch.equal(a, b):
        return True
    scale = float(b.abs().max())
    tol = 2e-5 * max(scale, 1.0) if a.dtype == torch.float32 and b.dtype == torch.float32 else None
    return bool(((a - b).abs() <= (tol if tol is not None else 1.0)).all())


def check(outs, refs, live, what):
    for o, r in zip(outs, refs):
        d = (o[:live].float() - r[:live].float()).abs()
        if o.dtype == torch.float32:
            assert float(d.max()) <= 2e-5 * max(1.0, float(r[:live].abs().max())), what
        else:   # bf16: the float32 values agree to ~1e-6 relative, so the roundings differ by at most one unit in the last place
            ulp = torch.maximum(r[:live].float().abs(), torch.tensor(1e-2 * float(r[:live].float().abs().max()), device=DEV)) * 2.0 ** -7
            assert bool((d <= ulp).all()), (what, float((d / ulp).max()))
            assert float((d > 0).float().mean()) < 0.02, what   # and only a few entries sit on a rounding boundary


@pytest.mark.parametrize("Ks,N,gs_pair,live", [([2048], 2048, (64, 64), 32), ([2048, 1024], 2048, (64, 64), 32), ([4096], 1024, (128, 32), 32),
                                               ([1024, 4096], 1024, (32, 32), 17), ([512], 256, (8, 8), 32), ([512, 512], 512, (16, 16), 1),
                                               ([2048], 2048, (64, 64), 64), ([1024, 4096], 1024, (128, 32), 50), ([256], 512, (16, 32), 33),
                                               ([128, 64, 192], 256, (16, 32), 32), ([2048, 2048], 512, (16, 64), 32), ([64, 64], 128, (4, 8), 32),
                                               ([512, 1024, 1024], 512, (16, 32), 32), ([256, 512, 512], 256, (8, 16), 23), ([1024, 1024], 512, (16, 16), 32)])
def test_split_k_kernel_matches_the_tile_kernel(Ks, N, gs_pair, live):
    rng = np.random.default_rng(sum(Ks) + N + live)
    torch.manual_seed(sum(Ks) + N)
    bf = torch.bfloat16
    M = 128
    As = [torch.randn(M, k, device=DEV).to(bf) for k in Ks]
    W = (2 * torch.randn(N, sum(Ks), device=DEV) / sum(Ks) ** 0.5).to(bf)
    gam, bet = torch.rand(N, device=DEV) + 0.5, 0.1 * torch.randn(N, device=DEV)
    kw = dict(bias=torch.randn(N, device=DEV), rowbias=torch.randn(7, N, device=DEV), rowbias_idx=torch.randint(0, 7, (M,), device=DEV),
              residual=torch.randn(M, N, device=DEV).to(bf), out_op=True, gn=[(gam, bet, gs_pair[0]), (gam, bet, gs_pair[1])], m_valid=live, dtype="bf16")
    scratch = torch.zeros(SCRATCH_BYTES, device=DEV, dtype=torch.uint8)
    scratch[4096:] = 0x7f                                   # stale partial sums must not matter
    ref = flat(ops._raw_gemm(As, W, tile_hint=128, **kw))
    one = flat(ops._raw_gemm(As, W, tile_hint=327, splitk_scratch=scratch, **kw))
    two = flat(ops._raw_gemm(As, W, tile_hint=327, splitk_scratch=scratch, **kw))
    dflt = flat(ops._raw_gemm(As, W, splitk_scratch=scratch, **kw))
    assert len(one) == 4
    check(one, ref, live, dict(Ks=Ks, N=N, live=live))
    # the default choice takes this kernel for the layers that stream at least 4 MB of weights through 32- or 64-column tiles
    # (and, with 32-column tiles, the narrower layers from 20 K tiles on)
    pays = (max(gs_pair) <= 64 and sum(Ks) >= 2048 and N * sum(Ks) >= 1 << 22) or (max(gs_pair) <= 32 and sum(Ks) >= 1280 and N * sum(Ks) >= 1 << 18)
    for a, b, c, r in zip(one, two, dflt, ref):
        assert torch.equal(a[:live], b[:live])              # launch to launch, bit for bit
        assert torch.equal(c[:live], a[:live] if pays else r[:live])
    assert int(scratch[:4096].view(torch.int32).abs().sum()) == 0
    # without a scratch buffer, or with the switch off, the result is the tile kernels' bit for bit
    adcdiff._lib.debug_set("gemm_splitk", 0)
    try:
        off = flat(ops._raw_gemm(As, W, splitk_scratch=scratch, **kw))
    finally:
        adcdiff._lib.debug_set("gemm_splitk", 1)
    for a, b, c in zip(off, ref, flat(ops._raw_gemm(As, W, **kw))):
        assert torch.equal(a[:live], b[:live]) and torch.equal(c[:live], b[:live])


def test_split_k_applies_the_schedule_update():
    """The output GEMM of a sampling step (reverse-step update in the epilogue, Philox noise keyed by global row) on the split-K kernel."""
    torch.manual_seed(3)
    bf = torch.bfloat16
    M, K, N, live = 128, 256, 512, 32
    A, W = torch.randn(M, K, device=DEV).to(bf), (torch.randn(N, K, device=DEV) / 16).to(bf)
    gam, bet, bias = torch.rand(N, device=DEV) + 0.5, 0.1 * torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    x = 4 * torch.randn(M, N, device=DEV)
    scratch = torch.zeros(SCRATCH_BYTES, device=DEV, dtype=torch.uint8)
    for kind, c in ((1, (1.0101, 0.02, 0.9, 0.14)), (2, (0.9, 0.436, 0.5, 0.85, 0.1))):
        outs = []
        for h, sc in ((327, scratch), (128, None)):
            outs.append(flat(ops._raw_gemm([A], W, bias=bias, m_valid=live, dtype="bf16", tile_hint=h, out_op=True, gn=[(gam, bet, 16)], splitk_scratch=sc,
                                           update=dict(kind=kind, c=c, x=x.clone(), noise=None, seed=5, step=3, row_offset=40))))
        check(outs[0], outs[1], live, kind)


def test_random_shapes_on_the_split_k_kernel():
    """120 random launches (N, one to three K segments, GroupNorm group sizes, 1..64 live rows, every epilogue input present or not)
    forced onto the split-K kernel: every output within the stated tolerance of the 128-row tile kernel, and the second of two
    identical launches equal to the first bit for bit."""
    rng = np.random.default_rng(23)
    torch.manual_seed(23)
    bf = torch.bfloat16
    scratch = torch.zeros(SCRATCH_BYTES, device=DEV, dtype=torch.uint8)
    for case in range(120):
        N = int(rng.choice([128, 256, 512, 1024, 2048]))
        Ks = [int(rng.choice([64, 128, 256, 512, 1024, 2048])) for _ in range(int(rng.integers(1, 4)))]
        if sum(Ks) < 128:
            Ks.append(128)                                   # at least two K tiles: one per split
        gss = [g for g in (4, 8, 16, 32, 64, 128) if N % g == 0]
        gs = [int(rng.choice(gss)) for _ in range(int(rng.integers(0, 3)))]
        live = int(rng.integers(1, 65))
        As = [torch.randn(128, k, device=DEV).to(bf) for k in Ks]
        W = (torch.randn(N, sum(Ks), device=DEV) / sum(Ks) ** 0.5).to(bf)
        kw = dict(dtype="bf16", m_valid=live)
        if rng.random() < 0.7:
            kw["bias"] = torch.randn(N, device=DEV)
        if rng.random() < 0.4:
            kw["rowbias"], kw["rowbias_idx"] = torch.randn(9, N, device=DEV), torch.randint(0, 9, (128,), device=DEV)
        if rng.random() < 0.5:
            kw["residual"] = torch.randn(128, N, device=DEV).to(bf)
        if rng.random() < 0.5 or not gs:
            kw["out_op"] = True
        kw["gn"] = [(torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV), g) for g in gs]
        what = dict(case=case, N=N, Ks=Ks, gs=gs, live=live)
        ref = flat(ops._raw_gemm(As, W, tile_hint=128, **kw))
        one = flat(ops._raw_gemm(As, W, tile_hint=327, splitk_scratch=scratch, **kw))
        two = flat(ops._raw_gemm(As, W, tile_hint=327, splitk_scratch=scratch, **kw))
        check(one, ref, live, what)
        for a, b in zip(one, two):
            assert torch.equal(a[:live], b[:live]), what
    assert int(scratch[:4096].view(torch.int32).abs().sum()) == 0
