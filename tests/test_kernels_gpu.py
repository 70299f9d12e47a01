"""GPU parity of the single kernels against the oracle, called through the C ABI (via adcdiff.ops).

Tolerances (relative to max(1, max|ref|) unless stated):
* float32 GEMM: 2e-6 -- v_mfma_f32_32x32x2_f32 is an exact fma chain, only the summation order differs;
* bfloat16 GEMM: compared with a float64 product of the SAME bf16-rounded operands, 1e-5; against the
  unrounded float32 operands the error is the bf16 rounding itself (~2^-9 per operand), gate 2e-2;
* GroupNorm+SiLU float32: 2e-6; bf16 output: 2^-8 relative rounding of the output;
* schedule update / DDIM / q_sample with injected noise: bit-exact (built with -ffp-contract=off);
* Philox noise: 4e-6 absolute (hardware log2 / sin / cos units against float64 libm).
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import ddpm, philox

pytestmark = pytest.mark.gpu

adcdiff = pytest.importorskip("adcdiff")
ops = adcdiff.ops
DEV = "cuda"


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV).to(dtype)


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).to(torch.float32).numpy()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,Ks", [(128, 64, [64]), (256, 128, [128, 64]), (384, 192, [64, 128, 64]), (128, 256, [512])])
def test_gemm_segments_and_epilogue(dtype, M, N, Ks):
    rng = np.random.default_rng(7)
    As = [rng.standard_normal((M, k)).astype(np.float32) for k in Ks]
    W = (rng.standard_normal((N, sum(Ks))) / np.sqrt(sum(Ks))).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    table = rng.standard_normal((10, N)).astype(np.float32)
    idx = rng.integers(0, 10, M).astype(np.int64)
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    if dtype == "bf16":
        As_r, W_r, res_r = [_bf16_round(a) for a in As], _bf16_round(W), _bf16_round(res)
    else:
        As_r, W_r, res_r = As, W, res
    prod = np.concatenate(As_r, 1).astype(np.float64) @ W_r.astype(np.float64).T
    ref = prod + bias + table[idx] + res_r
    out = ops.gemm([_t(a, td) for a in As], _t(W, td), bias=_t(bias), rowbias=_t(table), rowbias_idx=_t(idx, torch.int64),
                   residual=_t(res, td), dtype=dtype)
    assert rel_err(out.cpu().numpy(), ref) < (1e-5 if dtype == "bf16" else 2e-6)
    # SiLU epilogue, broadcast row bias, operand-typed output only, partial store
    ref2 = prod + table[3]
    ref2 = ref2 / (1 + np.exp(-ref2))
    out2 = ops.gemm([_t(a, td) for a in As], _t(W, td), rowbias=_t(table[3]), act=1, out_f32=False, out_op=True, m_valid=M - 5,
                    dtype=dtype)
    assert out2.dtype == td
    got = out2.float().cpu().numpy()
    assert rel_err(got[:M - 5], ref2[:M - 5]) < (8e-3 if dtype == "bf16" else 2e-6)


def _gn_silu_ref(v, gamma, beta, gs):
    M, N = v.shape
    vg = v.reshape(M, N // gs, gs)
    y = (vg - vg.mean(2, keepdims=True)) / np.sqrt(vg.var(2, keepdims=True) + 1e-5)
    y = y.reshape(M, N) * gamma + beta
    return y / (1 + np.exp(-y))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("N,gs_pair", [(128, (4, 128)), (256, (8, 16)), (256, (32, 64)), (64, (4, 64)), (192, (16, 32)), (384, (128, 8))])
def test_gemm_fused_groupnorm_outputs(dtype, N, gs_pair):
    """The epilogue's SiLU(GroupNorm(v)) outputs for every supported group size, next to v itself."""
    rng = np.random.default_rng(N + gs_pair[0])
    M, K = 256, 192
    A = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32) * 3
    bias = rng.standard_normal(N).astype(np.float32)
    gam = [(1 + 0.2 * rng.standard_normal(N)).astype(np.float32) for _ in range(2)]
    bet = [(0.1 * rng.standard_normal(N)).astype(np.float32) for _ in range(2)]
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    A_r, W_r = (_bf16_round(A), _bf16_round(W)) if dtype == "bf16" else (A, W)
    v = A_r.astype(np.float64) @ W_r.astype(np.float64).T + bias
    if N % 128 != 0 and max(gs_pair) > 64:
        with pytest.raises(adcdiff.AdcdiffError):
            ops.gemm([_t(A, td)], _t(W, td), bias=_t(bias), gn=[(_t(gam[0]), _t(bet[0]), max(gs_pair))], dtype=dtype)
        return
    v32, vop, n0, n1 = ops.gemm([_t(A, td)], _t(W, td), bias=_t(bias), out_op=True, m_valid=M - 3,
                                gn=[(_t(gam[0]), _t(bet[0]), gs_pair[0]), (_t(gam[1]), _t(bet[1]), gs_pair[1])], dtype=dtype)
    tol = 1e-5 if dtype == "bf16" else 2e-6
    assert rel_err(v32.cpu().numpy()[:M - 3], v[:M - 3]) < tol
    for got, g, b, gs in ((n0, gam[0], bet[0], gs_pair[0]), (n1, gam[1], bet[1], gs_pair[1])):
        ref = _gn_silu_ref(v, g, b, gs)
        assert got.dtype == td
        err = np.abs(got.float().cpu().numpy()[:M - 3] - ref[:M - 3]).max()
        # groups of 4 / 8 channels amplify the accumulators' float32 rounding by 1 / (group std); see test_groupnorm_silu
        assert err < (2.0 ** -8 * np.abs(ref).max() + 1e-5 if dtype == "bf16" else (3e-5 if gs <= 8 else 5e-6)), (gs, err)
    if dtype == "bf16":
        assert np.array_equal(vop.float().cpu().numpy()[:M - 3], _bf16_round(v32.cpu().numpy())[:M - 3])


@pytest.mark.parametrize("hint", [256, 2564])
@pytest.mark.parametrize("Ks", [[64], [128, 64, 192], [512, 1024], [64, 64], [192, 64, 64]])
def test_gemm_256_tile_kernel_matches_128_tile_kernel(Ks, hint):
    """The 256 x 256 kernels (used for the large layers; hint 256 = eight waves, 2564 = four waves with the assembly main
    loop) against float64 and against the 128-row kernel, with every epilogue output; K = 64 is the single-tile case (the
    four-wave kernel needs two tiles and hands it to the eight-wave one), the others exercise the prefetch and the K
    segments.  The two 256-tile kernels agree bit for bit: same k order, same GroupNorm sums."""
    rng = np.random.default_rng(sum(Ks))
    M, N = 768, 512
    As = [rng.standard_normal((M, k)).astype(np.float32) for k in Ks]
    W = (rng.standard_normal((N, sum(Ks))) / np.sqrt(sum(Ks))).astype(np.float32) * 2
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((M, N)).astype(np.float32)
    table = rng.standard_normal((7, N)).astype(np.float32)
    idx = rng.integers(0, 7, M).astype(np.int64)
    gam = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    bet = (0.1 * rng.standard_normal(N)).astype(np.float32)
    bf = torch.bfloat16
    kw = dict(bias=_t(bias), rowbias=_t(table), rowbias_idx=_t(idx, torch.int64), residual=_t(res, bf), out_op=True,
              gn=[(_t(gam), _t(bet), 16), (_t(gam), _t(bet), 128)], m_valid=M - 9, dtype="bf16")
    big = ops.gemm([_t(a, bf) for a in As], _t(W, bf), tile_hint=hint, **kw)
    small = ops.gemm([_t(a, bf) for a in As], _t(W, bf), tile_hint=128, **kw)
    if hint != 256:
        for b, o in zip(big, ops.gemm([_t(a, bf) for a in As], _t(W, bf), tile_hint=256, **kw)):
            assert torch.equal(b[:M - 9], o[:M - 9])
    v = np.concatenate([_bf16_round(a) for a in As], 1).astype(np.float64) @ _bf16_round(W).astype(np.float64).T + bias + table[idx] + \
        _bf16_round(res)
    assert rel_err(big[0].cpu().numpy()[:M - 9], v[:M - 9]) < 1e-5
    for b, s_ in zip(big, small):
        # same products and the same k order within a tile, but the two kernels split K across MFMA chains alike, so
        # the float32 sums agree to rounding; the bf16 outputs may differ in the last bit
        assert rel_err(b.float().cpu().numpy()[:M - 9], s_.float().cpu().numpy()[:M - 9]) < 8e-3
    ref = _gn_silu_ref(v, gam, bet, 128)
    assert np.abs(big[3].float().cpu().numpy()[:M - 9] - ref[:M - 9]).max() < 2.0 ** -8 * np.abs(ref).max() + 1e-5


@pytest.mark.parametrize("residual", [False, True])
@pytest.mark.parametrize("outputs", ["op", "gn", "op+gn", "op+gn+gn"])
def test_four_wave_kernel_specialised_epilogues_match_the_general_one(residual, outputs):
    """Launches without a per-row bias table, activation, schedule update or float32 output run the four-wave kernel in a variant
    that knows so at compile time (and whether there is a residual) and fetches the residual one batch of rows ahead; hint 2567
    forces the general variant.  Same arithmetic in the same order: every output agrees bit for bit, dead rows stay untouched."""
    rng = np.random.default_rng(17 + residual)
    M, N, Ks = 1024, 512, [256, 128]
    bf = torch.bfloat16
    As = [_t(rng.standard_normal((M, k)).astype(np.float32), bf) for k in Ks]
    W = _t((rng.standard_normal((N, sum(Ks))) / np.sqrt(sum(Ks))).astype(np.float32) * 2, bf)
    bias, row = _t(rng.standard_normal(N).astype(np.float32)), _t(rng.standard_normal(N).astype(np.float32))
    gam, bet = _t((1 + 0.2 * rng.standard_normal(N)).astype(np.float32)), _t((0.1 * rng.standard_normal(N)).astype(np.float32))
    kw = dict(bias=bias, rowbias=row, out_f32=False, out_op="op" in outputs, m_valid=M - 37, dtype="bf16",
              gn=[(gam, bet, 16), (gam, bet, 128)][:outputs.count("gn")])   # rowbias without an index: one row for the whole batch
    if residual:
        kw["residual"] = _t(rng.standard_normal((M, N)).astype(np.float32), bf)
    special = ops._raw_gemm(As, W, tile_hint=2564, **kw)
    general = ops._raw_gemm(As, W, tile_hint=2567, **kw)
    special = special if isinstance(special, (tuple, list)) else [special]
    general = general if isinstance(general, (tuple, list)) else [general]
    assert len(special) == len(general) == len(outputs.split("+"))
    for a, b in zip(special, general):
        assert torch.equal(a[:M - 37], b[:M - 37])


def test_gemm_asymmetric_layout():
    """A = I against an asymmetric W catches a transposed or permuted accumulator map."""
    M = N = K = 128
    W = (np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 0.5).astype(np.float32)
    for dtype, td in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        out = ops.gemm([_t(np.eye(M, K, dtype=np.float32), td)], _t(W, td), dtype=dtype).cpu().numpy()
        assert np.array_equal(out, _bf16_round(W).T if dtype == "bf16" else W.T)
    M = N = K = 256
    W = (np.arange(N)[:, None] * 3 + np.arange(K)[None, :] * 0.5).astype(np.float32)
    for hint in (256, 2564):
        out = ops.gemm([_t(np.eye(M, K, dtype=np.float32), torch.bfloat16)], _t(W, torch.bfloat16), dtype="bf16", tile_hint=hint).cpu().numpy()
        assert np.array_equal(out, _bf16_round(W).T)


@pytest.mark.parametrize("dtype,M,N,hint", [("bf16", 256, 256, 128), ("bf16", 512, 256, 256), ("bf16", 512, 256, 2561), ("bf16", 512, 256, 2564), ("fp32", 128, 128, 0),
                                            ("bf16", 128, 128, 0)])
def test_gemm_applies_the_schedule_update_like_the_stand_alone_kernels(dtype, M, N, hint):
    """The sampler's reverse step inside the output GEMM's epilogue: the latent it writes equals, bit for bit, what
    adcdiff_ddpm_step / adcdiff_ddim_step make of the same GEMM's stored result (injected noise and in-kernel Philox, in
    place on x), and the operand-type copy and the SiLU(GroupNorm) output are formed from that latent."""
    rng = np.random.default_rng(M + N)
    K, gs = 192, 16
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    A = _t(rng.standard_normal((M, K)).astype(np.float32), td)
    W = _t((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32), td)
    bias = _t(rng.standard_normal(N).astype(np.float32))
    gam = (1 + 0.2 * rng.standard_normal(N)).astype(np.float32)
    bet = (0.1 * rng.standard_normal(N)).astype(np.float32)
    x = _t((rng.standard_normal((M, N)) * 4).astype(np.float32))
    nz = _t(rng.standard_normal((M, N)).astype(np.float32))
    live = M - 7
    eps = ops._raw_gemm([A], W, bias=bias, m_valid=live, dtype=dtype, tile_hint=hint)
    ddpm_c, ddim_c = (1.0101, 0.02, 0.9, 0.14), (0.9, 0.436, 0.5, 0.85, 0.1)
    cases = [(1, ddpm_c, nz), (1, ddpm_c, None), (1, ddpm_c[:3] + (0.0,), None), (2, ddim_c, nz), (2, ddim_c, None), (2, ddim_c[:4] + (0.0,), None)]
    for kind, c, noise in cases:
        step = ops._raw_ddpm_step if kind == 1 else ops._raw_ddim_step
        want = step(x, eps, c, noise=noise, seed=91, step=17, row_offset=1000)
        xin = x.clone()
        got, got_op, got_gn = ops._raw_gemm([A], W, bias=bias, m_valid=live, dtype=dtype, tile_hint=hint, out_op=True,
                                            gn=[(_t(gam), _t(bet), gs)],
                                            update=dict(kind=kind, c=c, x=xin, noise=noise, seed=91, step=17, row_offset=1000))
        assert torch.equal(got[:live], want[:live]), (kind, c, noise is None)
        assert torch.equal(got_op[:live], want[:live].to(td))
        ref = _gn_silu_ref(want.cpu().numpy().astype(np.float64), gam, bet, gs)
        err = np.abs(got_gn.float().cpu().numpy()[:live] - ref[:live]).max()
        assert err < (2.0 ** -8 * np.abs(ref).max() + 1e-5 if dtype == "bf16" else 2e-5), err
        assert torch.equal(xin, x)   # the latent is only read unless out_f32 aliases it
    # in place: out_f32 = x
    lib_args = dict(kind=1, c=ddpm_c, x=x, noise=None, seed=91, step=17, row_offset=1000)
    want = ops._raw_ddpm_step(x, eps, ddpm_c, seed=91, step=17, row_offset=1000)
    got = ops._raw_gemm([A], W, bias=bias, m_valid=live, dtype=dtype, tile_hint=hint, update=lib_args, out_f32=x)
    assert got.data_ptr() == x.data_ptr() and torch.equal(x[:live], want[:live])
    with pytest.raises(adcdiff.AdcdiffError):
        ops._raw_gemm([A], W, bias=bias, dtype=dtype, act=1, update=lib_args)


@pytest.mark.parametrize("ta,M,N,K,slices", [(True, 128, 128, 64, 1), (True, 256, 384, 512, 1), (True, 128, 256, 1024, 4), (False, 128, 128, 64, 1),
                                               (False, 384, 256, 448, 1), (False, 256, 128, 768, 3)])
def test_contraction_major_gemm_equals_the_nt_kernel_on_transposed_copies(ta, M, N, K, slices):
    """adcdiff_gemm_tr reads operands as the forward pass stored them (contraction index = row) through transposed LDS
    reads; the same product through the NT kernel on explicitly transposed copies gives the same float32 bits (same MFMA,
    same k order), and both match float64.  Asymmetric operands and strided views catch a swapped or permuted layout."""
    rng = np.random.default_rng(M + N + K)
    bf = torch.bfloat16
    lda_pad, ldb_pad = 8, 16
    Bfull = _t(rng.standard_normal((K, N + ldb_pad)).astype(np.float32), bf)
    Bm = Bfull[:, :N]                                           # [K, N], row stride N + 16
    if ta:
        Afull = _t(rng.standard_normal((K, M + lda_pad)).astype(np.float32), bf)
        A = Afull[:, :M]                                        # [K, M]
        A_nt = A.t().contiguous()
    else:
        Afull = _t(rng.standard_normal((M, K + lda_pad)).astype(np.float32), bf)
        A = Afull[:, :K]                                        # [M, K]
        A_nt = A.contiguous()
    got = ops._raw_gemm_tr(A, Bm, ta, slices=slices)
    ref64 = A_nt.double().cpu().numpy() @ Bm.double().cpu().numpy()
    if slices > 1:
        assert got.shape == (slices, M, N)
        ks = K // slices
        for s_ in range(slices):
            part = ops._raw_gemm([A_nt[:, s_ * ks:(s_ + 1) * ks].contiguous()], Bm.t()[:, s_ * ks:(s_ + 1) * ks].contiguous(), dtype="bf16", tile_hint=128)
            assert torch.equal(got[s_], part), s_
        got = got.sum(0)
    else:
        nt = ops._raw_gemm([A_nt], Bm.t().contiguous(), dtype="bf16", tile_hint=128)
        assert torch.equal(got, nt)
    assert rel_err(got.cpu().numpy(), ref64) < 1e-5
    with pytest.raises(adcdiff.AdcdiffError):
        ops._raw_gemm_tr(A, Bm, ta, slices=5)


@pytest.mark.parametrize("C,G", [(128, 32), (256, 32), (512, 32), (1024, 32), (2048, 32), (4096, 32), (192, 32), (64, 32), (2048, 16)])
def test_groupnorm_silu(C, G):
    rng = np.random.default_rng(C)
    M = 37
    x = (rng.standard_normal((M, C)) * 3 + 0.5).astype(np.float32)
    gamma = (1 + 0.2 * rng.standard_normal(C)).astype(np.float32)
    beta = (0.1 * rng.standard_normal(C)).astype(np.float32)
    xg = x.astype(np.float64).reshape(M, G, C // G)
    y = (xg - xg.mean(2, keepdims=True)) / np.sqrt(xg.var(2, keepdims=True) + 1e-5)
    y = y.reshape(M, C) * gamma + beta
    ref = y / (1 + np.exp(-y))
    out = ops.groupnorm_silu(_t(x), G, _t(gamma), _t(beta), out_dtype="fp32")
    # two-channel groups amplify input rounding by 1/|x0 - x1|; every other size is well conditioned
    assert rel_err(out.cpu().numpy(), ref) < (5e-5 if C // G == 2 else 2e-6)
    outb, raw = ops.groupnorm_silu(_t(x), G, _t(gamma), _t(beta), out_dtype="bf16", with_raw=True)
    assert np.abs(outb.float().cpu().numpy() - ref).max() <= 2.0 ** -8 * np.abs(ref).max() + 1e-6
    assert np.array_equal(raw.float().cpu().numpy(), _bf16_round(x))


def test_philox_noise_matches_oracle():
    z = ops.philox_normal(64, 512, 2025, philox.STREAM_STEP, 7, row_offset=1000).cpu().numpy()
    ref = philox.normal(2025, philox.STREAM_STEP, 7, 1000, 64, 512)
    assert np.abs(z - ref).max() < 4e-6
    z2 = ops.philox_normal(3, 128, (5 << 32) + 9, philox.STREAM_XT, 0).cpu().numpy()
    assert np.abs(z2 - philox.normal((5 << 32) + 9, philox.STREAM_XT, 0, 0, 3, 128)).max() < 4e-6


def test_schedule_update_bit_exact_with_injected_noise():
    s = ddpm.make_schedule()
    B, D = 33, 512
    x = philox.normal(1, 50, 0, 0, B, D) * 5
    eps = philox.normal(1, 51, 0, 0, B, D)
    nz = philox.normal(1, philox.STREAM_STEP, 3, 0, B, D)
    for i in (999, 500, 1, 0):
        coef = (s["sqrt_recip_alpha"][i], s["beta"][i], s["sqrt_one_minus_alpha_cumprod"][i],
                np.sqrt(s["posterior_variance"][i]) if i > 0 else 0.0)
        out = ops.ddpm_step(_t(x), _t(eps), coef, noise=_t(nz)).cpu().numpy()
        assert np.array_equal(out, ddpm.p_sample_update(s, x, eps, i, nz)), i
        # in-kernel Philox noise: same update, noise equal to a few ulps
        out2 = ops.ddpm_step(_t(x), _t(eps), coef, seed=77, step=i, row_offset=12).cpu().numpy()
        ref2 = ddpm.p_sample_update(s, x, eps, i, philox.normal(77, philox.STREAM_STEP, i, 12, B, D))
        assert np.abs(out2 - ref2).max() < 4e-6 * max(1.0, float(coef[3])) + 1e-6


def test_ddim_and_q_sample_bit_exact():
    s = ddpm.make_schedule()
    B, D = 16, 256
    x = philox.normal(2, 50, 0, 0, B, D) * 3
    eps = philox.normal(2, 51, 0, 0, B, D)
    nz = philox.normal(2, philox.STREAM_STEP, 3, 0, B, D)
    one = np.float32(1)
    for (t, tp), eta in (((981, 961), 0.0), ((21, 1), 0.5), ((1, 0), 0.5)):
        a_t, a_p = s["alpha_cumprod"][t], s["alpha_cumprod"][tp]
        e2 = np.float32(eta ** 2)
        dir_coef = np.sqrt(one - a_p - e2 * (one - a_p) / (one - a_t) * (one - a_t / a_p))
        sigma = np.float32(eta) * np.sqrt((one - a_p) / (one - a_t)) * np.sqrt(one - a_t / a_p) if eta > 0 else 0.0
        coef = (np.sqrt(one - a_t), np.sqrt(a_t), np.sqrt(a_p), dir_coef, sigma)
        out = ops.ddim_step(_t(x), _t(eps), coef, noise=_t(nz) if eta > 0 else None).cpu().numpy()
        assert np.array_equal(out, ddpm.ddim_update(s, x, eps, t, tp, eta, nz)), (t, tp, eta)
    t = np.array([0, 999, 5, 500] * 4, dtype=np.int64)
    out, used = ops.q_sample(_t(x), _t(t, torch.int64), _t(s["sqrt_alpha_cumprod"]), _t(s["sqrt_one_minus_alpha_cumprod"]),
                             noise=_t(nz), return_noise=True)
    assert np.array_equal(out.cpu().numpy(), ddpm.q_sample(s, x, t, nz))
    assert np.array_equal(used.cpu().numpy(), nz)
    out, used = ops.q_sample(_t(x), _t(t, torch.int64), _t(s["sqrt_alpha_cumprod"]), _t(s["sqrt_one_minus_alpha_cumprod"]),
                             seed=9, step=4, row_offset=2, return_noise=True)
    assert np.abs(used.cpu().numpy() - philox.normal(9, philox.STREAM_QNOISE, 4, 2, B, D)).max() < 4e-6


@pytest.mark.parametrize("D", [128, 512, 1024, 4096])
def test_latent_summary_is_numpy_bit_for_bit(D):
    """(mean, sum, x[3], x[5], x[8]) per row against numpy's own float32 np.mean / np.sum (the arithmetic of
    molecule_generator.py:1080-1081, :1160): bit-exact, on rows of mixed magnitude and sign, through a strided view."""
    from oracle import latent
    rng = np.random.default_rng(D)
    B = 203
    x = (rng.standard_normal((B, D)) * 10.0 ** rng.uniform(-3, 3, size=(B, 1))).astype(np.float32)
    x[5] = 0.0
    x[6, :] = 1e-30
    xt = torch.from_numpy(np.concatenate([x, np.zeros((B, 16), np.float32)], axis=1)).cuda()[:, :D]   # row stride D + 16
    got = ops.latent_summary(xt).cpu().numpy()
    ref = latent.summaries(x)
    assert got.shape == (B, 5) and got.tobytes() == ref.tobytes()


def test_latent_summary_rejects_unsupported_width():
    with pytest.raises(Exception, match="128"):
        ops.latent_summary(torch.zeros(4, 320, device="cuda"))


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 2e-5)])
@pytest.mark.parametrize("M,N,K,S", [(128, 64, 2048, 4), (256, 512, 4096, 8), (512, 256, 32768, 32)])
def test_split_k_partials_sum_to_the_unsplit_product(dtype, tol, M, N, K, S):
    """split_k: the S partial products of the K slices, summed, equal the unsplit GEMM on the same (already rounded)
    operands up to float32 summation order (2e-5 of the largest entry: sqrt(K) roundings of 6e-8); each partial equals the GEMM of its slice bit for
    bit.  Shapes of the weight-gradient GEMMs (M x N small, K = batch)."""
    td = torch.bfloat16 if dtype == "bf16" else torch.float32
    gen = torch.Generator(device="cpu").manual_seed(K + S)
    a = torch.randn(M, K, generator=gen).cuda().to(td)
    w = (torch.randn(N, K, generator=gen) / K ** 0.5).cuda().to(td)
    full = ops.gemm([a], w, dtype=dtype)
    parts = ops.gemm([a], w, dtype=dtype, split_k=S)
    assert parts.shape == (S, M, N)
    scale = float(full.abs().max())
    assert float((parts.sum(0) - full).abs().max()) <= tol * scale
    ks = K // S
    for s_ in (0, S - 1):
        one = ops.gemm([a[:, s_ * ks:(s_ + 1) * ks]], w[:, s_ * ks:(s_ + 1) * ks], dtype=dtype)
        assert torch.equal(one, parts[s_])
    with pytest.raises(Exception, match="split_k"):
        ops.gemm([a], w, dtype=dtype, split_k=S, bias=torch.zeros(N, device="cuda"))


def test_custom_operators_through_the_dispatcher():
    """torch.ops.adcdiff.*: called directly (not through the python wrappers), results equal the raw C-ABI calls bit for bit, and
    torch.library.opcheck accepts schema, fake-tensor rule and dispatch registration of the operators."""
    rng = np.random.default_rng(3)
    x, eps = _t(rng.standard_normal((96, 128)).astype(np.float32)), _t(rng.standard_normal((96, 128)).astype(np.float32))
    coef = (1.01, 0.02, 0.7, 0.1)
    a = torch.ops.adcdiff.ddpm_step(x, eps, None, *coef, 2025, 17, 5)
    b = ops._raw_ddpm_step(x, eps, coef, seed=2025, step=17, row_offset=5)
    assert torch.equal(a, b)
    z = torch.ops.adcdiff.philox_normal(96, 128, 2025, philox.STREAM_STEP, 17, 5, torch.device(DEV))
    assert torch.equal(torch.ops.adcdiff.ddpm_step(x, eps, z, *coef, 0, 0, 0), a)      # injected noise == in-kernel noise of the same counters
    s = torch.ops.adcdiff.latent_summary(x)
    assert np.array_equal(s.cpu().numpy()[:, 1], x.cpu().numpy().sum(axis=1, dtype=np.float32))
    W = _t((rng.standard_normal((128, 128)) / 11.0).astype(np.float32))
    A = torch.zeros(128, 128, device=DEV)
    A[:96] = x
    y, = torch.ops.adcdiff.gemm([A], W, None, None, None, None, 0, True, False, [], [], [], 96, 0, 1e-5, 0, 0)
    assert rel_err(y[:96].cpu().numpy(), x.cpu().numpy() @ W.cpu().numpy().T) < 2e-6
    torch.library.opcheck(torch.ops.adcdiff.ddpm_step, (x, eps, None, *coef, 2025, 17, 5))
    torch.library.opcheck(torch.ops.adcdiff.latent_summary, (x,))
    # all 128 rows valid here: rows beyond m_valid are never written, so their (uninitialised) contents differ between two calls
    torch.library.opcheck(torch.ops.adcdiff.gemm, ([A], W, None, None, None, None, 0, True, False, [], [], [], 0, 0, 1e-5, 0, 0))


def test_matrix_core_probe_reports_a_plausible_rate_and_clock():
    """adcdiff_matrix_core_probe: bf16 matrix-core rate on register operands and the shader clock it ran at.  On an MI355X the rate
    lies between a tenth of and the data-sheet 2.5 PFLOP/s, the clock between 0.5 and 2.5 GHz, and rate / (CUs x 4 SIMDs x 1024 flop
    per cycle x clock) -- the share of cycles the matrix cores were issuing -- is at most 1."""
    from adcdiff import _lib
    tf, mhz = _lib.matrix_core_probe(2000, 8)
    assert 250.0 < tf < 2600.0 and 500.0 < mhz < 2500.0
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert tf * 1e12 / (cus * 4 * 1024 * mhz * 1e6) < 1.02
    with pytest.raises(adcdiff.AdcdiffError):
        _lib.matrix_core_probe(0, 1)
