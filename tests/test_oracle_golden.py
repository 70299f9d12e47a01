"""Pin the numpy oracle to the reference: every golden file was written by the reference itself
(tests/golden/make_golden.py imports /root/reference/src/models/diffusion_model.py).

Measured floor between the float32 numpy restatement and the reference's float32 torch path on
the golden inputs: <= 1e-6 for one denoiser forward, <= 4e-6 on the final latents of the
T = 1000 reverse loop (relative to max(1, max|ref|)).  The gates below are 10x that floor.
"""
import numpy as np
import pytest

from conftest import rel_err
from oracle import ddpm, philox
from oracle.weights import make_weights, state_dict_spec

NOISE_SEED = 2025
FWD_TOL = 1e-5
LOOP_TOL = 4e-5


def tiny_weights(g):
    w = make_weights(1234, 128, 128, 128)
    w[ddpm.FREQ_KEY] = g["sinusoid_freq_h128"]
    return w


def test_schedule_tables_exact(golden_tiny):
    g = golden_tiny
    s = ddpm.make_schedule(1000, 1e-4, 2e-2)
    for k in ("beta", "alpha", "alpha_cumprod", "alpha_cumprod_prev", "posterior_variance"):
        assert np.array_equal(s[k], g[k]), k
    # torch's vectorised float32 sqrt is not correctly rounded on every CPU; one ulp allowed
    for k in ("sqrt_recip_alpha", "sqrt_one_minus_alpha_cumprod"):
        assert np.abs(s[k] - g[k]).max() <= 1.2e-7, k
    assert s["posterior_variance"][0] == np.float32(1e-20)
    assert [t for t, _ in ddpm.ddim_timesteps(1000, 50)][::-1] == g["ddim_timesteps"].tolist()


def test_linspace_matches_torch_bit_for_bit():
    torch = pytest.importorskip("torch")
    for steps in (50, 100, 999, 1000, 4000):
        assert np.array_equal(ddpm.linspace_f32(1e-4, 2e-2, steps), torch.linspace(1e-4, 2e-2, steps).numpy())


def test_time_embedding(golden_tiny):
    g = golden_tiny
    w = tiny_weights(g)
    te = ddpm.time_embedding(w, g["time_embed_t"], 128)
    assert rel_err(te, g["time_embed"]) < 2e-6
    # without the reference's table the library-dependent exp() ulps show up, bounded by 1e-4
    del w[ddpm.FREQ_KEY]
    assert rel_err(ddpm.time_embedding(w, g["time_embed_t"], 128), g["time_embed"]) < 1e-4


def test_denoiser_forward_and_blocks(golden_tiny):
    g = golden_tiny
    w = tiny_weights(g)
    x = philox.normal(NOISE_SEED, 50, 0, 0, 8, 128)
    cond = philox.normal(NOISE_SEED, 51, 0, 0, 8, 128)
    col = {}
    y = ddpm.unet_forward(w, x, g["fwd_t"], None, col)
    assert rel_err(y, g["fwd_uncond"]) < FWD_TOL
    for name in ["encoder.%d" % i for i in range(4)] + ["middle.0", "middle.1"] + ["decoder.%d" % i for i in range(4)]:
        assert rel_err(col[name], g["fwd_uncond_block." + name]) < FWD_TOL, name
    assert rel_err(ddpm.unet_forward(w, x, g["fwd_t"], cond), g["fwd_cond"]) < FWD_TOL


def test_q_sample_and_single_steps(golden_tiny):
    g = golden_tiny
    w = tiny_weights(g)
    s = ddpm.make_schedule()
    x = philox.normal(NOISE_SEED, 50, 0, 0, 8, 128)
    cond = philox.normal(NOISE_SEED, 51, 0, 0, 8, 128)
    nz = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 0, 0, 8, 128)
    assert rel_err(ddpm.q_sample(s, x, g["fwd_t"], nz), g["q_sample"]) < 1e-6
    for i in (999, 1, 0):
        out = ddpm.p_sample(w, s, x, i, None, philox.normal(NOISE_SEED, philox.STREAM_STEP, i, 0, 8, 128))
        assert rel_err(out, g["p_sample_t%d" % i]) < FWD_TOL, i
    out = ddpm.p_sample(w, s, x, 500, cond, philox.normal(NOISE_SEED, philox.STREAM_STEP, 500, 0, 8, 128))
    assert rel_err(out, g["p_sample_cond_t500"]) < FWD_TOL


def test_reverse_loops(golden_tiny):
    g = golden_tiny
    w = tiny_weights(g)
    s = ddpm.make_schedule()
    cond = philox.normal(NOISE_SEED, 51, 0, 0, 8, 128)
    out = ddpm.p_sample_loop(w, s, ddpm.philox_x_T(NOISE_SEED, 0, 8, 128), None,
                             ddpm.philox_noise_fn(NOISE_SEED, 0, 8, 128))
    assert rel_err(out, g["loop_T1000"]) < LOOP_TOL
    out = ddpm.p_sample_loop(w, s, ddpm.philox_x_T(NOISE_SEED + 1, 0, 8, 128), cond,
                             ddpm.philox_noise_fn(NOISE_SEED + 1, 0, 8, 128))
    assert rel_err(out, g["loop_T1000_cond"]) < LOOP_TOL
    s50 = ddpm.make_schedule(50)
    out = ddpm.p_sample_loop(w, s50, ddpm.philox_x_T(NOISE_SEED, 0, 8, 128), None,
                             ddpm.philox_noise_fn(NOISE_SEED, 0, 8, 128))
    assert rel_err(out, g["loop_T50"]) < LOOP_TOL


def test_ddim(golden_tiny):
    g = golden_tiny
    w = tiny_weights(g)
    s = ddpm.make_schedule()
    cond = philox.normal(NOISE_SEED, 51, 0, 0, 8, 128)
    out = ddpm.ddim_sample_loop(w, s, ddpm.philox_x_T(NOISE_SEED, 0, 8, 128), None, 50, 0.0)
    assert rel_err(out, g["ddim50_eta0"]) < LOOP_TOL
    out = ddpm.ddim_sample_loop(w, s, ddpm.philox_x_T(NOISE_SEED, 0, 8, 128), cond, 50, 0.5,
                                ddpm.philox_noise_fn(NOISE_SEED, 0, 8, 128))
    assert rel_err(out, g["ddim50_eta05_cond"]) < LOOP_TOL


def test_p_losses_value(golden_tiny):
    g = golden_tiny
    w = tiny_weights(g)
    s = ddpm.make_schedule()
    nz = philox.normal(NOISE_SEED, philox.STREAM_QNOISE, 0, 0, 8, 128)
    loss = ddpm.p_losses(w, s, g["train_x0"], g["train_t"], nz)
    assert abs(float(loss) - float(g["p_losses"])) < 1e-5 * float(g["p_losses"])
    # the recorded timestep draw is the counter-RNG one
    tq = (philox.uniform_u32(NOISE_SEED, philox.STREAM_TDRAW, 0, 0, 8, 4)[:, 0] % 1000).astype(np.int64)
    assert np.array_equal(tq, g["train_t"])


@pytest.mark.parametrize("name,dims", [("config_a", (512, 256, 512)), ("config_b", (512, 128, 512))])
def test_full_size_configs(name, dims, golden_tiny):
    """Config A (emb 512 / hidden 256) and B (hidden 128): forward probes on all 32 golden rows,
    T = 1000 loop on the first 4 rows (rows are independent, so a row subset is exact)."""
    from conftest import load_golden
    g = load_golden(name)
    w = make_weights(1234, *dims)
    assert [k for k, _, _ in state_dict_spec(*dims)] == g["state_dict_keys"].tolist()
    assert abs(sum(float(v.astype(np.float64).sum()) for v in w.values()) - float(g["weights_checksum"])) < 1e-6
    w[ddpm.FREQ_KEY] = golden_tiny["sinusoid_freq_h%d" % dims[1]]
    cols = g["fwd.cols"]
    x = philox.normal(NOISE_SEED, 50, 0, 0, 32, 512)
    y = ddpm.unet_forward(w, x, g["fwd_t"])
    assert rel_err(y[:, cols], g["fwd.values"]) < FWD_TOL
    assert rel_err(np.sqrt((y.astype(np.float64) ** 2).sum(1)), g["fwd.row_norm"]) < FWD_TOL
    y = ddpm.unet_forward(w, x, np.full(32, 999))
    assert rel_err(y[:, cols], g["fwd_t999.values"]) < FWD_TOL
    s = ddpm.make_schedule()
    rows = 4
    out = ddpm.p_sample_loop(w, s, ddpm.philox_x_T(NOISE_SEED, 0, rows, 512), None,
                             ddpm.philox_noise_fn(NOISE_SEED, 0, rows, 512))
    assert rel_err(out[:, cols], g["loop_T1000.values"][:rows]) < LOOP_TOL
    out = ddpm.ddim_sample_loop(w, s, ddpm.philox_x_T(NOISE_SEED, 0, rows, 512), None, 50, 0.0)
    assert rel_err(out[:, cols], g["ddim50.values"][:rows]) < LOOP_TOL


def test_latent_pairwise_restatement_is_numpy_sum():
    """The summation order written out in oracle.latent.pairwise_sum_f32 (the order the GPU kernel reproduces) is np.sum's."""
    from oracle import latent
    rng = np.random.default_rng(1)
    for n in (5, 8, 100, 128, 512, 1000, 4096):
        for _ in range(5):
            a = (rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3)).astype(np.float32)
            assert latent.pairwise_sum_f32(a).tobytes() == np.sum(a).tobytes()


def test_smiles_candidates_follow_the_reference_thresholds():
    """Host half of the latent -> SMILES heuristic (molecule_generator.py:1084-1126, :1158-1161), product code against
    the oracle's restatement on summary rows around every threshold."""
    from oracle import latent
    import adcdiff
    from adcdiff.generation.molecule_generator import MoleculeGenerator
    rows = [(m, s, f3, f5, f8) for m in (0.05, 0.1, 0.2, 0.31, 0.5, 0.6) for s in (-3.3, 0.0, 0.07, 12.9)
            for f3, f5, f8 in ((0.9, 0.0, 0.0), (0.0, 0.71, 0.9), (0.0, 0.0, 0.61), (0.8, 0.7, 0.6))]
    for r in rows:
        assert MoleculeGenerator.smiles_candidates(r) == latent.decode_candidates(r)
    assert MoleculeGenerator.smiles_candidates((0.6, 0.0, 0.0, 0.9, 0.0)) == ("c1ccc(O)cc1", "CCO")
    assert MoleculeGenerator.smiles_candidates((0.2, 1.0, 0.9, 0.0, 0.0))[0] == "CCOC(=O)O"
