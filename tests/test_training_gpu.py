"""GPU parity of the training step (p_losses forward + backward, train_step) against numbers the reference produced
(tests/golden/tiny.npz: loss, gradients of 14 tensors, global gradient norm, weights after one AdamW step).

Tolerances: float32 path -- loss 1e-5 relative, gradients 2e-4 of the tensor's largest entry (the reference's own float32
backward differs from a float64 evaluation by ~1e-5), weights after the step 2e-6 absolute (lr = 1e-4);
bf16-operand path -- loss 2e-2, gradients 6e-2 of the largest entry.
"""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import philox
from oracle.weights import make_weights

pytestmark = pytest.mark.gpu

adcdiff = pytest.importorskip("adcdiff")
SEED = 2025
DIMS = (128, 128, 128)


def build(dtype, **extra):
    cfg = dict(input_dim=DIMS[0], hidden_dim=DIMS[1], output_dim=DIMS[2], num_timesteps=1000, beta_start=1e-4, beta_end=2e-2,
               compute_dtype=dtype, **extra)
    m = adcdiff.DiffusionModel(cfg)
    w = make_weights(1234, *DIMS)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    return m


def cu(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda").to(dtype)


def golden_tensor(g, key, got):
    """Compare with a golden tensor stored whole or as a strided sample + norm."""
    got = got.detach().float().cpu().numpy()
    if key in g:
        return got, g[key]
    stride = int(g[key + ".stride"])
    return got.ravel()[::stride], g[key + ".sample"]


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_loss_and_gradients_match_reference(dtype, golden_tiny):
    g = golden_tiny
    m = build(dtype)
    m.model.eval()   # the golden gradients were taken without dropout
    nz = cu(philox.normal(SEED, philox.STREAM_QNOISE, 0, 0, 8, 128))
    loss = m.p_losses(cu(g["train_x0"]), cu(g["train_t"], torch.int64), noise=nz)
    assert abs(float(loss.detach()) - float(g["p_losses"])) < (1e-5 if dtype == "fp32" else 2e-2) * float(g["p_losses"])
    loss.backward()
    grads = {k: p.grad for k, p in m.model.named_parameters()}
    tol = 2e-4 if dtype == "fp32" else 6e-2
    def base(k):
        for suffix in (".sample", ".stride", ".norm"):
            if k.endswith(suffix) and (k[:-len(suffix)] + ".sample") in g.files:
                return k[:-len(suffix)]
        return k
    names = sorted({base(k)[5:] for k in g.files if k.startswith("grad.")})
    assert len(names) >= 14
    for name in names:
        got, ref = golden_tensor(g, "grad." + name, grads[name])
        assert np.abs(got - ref).max() <= tol * max(np.abs(ref).max(), 1e-6), name
    sq = sum(float((p.grad.double() ** 2).sum()) for p in m.model.parameters() if p.grad is not None)
    assert abs(np.sqrt(sq) - float(g["grad_global_norm"])) < (1e-4 if dtype == "fp32" else 3e-2) * float(g["grad_global_norm"])
    assert all(p.grad is None for p in m.model.condition_embed.parameters())


def test_train_step_matches_reference(golden_tiny, monkeypatch):
    g = golden_tiny
    m = build("fp32", use_mixed_precision=False)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0   # as in the golden capture: train_step calls .train(), dropout off keeps it deterministic
    nz = philox.normal(SEED, philox.STREAM_QNOISE, 0, 0, 8, 128)
    # feed the recorded timesteps and noise through the two places train_step draws them
    real_randint = torch.randint
    monkeypatch.setattr(torch, "randint", lambda *a, **k: cu(g["train_t"], torch.int64) if "device" in k else real_randint(*a, **k))
    real_q = adcdiff.ops.q_sample
    monkeypatch.setattr(adcdiff.ops, "q_sample", lambda *a, **k: real_q(*a, **{**k, "noise": cu(nz)}))
    res = m.train_step(cu(g["train_x0"]), 0)
    assert set(res) == {"loss", "lr", "gpu_memory", "gpu_memory_percent", "system_memory_percent"}
    assert abs(res["loss"] - float(g["train_step_loss"])) < 1e-5 * float(g["train_step_loss"])
    sd = m.model.state_dict()
    w0 = make_weights(1234, *DIMS)
    for name in ("output_layer.weight", "middle.0.linear1.weight", "encoder.0.norm1.weight", "time_embed.linear1.bias",
                 "decoder.2.time_proj.1.weight"):
        got, ref = golden_tensor(g, "train_step_after." + name, sd[name])
        # AdamW's first step is lr * g / (|g| + 1e-8): entries whose gradient is within float32 noise of zero may move by a
        # different fraction of lr = 1e-4, every other entry lands within 2e-6
        diff = np.abs(got - ref)
        assert (diff < 2e-6).mean() > 0.995 and diff.max() < 5e-5, (name, float(diff.max()), float((diff < 2e-6).mean()))
        assert np.abs(got - (w0[name].ravel()[::int(g["train_step_after." + name + ".stride"])] if "train_step_after." + name not in g
                             else w0[name])).max() > 5e-5, name   # the step really moved the weights (lr = 1e-4)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_training_reduces_loss_with_dropout_and_accumulation(dtype):
    """Config-B-shaped run: loss falls over a few steps on a fixed batch with dropout active and chunking above max_batch_size."""
    torch.manual_seed(0)
    m = build(dtype, learning_rate=1e-3, max_batch_size=256, gradient_accumulation_steps=1)
    x = cu(philox.normal(7, 52, 0, 0, 512, 128))
    t_fix = torch.arange(512, device="cuda") % 1000
    nz_fix = cu(philox.normal(7, 53, 0, 0, 512, 128))
    m.model.eval()
    before = float(m.p_losses(x, t_fix, noise=nz_fix).detach())
    losses = [m.train_step(x, s)["loss"] for s in range(8)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0]
    m.model.eval()
    after = float(m.p_losses(x, t_fix, noise=nz_fix).detach())
    assert after < 0.99 * before   # same timesteps and noise: the optimiser's updates reached the packed weights the kernels read
    out = m.sample(4, use_ddim=False, seed=1, t_start=999, t_stop=990)   # the packed weights follow the optimiser
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_large_batch_gradient_is_mean_of_half_batch_gradients(dtype):
    """emb 512 / hidden 256 at batch 4096 (the per-GPU training shape): the loss is a mean over rows, so the gradient of
    the whole batch equals the mean of the gradients of its two halves on the same timesteps and noise.  Exercises the
    training workspace at full size.  Tolerance: float32 summation order only (1e-4 of the largest entry); bf16 operands
    round identically row by row, so the same bound holds with 2e-3 for the bf16-rounded dW accumulations."""
    dims = (512, 256, 512)
    cfg = dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype=dtype, max_batch_size=4096)
    m = adcdiff.DiffusionModel(cfg)
    w = make_weights(1234, *dims)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in w.items()})
    m.model.eval()
    B = 4096
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, dims[0], generator=gen).cuda()
    nz = torch.randn(B, dims[0], generator=gen).cuda()
    t = torch.randint(0, 1000, (B,), generator=gen).cuda()

    def grads(lo, hi):
        m.model.zero_grad(set_to_none=True)
        loss = m.p_losses(x[lo:hi], t[lo:hi], noise=nz[lo:hi])
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}

    l_all, g_all = grads(0, B)
    l_a, g_a = grads(0, B // 2)
    l_b, g_b = grads(B // 2, B)
    assert abs(l_all - 0.5 * (l_a + l_b)) < 1e-5 * l_all
    tol = 1e-4 if dtype == "fp32" else 2e-3
    assert len(g_all) >= 100
    for k, ga in g_all.items():
        ref = 0.5 * (g_a[k] + g_b[k])
        assert torch.isfinite(ga).all(), k
        assert float((ga - ref).abs().max()) <= tol * max(float(ref.abs().max()), 1e-6), k


@pytest.mark.parametrize("conditional", [False, True])
@pytest.mark.parametrize("dims,B", [((128, 128, 128), 64), ((512, 256, 512), 1024)])
def test_backward_without_transposed_copies_gives_the_same_gradients(dims, B, conditional, monkeypatch):
    """bf16 batches that are multiples of 64 run the backward GEMMs on the stored operands (adcdiff_gemm_tr); the NT
    kernel on transposed copies (debug switch train_transpose = 1, and every other batch size or float32) must give the same
    weight gradients bit for bit: same MFMA, same k order, same slicing of the batch contraction.  Dropout active."""
    cfg = dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype="bf16", max_batch_size=4096)
    m = adcdiff.DiffusionModel(cfg)
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, *dims).items()})
    m.model.train()
    gen = torch.Generator(device="cpu").manual_seed(11)
    x = torch.randn(B, dims[0], generator=gen).cuda()
    nz = torch.randn(B, dims[0], generator=gen).cuda()
    t = torch.randint(0, 1000, (B,), generator=gen).cuda()
    cond = torch.randn(B, dims[0], generator=gen).cuda() if conditional else None

    def grads(transpose, grouped=False):
        adcdiff._lib.debug_set("train_transpose", 1 if transpose else 0)
        adcdiff._lib.debug_set("train_group_dw", 1 if grouped else 0)
        m.model.zero_grad(set_to_none=True)
        torch.manual_seed(3)             # the dropout seed is drawn from torch: same masks in both passes
        loss = m.p_losses(x, t, cond, noise=nz)
        loss.backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}

    try:
        direct, copies, grouped = grads(False), grads(True), grads(False, grouped=True)
    finally:
        adcdiff._lib.debug_set("train_transpose", 0)
        adcdiff._lib.debug_set("train_group_dw", 1)
    # the shipped form: all weight-gradient products of the pass in one launch, none of them split over the batch -- the same
    # numbers up to the float32 order in which the batch slices were added
    assert grouped.keys() == direct.keys()
    for k in direct:
        assert float((grouped[k] - direct[k]).abs().max()) <= 2e-5 * max(float(direct[k].abs().max()), 1e-6) + (2e-2 * float(direct[k].abs().max()) if "time" in k else 0.0), k
    assert len(direct) >= 100 and direct.keys() == copies.keys()
    exact = 0
    for k in direct:
        assert torch.isfinite(direct[k]).all(), k
        if k.endswith(("linear1.weight", "linear2.weight", "residual_proj.weight", "output_layer.weight")) and not k.startswith("time_embed"):
            assert torch.equal(direct[k], copies[k]), k
            exact += 1
        elif conditional and k.endswith("weight") and ("time_proj" in k or "condition_embed" in k):
            assert torch.equal(direct[k], copies[k]), k    # with a condition these are products over the batch rows like the others
            exact += 1
        elif "time" in k and k.endswith("weight"):
            # the time branch's large products run on bf16 copies of the table gradient in the direct path, float32 in the other
            assert float((direct[k] - copies[k]).abs().max()) <= 2e-2 * max(float(copies[k].abs().max()), 1e-6), k
        else:   # bias and GroupNorm gradients are accumulated with atomics: equal to summation order
            tol = 2e-2 if "time_embed" in k else 1e-5
            assert float((direct[k] - copies[k]).abs().max()) <= tol * max(float(copies[k].abs().max()), 1e-6), k
    assert exact >= 25


@pytest.mark.parametrize("dropout", [0.0, 0.1])
@pytest.mark.parametrize("dtype,dims,B", [("fp32", (128, 128, 128), 40), ("bf16", (128, 128, 128), 64), ("bf16", (512, 256, 512), 1024)])
def test_training_forward_with_groupnorm_in_the_gemm_epilogues_matches_the_stand_alone_kernels(dtype, dims, B, dropout):
    """The training forward applies norm1 of the next block, a skip's norm1 slice, output_norm and (without dropout) norm2 in the
    epilogue of the GEMM that produces their input; debug switch train_fused_fwd = 0 runs every GroupNorm as a launch of its own
    (the path the golden gradients were first pinned on).  Same tape, same backward: loss and gradients agree to the float32
    summation order of the group statistics (a last bf16 place of an activation here and there in bf16)."""
    cfg = dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype=dtype, max_batch_size=4096)
    m = adcdiff.DiffusionModel(cfg)
    for mod in m.model.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = dropout
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, *dims).items()})
    m.model.train()
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(B, dims[0], generator=gen).cuda()
    nz = torch.randn(B, dims[0], generator=gen).cuda()
    t = torch.randint(0, 1000, (B,), generator=gen).cuda()

    def run(fused):
        adcdiff._lib.debug_set("train_fused_fwd", 1 if fused else 0)
        m.model.zero_grad(set_to_none=True)
        torch.manual_seed(3)             # the dropout seed is drawn from torch: same masks in both passes
        loss = m.p_losses(x, t, None, noise=nz)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}

    try:
        (l1, g1), (l0, g0) = run(True), run(False)
    finally:
        adcdiff._lib.debug_set("train_fused_fwd", 1)
    ltol, gtol = (1e-6, 2e-4) if dtype == "fp32" else (2e-3, 3e-2)
    assert abs(l1 - l0) <= ltol * abs(l0), (l1, l0)
    assert g1.keys() == g0.keys() and len(g0) >= 100
    worst = 0.0
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        err = float((g1[k] - g0[k]).abs().max()) / max(float(g0[k].abs().max()), 1e-6)
        worst = max(worst, err)
        assert err <= gtol, (k, err)
    print("fused training forward: loss %.6f / %.6f, worst gradient difference %.2e of the tensor's scale" % (l1, l0, worst))


@pytest.mark.parametrize("opt_cls", ["AdamW", "Adam"])
def test_fused_clip_adamw_matches_torch(opt_cls):
    """clip_grad_norm_(1.0) + optimizer.step() of torch against the three-launch version on the same tensors: ragged sizes
    (1 element, not a multiple of 4, several chunks), three steps, gradients large enough for the clip to act on step 1 and
    small enough not to on step 3.  Tolerance: the update is lr * m / (sqrt(v) + eps) with |m / sqrt(v)| <= ~1, so 1e-3
    of lr = 1e-6 absolute covers the differing fma contraction of the two implementations; moments 1e-5 relative or 1e-6
    of the tensor's largest entry."""
    from adcdiff.optim import FusedClipAdamW
    sizes = [(1,), (7,), (130, 33), (4096,), (257, 129), (3, 5, 7), (64, 1024)]
    gen = torch.Generator(device="cpu").manual_seed(3)
    init = [torch.randn(*sz, generator=gen) for sz in sizes]
    grads = [[torch.randn(*sz, generator=gen) * scale for sz in sizes] for scale in (1.0, 0.05, 0.001)]
    cls = getattr(torch.optim, opt_cls)

    def run(fused):
        params = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        opt = cls(params, lr=1e-3, weight_decay=1e-2)
        f = FusedClipAdamW(opt, max_norm=1.0) if fused else None
        norms = []
        for gs in grads:
            for p_, g_ in zip(params, gs):
                p_.grad = g_.clone().cuda()
            if fused:
                norms.append(float(f.step()))
            else:
                norms.append(float(torch.nn.utils.clip_grad_norm_(params, max_norm=1.0)))
                opt.step()
            opt.zero_grad()
        return params, opt, norms

    pa, oa, na = run(False)
    pb, ob, nb = run(True)
    assert np.allclose(na, nb, rtol=2e-6)
    assert na[0] > 1.0 > na[2]   # the clip acts on the first step only
    for a, b in zip(pa, pb):
        assert float((a.detach() - b.detach()).abs().max()) <= 1e-6
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == 3.0
        for k in ("exp_avg", "exp_avg_sq"):
            # entries that nearly cancel carry the rounding of their terms: absolute bound from the tensor's scale
            assert torch.allclose(sa[k], sb[k], rtol=1e-5, atol=1e-6 * float(sa[k].abs().max())), k
    assert all(p_._version >= 3 for p_ in pb)   # every fused step is visible to whoever caches data derived from the parameters
    # the state the fused step leaves is a torch optimizer state: it round-trips through state_dict / load_state_dict
    sd = ob.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ob.load_state_dict(sd)


def _feed_draws(monkeypatch, t, nz):
    """Route recorded timesteps and noise through the two places train_step draws them."""
    real_randint = torch.randint
    monkeypatch.setattr(torch, "randint", lambda *a, **k: cu(t, torch.int64) if "device" in k else real_randint(*a, **k))
    real_q = adcdiff.ops.q_sample
    monkeypatch.setattr(adcdiff.ops, "q_sample", lambda *a, **k: real_q(*a, **{**k, "noise": cu(nz)}))


def test_checkpoint_chain_matches_reference(tmp_path, monkeypatch):
    """train_step -> save_model -> load_model into a fresh model -> forward and a second train_step, against what the
    reference produced on the same chain through its own save_model / load_model (tests/golden/tiny_ckpt.npz, written by
    tests/golden/make_checkpoint.py).  The second step's AdamW update uses the restored moments and step count (bias
    corrections of step 2), so a checkpoint that dropped or reordered optimizer state would show in the weights.
    Tolerances as for the single step: loss 1e-5 relative; weights 99.5 % within 2e-6 of the reference's, all within 1e-4
    (two steps of lr = 1e-4 on entries whose gradient is float32 noise)."""
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN, "tiny_ckpt.npz"))
    m = build("fp32", use_mixed_precision=False)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    with monkeypatch.context() as mp_:
        _feed_draws(mp_, g["step1_t"], philox.normal(SEED, philox.STREAM_QNOISE, 0, 0, 8, 128))
        res = m.train_step(cu(g["step1_x0"]), 0)
    assert abs(res["loss"] - float(g["step1_loss"])) < 1e-5 * float(g["step1_loss"])
    path = str(tmp_path / "ckpt.pth")
    m.save_model(path)
    saved = torch.load(path, map_location="cpu", weights_only=True)
    assert set(saved) == {"model_state_dict", "optimizer_state_dict", "config"}
    assert len(saved["optimizer_state_dict"]["state"]) == int(g["opt_n_state"])
    assert float(saved["optimizer_state_dict"]["state"][0]["step"]) == float(g["opt_step"])
    norm = np.sqrt(sum(float((s["exp_avg"].double() ** 2).sum()) for s in saved["optimizer_state_dict"]["state"].values()))
    assert abs(norm - float(g["opt_exp_avg_norm"])) < 1e-4 * float(g["opt_exp_avg_norm"])

    fresh = adcdiff.DiffusionModel(dict(input_dim=128, hidden_dim=128, output_dim=128, compute_dtype="fp32", use_mixed_precision=False))
    for mod in fresh.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    fresh.load_model(path)
    a, b = fresh.optimizer.state_dict(), saved["optimizer_state_dict"]
    assert a["param_groups"] == b["param_groups"]
    for i, st in b["state"].items():
        for k in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(a["state"][i][k].cpu(), st[k]), (i, k)
    for k in ("output_layer.weight", "middle.0.linear1.weight", "encoder.3.residual_proj.bias"):
        got, ref = golden_tensor(g, "loaded." + k, fresh.model.state_dict()[k])
        diff = np.abs(got - ref)
        assert (diff < 2e-6).mean() > 0.995 and diff.max() < 5e-5, k
    fresh.model.eval()
    y = fresh.model(cu(philox.normal(SEED, 50, 0, 0, 8, 128)), cu(g["fwd_t"], torch.int64)).cpu().numpy()
    assert rel_err(y, g["fwd_restored"]) < 1e-4     # forward of weights that differ from the reference's by <= 5e-5
    with monkeypatch.context() as mp_:
        _feed_draws(mp_, g["step2_t"], philox.normal(SEED, philox.STREAM_QNOISE, 1, 0, 8, 128))
        res2 = fresh.train_step(cu(g["step2_x0"]), 1)
    assert abs(res2["loss"] - float(g["step2_loss"])) < 2e-5 * float(g["step2_loss"])
    sd = fresh.model.state_dict()
    for k in ("output_layer.weight", "middle.0.linear1.weight", "encoder.0.norm1.weight", "time_embed.linear1.bias",
              "decoder.2.time_proj.1.weight", "encoder.3.residual_proj.bias"):
        got, ref = golden_tensor(g, "step2_after." + k, sd[k])
        diff = np.abs(got - ref)
        assert (diff < 4e-6).mean() > 0.99 and diff.max() < 1e-4, (k, float(diff.max()), float((diff < 4e-6).mean()))
    assert float(fresh.optimizer.state_dict()["state"][0]["step"]) == 2.0


def test_stale_training_tape_raises():
    """The taped activations of a training forward live in the engine's one workspace: using the model again before
    backward() must raise instead of differentiating through the wrong activations."""
    from adcdiff._lib import AdcdiffError
    m = build("fp32")
    x, t = cu(philox.normal(SEED, 52, 0, 0, 8, 128)), torch.arange(8, device="cuda") * 100
    l1 = m.p_losses(x, t)
    l2 = m.p_losses(x[:4], t[:4])
    with pytest.raises(AdcdiffError, match="workspace was used again"):
        (l1 + l2).backward()
    m.model.zero_grad(set_to_none=True)
    l1 = m.p_losses(x, t)
    m.sample(4, use_ddim=False, seed=1, t_start=999, t_stop=998)
    with pytest.raises(AdcdiffError, match="workspace was used again"):
        l1.backward()
    m.model.zero_grad(set_to_none=True)
    l1 = m.p_losses(x, t)
    big = m.p_losses(cu(philox.normal(SEED, 52, 0, 0, 600, 128)), torch.arange(600, device="cuda"))   # re-allocates the workspace
    with pytest.raises(AdcdiffError):
        l1.backward()
    big.backward()                                   # the latest loss is still differentiable
    assert all(torch.isfinite(p.grad).all() for p in m.model.parameters() if p.grad is not None)
    with pytest.raises(RuntimeError):                # AdcdiffError is one; autograd's own "second time" error would do too
        big.backward()                               # and only once


def test_gradient_accumulation_keeps_earlier_gradients():
    """Two backward passes without zero_grad: the second must add to the first, although both are computed into the same
    persistent gradient arena (which the first pass's .grad tensors alias)."""
    m = build("fp32")
    m.model.eval()
    m.model.alias_gradients = True                   # as inside train_step
    x, t = cu(philox.normal(SEED, 52, 0, 0, 16, 128)), torch.arange(16, device="cuda") * 50
    nz = cu(philox.normal(SEED, 53, 0, 0, 16, 128))
    m.p_losses(x[:8], t[:8], noise=nz[:8]).backward()
    flat = m.model.grad_arena_if_aliased()
    assert flat is not None                          # nothing was copied: the gradients are views of the arena
    ga = {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}
    m.p_losses(x[8:], t[8:], noise=nz[8:]).backward()
    assert m.model.grad_arena_if_aliased() is None   # the accumulated gradients moved to a buffer of their own
    total = {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}
    m.model.zero_grad(set_to_none=True)
    m.p_losses(x[8:], t[8:], noise=nz[8:]).backward()
    for k, p in m.model.named_parameters():
        if p.grad is not None:
            assert torch.allclose(total[k], ga[k] + p.grad, rtol=0, atol=1e-6 * max(1.0, float(total[k].abs().max()))), k


def test_public_p_losses_honours_the_autograd_contract():
    """Outside train_step the gradients are ordinary autograd results: torch.autograd.grad returns them, tensor hooks and
    post-accumulate hooks fire on the first backward, and a .grad kept across steps is not overwritten by the next one."""
    m = build("fp32")
    m.model.eval()
    x, t = cu(philox.normal(SEED, 52, 0, 0, 16, 128)), torch.arange(16, device="cuda") * 50
    nz = cu(philox.normal(SEED, 53, 0, 0, 16, 128))
    w = m.model.middle[0].linear1.weight
    seen = []
    h1 = w.register_hook(lambda g: seen.append(("tensor", float(g.abs().sum()))))
    h2 = w.register_post_accumulate_grad_hook(lambda p: seen.append(("post", float(p.grad.abs().sum()))))
    m.p_losses(x[:8], t[:8], noise=nz[:8]).backward()
    assert [k for k, _ in seen] == ["tensor", "post"] and seen[0][1] > 0 and seen[0][1] == seen[1][1]
    h1.remove(); h2.remove()
    kept = w.grad
    snapshot = kept.clone()
    assert m.model.grad_arena_if_aliased() is None
    m.model.zero_grad(set_to_none=True)
    params = [p for p in m.model.parameters() if p.requires_grad]
    got = torch.autograd.grad(m.p_losses(x[8:], t[8:], noise=nz[8:]), params, allow_unused=True)
    assert all(p.grad is None for p in params)                      # autograd.grad does not write .grad
    assert torch.equal(kept, snapshot)                              # the earlier step's gradient tensor is untouched
    have = [g for g in got if g is not None]
    assert len(have) >= 100 and all(torch.isfinite(g).all() for g in have)
    m.p_losses(x[8:], t[8:], noise=nz[8:]).backward()
    for p, g in zip(params, got):
        if g is not None:   # bias and GroupNorm gradients are summed with atomics: equal up to the summation order
            assert torch.allclose(p.grad, g, rtol=1e-4, atol=1e-6 * float(g.abs().max()))
    # a scaled loss scales the gradients (the factor is applied to d(eps) before the backward kernels)
    m.model.zero_grad(set_to_none=True)
    (m.p_losses(x[8:], t[8:], noise=nz[8:]) * 0.25).backward()
    g_w = next(g for p, g in zip(params, got) if p is w)
    assert torch.allclose(w.grad, 0.25 * g_w, rtol=1e-5, atol=1e-9)


def test_conditional_loss_broadcasts_one_condition_and_drops_out_in_train_mode():
    """A [D] or [1, D] condition is shared by the batch (as t_emb + c_emb broadcasts, diffusion_model.py:766) and equals the
    repeated condition; under no_grad in train mode the conditional loss has dropout on, like the unconditional one."""
    m = build("fp32")
    x, t = cu(philox.normal(SEED, 52, 0, 0, 16, 128)), torch.arange(16, device="cuda") * 60
    nz = cu(philox.normal(SEED, 53, 0, 0, 16, 128))
    c1 = cu(philox.normal(SEED, 51, 0, 0, 1, 128))
    m.model.eval()
    la = m.p_losses(x, t, c1[0], noise=nz); la.backward()
    ga = {k: p.grad.clone() for k, p in m.model.named_parameters() if p.grad is not None}
    m.model.zero_grad(set_to_none=True)
    lb = m.p_losses(x, t, c1.repeat(16, 1), noise=nz); lb.backward()
    assert float(la.detach()) == float(lb.detach())
    for k, p in m.model.named_parameters():
        if p.grad is not None:
            assert torch.allclose(ga[k], p.grad, rtol=1e-4, atol=1e-6 * float(p.grad.abs().max())), k
    assert "condition_embed.0.weight" in ga and float(ga["condition_embed.0.weight"].abs().max()) > 0
    with torch.no_grad():
        e1 = float(m.p_losses(x, t, c1, noise=nz))
        m.model.train()
        torch.manual_seed(11); d1 = float(m.p_losses(x, t, c1, noise=nz))
        torch.manual_seed(11); d2 = float(m.p_losses(x, t, c1, noise=nz))
    # the value-only eval forward runs the fused kernels, the differentiable one the taped forward: equal to summation order
    assert abs(e1 - float(la.detach())) < 1e-5 * e1 and d1 == d2 and abs(d1 - e1) > 1e-4 * e1


def test_validation_loss_follows_training_mode():
    """p_losses under no_grad: eval mode gives the dropout-free value; train mode (where train_step leaves the model,
    as in the reference's validation pass, main.py:646-649) applies dropout, reproducibly for a fixed torch seed."""
    m = build("fp32")
    x, t = cu(philox.normal(SEED, 52, 0, 0, 64, 128)), torch.arange(64, device="cuda") * 15
    nz = cu(philox.normal(SEED, 53, 0, 0, 64, 128))
    with torch.no_grad():
        m.model.eval()
        e1, e2 = float(m.p_losses(x, t, noise=nz)), float(m.p_losses(x, t, noise=nz))
        m.model.train()
        torch.manual_seed(11); d1 = float(m.p_losses(x, t, noise=nz))
        torch.manual_seed(11); d2 = float(m.p_losses(x, t, noise=nz))
        torch.manual_seed(12); d3 = float(m.p_losses(x, t, noise=nz))
    assert e1 == e2 and d1 == d2 and d1 != e1 and d3 != d1
    assert abs(d1 - e1) < 0.2 * e1


def _dp_worker(rank, world, port, out_dir):
    import os
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import load_golden
        g = load_golden("tiny")
        m = build("fp32", use_mixed_precision=False)
        m.model.eval()
        m.model.alias_gradients = True
        lo, hi = rank * 4, rank * 4 + 4
        nz = cu(philox.normal(SEED, philox.STREAM_QNOISE, 0, 0, 8, 128))
        m.p_losses(cu(g["train_x0"])[lo:hi], cu(g["train_t"], torch.int64)[lo:hi], noise=nz[lo:hi]).backward()
        from adcdiff.parallel import allreduce_gradients
        flat = m.model.grad_arena_if_aliased()
        assert flat is not None
        allreduce_gradients(m.model.parameters(), flat=flat)
        np.savez(os.path.join(out_dir, "dp%d.npz" % rank), **{k: p.grad.cpu().numpy() for k, p in m.model.named_parameters() if p.grad is not None})
        # the timestep draw of train_step: ranks seeded alike keep different slices of one global draw
        torch.manual_seed(3)
        import adcdiff.models.diffusion_model as dm
        r, w = dm._dp_rank_world()
        t = torch.randint(0, 1000, (w * 4,), device="cuda")[r * 4:(r + 1) * 4]
        np.save(os.path.join(out_dir, "t%d.npy" % rank), t.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_data_parallel_gradients_match_full_batch_reference(golden_tiny, tmp_path):
    """Two processes (gloo, both on this GPU) each back-propagate one half of the golden batch through the HIP path and
    average their gradient arenas in place: the result must be the reference's full-batch gradient (tiny.npz), on both ranks."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = golden_tiny
    a, b = np.load(str(tmp_path / "dp0.npz")), np.load(str(tmp_path / "dp1.npz"))
    checked = 0
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
        key = "grad." + k
        if key in g.files or key + ".sample" in g.files:
            got, ref = golden_tensor(g, key, torch.from_numpy(a[k]))
            assert np.abs(got - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6), k
            checked += 1
    assert checked >= 14
    t0, t1 = np.load(str(tmp_path / "t0.npy")), np.load(str(tmp_path / "t1.npy"))
    assert not np.array_equal(t0, t1)


def test_taped_forward_reproduces_bit_for_bit_at_1024_rows():
    """The training forward at 1024 rows runs its GEMMs on the one-wave kernel with sixteen row tiles per launch.  A piece of the next
    tile issued into a ring stage whose fragment reads had not all returned overtook them on a lightly loaded chip: eight rows of a
    tile were multiplied with the wrong K tile in one forward pass out of 3 to 50, by machine.  150 repetitions of one pass (same
    inputs, same dropout seed) must give the same predicted noise bit for bit."""
    dims = (512, 256, 512)
    m = adcdiff.DiffusionModel(dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], compute_dtype="bf16", max_batch_size=4096))
    m.model.train()
    gen = torch.Generator(device="cpu").manual_seed(11)
    B = 1024
    x = torch.randn(B, dims[0], generator=gen).cuda()
    t = torch.randint(0, 1000, (B,), generator=gen).cuda()
    dev = m.model._device()
    eng = m.model.engine()
    handle = eng.ensure(dev)
    eng.train_workspace_for(B, dev)
    first = None
    for _ in range(150):
        eps = torch.ops.adcdiff.unet_train_forward(handle.value, x, t, None, 0.1, 12345, m.model.output_dim, eng.workspace)
        if first is None:
            first = eps.clone()
        else:
            assert torch.equal(eps, first)
