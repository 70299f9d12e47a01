"""Drop-in mirror of the reference's ``src/models/diffusion_model.py`` class surface on MI355X.

Same constructor dict, attribute names, method names and ``state_dict`` keys as the reference
(``DiffusionModel`` diffusion_model.py:24-107, ``UNetModel`` :677-747, ``TimeEmbedding`` :792-800,
``ResidualBlock`` :826-873); the arithmetic is done by libadcdiff.so (hand-written gfx950 kernels
behind include/adcdiff.h).  The modules below hold the parameters under the reference's names; their
``forward`` methods call the HIP library and raise if the tensors are not on the GPU -- there is no
PyTorch or CPU implementation of the path in this package.

Deliberate differences from the reference, all of them places where the reference cannot run:
* ``_extract`` gathers on the tensor's own device (the reference indexes a device table with a CPU
  index, :505-507, which raises on a GPU);
* batches larger than ``max_batch_size`` are sampled in one piece (the reference's chunk path reads a
  non-existent attribute, :277); ``max_batch_size`` is kept as an attribute only;
* noise comes from a counter RNG (Philox4x32-10 keyed by seed, global row, timestep) instead of
  torch's global generator, so a batch sharded over GPUs draws the same noise as the unsharded one;
  the seed of each call is drawn from torch's CPU generator, so ``torch.manual_seed`` still makes
  runs reproducible;
* sampling runs the denoiser in eval mode (dropout off): the reference never calls ``.eval()`` and
  therefore samples with dropout active, which no parity test could pin.  ``p_losses`` follows
  ``self.model.training`` with and without gradients: a validation loss taken under ``no_grad`` right after
  ``train_step`` (which leaves the model in train mode, main.py:646-649) has dropout on, as in the reference;
* supported widths: ``input_dim``, ``hidden_dim`` and ``output_dim`` multiples of 64 with ``input_dim ==
  output_dim`` (the reverse loop feeds the output back as the input, :299, :702); training additionally
  needs ``hidden_dim`` a multiple of 128 and a power of two up to 1024 or a multiple of 1024.  Other
  widths raise ``ValueError`` at construction (``strict_shapes=False`` in the config builds the parameter
  containers anyway, e.g. to read or write a checkpoint);
* data-parallel training draws the timesteps of the GLOBAL batch on every rank and keeps its own slice, and keys
  the forward-diffusion noise by global row, so ranks seeded alike still train on independent samples.
"""
import ctypes as C
import gc
import logging
import math
from typing import Dict, Optional, Tuple

import psutil
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.optim import Adam, AdamW
from torch.optim.lr_scheduler import CosineAnnealingLR, OneCycleLR

from .. import _lib, ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.adcdiff.*)
from .._lib import AdcdiffError, check


def get_num_groups(dim):
    """Largest divisor of dim that is <= 32 (diffusion_model.py:723-730, 840-847)."""
    if dim < 4:
        return 1
    for num_groups in range(min(32, dim), 0, -1):
        if dim % num_groups == 0:
            return num_groups
    return 1


def _seed_from_torch():
    """A 62-bit seed drawn from torch's CPU generator (so torch.manual_seed controls sampling)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def check_supported_dims(input_dim, hidden_dim, output_dim):
    """The widths the HIP kernels take (MFMA tiles of 64 columns, GroupNorm groups that divide them); raises ValueError."""
    bad = [n for n, v in (("input_dim", input_dim), ("hidden_dim", hidden_dim), ("output_dim", output_dim))
           if not isinstance(v, int) or v <= 0 or v % 64 != 0]
    if bad:
        raise ValueError("adcdiff supports input_dim / hidden_dim / output_dim that are positive multiples of 64; got "
                         "input_dim=%r hidden_dim=%r output_dim=%r (%s). Pad the feature width to a multiple of 64, or pass "
                         "strict_shapes=False to build the parameter containers only." % (input_dim, hidden_dim, output_dim, ", ".join(bad)))
    if input_dim != output_dim:
        raise ValueError("adcdiff needs input_dim == output_dim (got %d and %d): the reverse loop feeds the [B, output_dim] "
                         "latent back as the [B, input_dim] input and the loss compares them element-wise" % (input_dim, output_dim))


def _dp_rank_world():
    """(rank, world size) of the default process group, (0, 1) without one."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class TimeEmbedding(nn.Module):
    """Parameters of the reference's TimeEmbedding (diffusion_model.py:792-823)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim
        self.linear1 = nn.Linear(dim, dim * 4)
        self.linear2 = nn.Linear(dim * 4, dim)

    def frequencies(self):
        """exp(arange(half) * -(ln 1e4 / (half - 1))) computed with the reference's own expression on
        the CPU (:812-814), so the table is the one the reference gets on the same host."""
        half_dim = self.dim // 2
        emb = math.log(10000) / (half_dim - 1)
        return torch.exp(torch.arange(half_dim) * -emb)

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        ops._need_gpu(t)
        freq = self.frequencies().to(t.device)
        arg = t[:, None] * freq[None, :]
        # the sinusoid of a handful of rows; the sampling path uses the library's per-timestep table instead
        emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
        emb = ops.linear(emb, self.linear1.weight, self.linear1.bias, act=1)
        return ops.linear(emb, self.linear2.weight, self.linear2.bias)


class ResidualBlock(nn.Module):
    """Parameters of the reference's ResidualBlock (diffusion_model.py:826-873); attention is dead code
    in the reference (never enabled, :702-719) and is not built."""

    def __init__(self, in_dim: int, out_dim: int, time_emb_dim: int = None, use_attention: bool = False,
                 dropout: float = 0.1):
        super().__init__()
        if use_attention:
            raise NotImplementedError("use_attention is never enabled by the reference and is not built")
        if time_emb_dim is None:
            time_emb_dim = out_dim
        if in_dim < 4 or out_dim < 4:
            raise ValueError("widths below 4 use LayerNorm in the reference and are not supported")
        self.use_attention = False
        self.in_dim, self.out_dim = in_dim, out_dim
        self.norm1 = nn.GroupNorm(get_num_groups(in_dim), in_dim)
        self.norm2 = nn.GroupNorm(get_num_groups(out_dim), out_dim)
        self.linear1 = nn.Linear(in_dim, out_dim)
        self.linear2 = nn.Linear(out_dim, out_dim)
        self.dropout = nn.Dropout(dropout)   # identity in the sampling path
        self.time_proj = nn.Sequential(nn.SiLU(), nn.Linear(time_emb_dim, out_dim))
        self.residual_proj = nn.Linear(in_dim, out_dim) if in_dim != out_dim else nn.Identity()

    def forward(self, x: torch.Tensor, t_emb: torch.Tensor) -> torch.Tensor:
        """Eval-mode forward of one block through the single-kernel entry points (float32 operands)."""
        ops._need_gpu(x, t_emb)
        B = x.shape[0]
        P = ops._pad_rows(B)
        a1 = torch.zeros((P, self.in_dim), device=x.device)
        a1[:B] = ops.groupnorm_silu(x.contiguous(), self.norm1.num_groups, self.norm1.weight.detach(), self.norm1.bias.detach())
        s = torch.zeros((P, t_emb.shape[1]), device=x.device)
        s[:B] = t_emb / (1.0 + torch.exp(-t_emb))
        w1 = torch.cat([self.linear1.weight.detach(), self.time_proj[1].weight.detach()], dim=1).contiguous()
        h = ops.gemm([a1, s], w1, bias=(self.linear1.bias + self.time_proj[1].bias).detach(), m_valid=B)
        a2 = torch.zeros((P, self.out_dim), device=x.device)
        a2[:B] = ops.groupnorm_silu(h[:B], self.norm2.num_groups, self.norm2.weight.detach(), self.norm2.bias.detach())
        if isinstance(self.residual_proj, nn.Identity):
            out = ops.gemm([a2], self.linear2.weight.detach().contiguous(), bias=self.linear2.bias.detach(),
                           residual=x.contiguous(), m_valid=B)
        else:
            xr = torch.zeros((P, self.in_dim), device=x.device)
            xr[:B] = x
            w2 = torch.cat([self.linear2.weight.detach(), self.residual_proj.weight.detach()], dim=1).contiguous()
            out = ops.gemm([a2, xr], w2, bias=(self.linear2.bias + self.residual_proj.bias).detach(), m_valid=B)
        return out[:B]


class _Engine:
    """Owns the packed-weight arena, the workspace and the native plan of one UNetModel."""

    def __init__(self, model, dtype_code):
        self.lib = _lib.load()
        self.model = model
        self.dtype = dtype_code
        self.handle = None
        self.arena = None
        self.workspace = None
        self.key = None
        self.version = None
        self.freq = None
        self.generation = 0     # bumped by every call that (re)uses the workspace: a taped forward is valid for one generation

    def _param_key(self):
        return tuple(p.data_ptr() for p in self.model.parameters())

    def _param_version(self):
        return sum(p._version for p in self.model.parameters())

    def destroy(self):
        if self.handle is not None:
            self.lib.adcdiff_unet_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def ensure(self, device):
        m = self.model
        key = self._param_key()
        if self.handle is None or key != self.key:
            self.destroy()
            for p in m.parameters():
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise AdcdiffError("UNetModel parameters must be contiguous float32 tensors on the GPU "
                                       "(found %s on %s); there is no CPU fallback" % (p.dtype, p.device))
            self.freq = m.time_embed.frequencies().to(device=device, dtype=torch.float32).contiguous()
            prm = _lib.UnetParams()
            prm.input_dim, prm.hidden_dim, prm.output_dim, prm.num_timesteps = m.input_dim, m.hidden_dim, m.output_dim, m.num_timesteps
            prm.time_lin1_w, prm.time_lin1_b = m.time_embed.linear1.weight.data_ptr(), m.time_embed.linear1.bias.data_ptr()
            prm.time_lin2_w, prm.time_lin2_b = m.time_embed.linear2.weight.data_ptr(), m.time_embed.linear2.bias.data_ptr()
            prm.cond0_w, prm.cond0_b = m.condition_embed[0].weight.data_ptr(), m.condition_embed[0].bias.data_ptr()
            prm.cond2_w, prm.cond2_b = m.condition_embed[2].weight.data_ptr(), m.condition_embed[2].bias.data_ptr()
            blocks = list(m.encoder) + list(m.middle) + list(m.decoder)
            for i, b in enumerate(blocks):
                bp = prm.blocks[i]
                bp.norm1_w, bp.norm1_b = b.norm1.weight.data_ptr(), b.norm1.bias.data_ptr()
                bp.norm2_w, bp.norm2_b = b.norm2.weight.data_ptr(), b.norm2.bias.data_ptr()
                bp.linear1_w, bp.linear1_b = b.linear1.weight.data_ptr(), b.linear1.bias.data_ptr()
                bp.linear2_w, bp.linear2_b = b.linear2.weight.data_ptr(), b.linear2.bias.data_ptr()
                bp.time_proj_w, bp.time_proj_b = b.time_proj[1].weight.data_ptr(), b.time_proj[1].bias.data_ptr()
                if isinstance(b.residual_proj, nn.Linear):
                    bp.residual_w, bp.residual_b = b.residual_proj.weight.data_ptr(), b.residual_proj.bias.data_ptr()
                else:
                    bp.residual_w, bp.residual_b = None, None
            prm.out_norm_w, prm.out_norm_b = m.output_norm.weight.data_ptr(), m.output_norm.bias.data_ptr()
            prm.out_w, prm.out_b = m.output_layer.weight.data_ptr(), m.output_layer.bias.data_ptr()
            prm.sinusoid_freq = self.freq.data_ptr()
            nbytes = self.lib.adcdiff_unet_arena_bytes(m.input_dim, m.hidden_dim, m.output_dim, m.num_timesteps, self.dtype)
            if nbytes == 0:
                raise AdcdiffError("unsupported denoiser dimensions: %s" % self.lib.adcdiff_last_error().decode())
            self.arena = torch.empty(nbytes, device=device, dtype=torch.uint8)
            handle = C.c_void_p()
            check(self.lib.adcdiff_unet_create(C.byref(prm), self.dtype, C.c_void_p(self.arena.data_ptr()), nbytes, C.byref(handle)))
            self.handle, self.key, self.version = handle, key, None
        version = self._param_version()
        if version != self.version:
            check(self.lib.adcdiff_unet_pack(self.handle, None, 0, ops._stream()))
            self.version = version
        return self.handle

    def train_workspace_for(self, batch, device):
        m = self.model
        need = self.lib.adcdiff_unet_train_workspace_bytes(m.input_dim, m.hidden_dim, m.output_dim, m.num_timesteps, self.dtype, batch)
        if need == 0:
            raise AdcdiffError(self.lib.adcdiff_last_error().decode())
        if self.workspace is None or self.workspace.numel() < need or self.workspace.device != device:
            self.workspace = None
            self.workspace = torch.empty(need, device=device, dtype=torch.uint8)
        self.generation += 1
        return C.c_void_p(self.workspace.data_ptr()), self.workspace.numel()

    def workspace_for(self, batch, device):
        m = self.model
        need = self.lib.adcdiff_unet_workspace_bytes(m.input_dim, m.hidden_dim, m.output_dim, self.dtype, batch)
        if need == 0:
            raise AdcdiffError(self.lib.adcdiff_last_error().decode())
        if self.workspace is None or self.workspace.numel() < need or self.workspace.device != device:
            self.workspace = None
            self.workspace = torch.empty(need, device=device, dtype=torch.uint8)
        self.generation += 1
        return C.c_void_p(self.workspace.data_ptr()), self.workspace.numel()


def _grad_slots(model):
    """(parameter, attribute path into _lib.UnetGrads) for every parameter the backward kernels produce a gradient for."""
    slots = [(model.time_embed.linear1.weight, ("time_lin1_w",)), (model.time_embed.linear1.bias, ("time_lin1_b",)),
             (model.time_embed.linear2.weight, ("time_lin2_w",)), (model.time_embed.linear2.bias, ("time_lin2_b",))]
    for i, b in enumerate(list(model.encoder) + list(model.middle) + list(model.decoder)):
        slots += [(b.norm1.weight, (i, "norm1_w")), (b.norm1.bias, (i, "norm1_b")), (b.norm2.weight, (i, "norm2_w")),
                  (b.norm2.bias, (i, "norm2_b")), (b.linear1.weight, (i, "linear1_w")), (b.linear1.bias, (i, "linear1_b")),
                  (b.linear2.weight, (i, "linear2_w")), (b.linear2.bias, (i, "linear2_b")),
                  (b.time_proj[1].weight, (i, "time_proj_w")), (b.time_proj[1].bias, (i, "time_proj_b"))]
        if isinstance(b.residual_proj, nn.Linear):
            slots += [(b.residual_proj.weight, (i, "residual_w")), (b.residual_proj.bias, (i, "residual_b"))]
    slots += [(model.output_norm.weight, ("out_norm_w",)), (model.output_norm.bias, ("out_norm_b",)),
              (model.output_layer.weight, ("out_w",)), (model.output_layer.bias, ("out_b",))]
    # last: written (and handed to autograd) only by a conditional loss; the reference's train_step never passes a condition
    slots += [(model.condition_embed[0].weight, ("cond0_w",)), (model.condition_embed[0].bias, ("cond0_b",)),
              (model.condition_embed[2].weight, ("cond2_w",)), (model.condition_embed[2].bias, ("cond2_b",))]
    return slots


_N_COND_SLOTS = 4


def _taped_forward_loss(model, x_noisy, t, noise, dropout_p, dropout_seed, condition=None):
    """Training-mode denoiser forward (dropout from the counter RNG; condition: None or [B, input_dim] rows whose
    condition_embed joins the time embedding) + mse_loss(noise, prediction) in the HIP library.
    Returns (loss, d_eps, engine, handle, workspace tensor, generation): the activations the backward pass needs stay
    in the engine's workspace until the next call that uses it (which bumps the generation)."""
    dev = model._device()
    eng = model.engine()
    handle = eng.ensure(dev)
    B = x_noisy.shape[0]
    eng.train_workspace_for(B, dev)
    eps = torch.ops.adcdiff.unet_train_forward(handle.value, x_noisy, t, condition, float(dropout_p), int(dropout_seed), model.output_dim, eng.workspace)
    loss, d_eps = torch.ops.adcdiff.mse_loss_grad(eps, noise, 1.0)
    return loss, d_eps, eng, handle, eng.workspace, eng.generation


class _NoisePredictionLoss(torch.autograd.Function):
    """mse_loss(noise, UNet(x_noisy, t)) with the denoiser's forward AND backward in the HIP library
    (adcdiff_unet_train_forward / adcdiff_mse_loss_grad / adcdiff_unet_backward).  The parameters enter as inputs
    only so that autograd routes the library's gradients to them.  The taped activations live in the engine's single
    workspace: backward raises if anything else has used the engine since this forward (a second p_losses, a sample(),
    a forward()), instead of differentiating through the wrong activations."""

    @staticmethod
    def forward(ctx, model, x_noisy, t, noise, dropout_p, dropout_seed, condition, *params):
        loss, d_eps, eng, handle, ws_tensor, generation = _taped_forward_loss(model, x_noisy, t, noise, dropout_p, dropout_seed, condition)
        ctx.model, ctx.eng, ctx.handle, ctx.d_eps, ctx.t = model, eng, handle, d_eps, t
        ctx.conditional = condition is not None
        ctx.inputs = (x_noisy, t, noise, condition)     # the tape points at x_noisy and t: they must outlive the forward call
        ctx.ws_tensor = ws_tensor            # keeps the buffer alive even if the engine re-allocates a larger one
        ctx.generation = generation
        ctx.version = eng._param_version()
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        model, eng = ctx.model, ctx.eng
        if eng._param_version() != ctx.version or eng.handle is None or eng.handle.value != ctx.handle.value:
            raise AdcdiffError("parameters changed between the training forward and backward")
        if eng.generation != ctx.generation or eng.workspace is not ctx.ws_tensor:
            raise AdcdiffError("the denoiser's workspace was used again (another p_losses / sample / forward call) between this "
                               "loss's forward and its backward: the taped activations are gone. Call backward() on each loss "
                               "before the next call into the model.")
        params = list(model.parameters())
        slots = _grad_slots(model)
        flat, views = model._grad_arena(slots)
        # gradient accumulation: gradients of an earlier micro-step that still live in the arena move to a buffer of their own
        if any(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for (p, _), v in zip(slots, views)):
            keep = flat.clone()
            off = 0
            for (p, _), v in zip(slots, views):
                if p.grad is not None and p.grad.data_ptr() == v.data_ptr():
                    p.grad = keep[off:off + p.numel()].view_as(p)
                off += (p.numel() + 3) // 4 * 4
        flat.zero_()                         # bias / GroupNorm gradients are accumulated with atomics
        grads = _lib.UnetGrads()
        for (p, path), g in zip(slots, views):
            target = grads if len(path) == 1 else grads.blocks[path[0]]
            setattr(target, path[-1], g.data_ptr())
        table = torch.frombuffer(bytearray(bytes(grads)), dtype=torch.int64)     # the adcdiff_unet_grads struct as a CPU tensor
        # d(loss * c) = c * d(loss): the backward pass is linear in d(eps), so the factor goes onto that [B, D] tensor
        # (a few MB) instead of a pass over the whole gradient arena; no host synchronisation
        d_eps = ctx.d_eps * grad_out.to(ctx.d_eps.dtype)
        torch.ops.adcdiff.unet_backward(ctx.handle.value, d_eps, table, flat, ctx.ws_tensor)
        eng.generation += 1                  # the tape is consumed
        # The gradients are RETURNED to autograd (its AccumulateGrad nodes install or add them, tensor hooks and
        # post-accumulate hooks fire, torch.autograd.grad() gets them).  train_step opts into aliasing: it hands out fresh
        # views of the persistent arena, which autograd adopts as .grad without a copy, so the all-reduce and the fused
        # optimizer work on one flat buffer -- such .grad tensors are overwritten by the next backward.  Every other caller
        # of p_losses gets views of a private copy that nothing overwrites.
        src = flat if model.alias_gradients else flat.clone()
        live = slots if ctx.conditional else slots[:-_N_COND_SLOTS]   # condition_embed has a gradient only under a conditional loss
        fresh, off = {}, 0
        for p, _ in live:
            fresh[id(p)] = src[off:off + p.numel()].view_as(p)
            off += (p.numel() + 3) // 4 * 4
        model._grad_live = len(live)
        out = [fresh.get(id(p)) if p.requires_grad else None for p in params]
        return (None, None, None, None, None, None, None) + tuple(out)


class UNetModel(nn.Module):
    """The reference's MLP "U-Net" noise predictor (diffusion_model.py:677-789) on the HIP library."""

    def __init__(self, input_dim: int, hidden_dim: int, output_dim: int, num_timesteps: int,
                 enable_memory_efficient_attention: bool = True, compute_dtype: str = "bf16", strict_shapes: bool = True):
        super().__init__()
        if strict_shapes:
            check_supported_dims(input_dim, hidden_dim, output_dim)
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.output_dim = output_dim
        self.num_timesteps = num_timesteps
        self.enable_memory_efficient_attention = enable_memory_efficient_attention
        self.compute_dtype = compute_dtype
        self.detect_shared_condition = False   # config key 'detect_shared_condition' of DiffusionModel, see expand_condition
        if input_dim <= 0:
            raise ValueError("input_dim must be positive")
        self.time_embed = TimeEmbedding(hidden_dim)
        self.condition_embed = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.SiLU(), nn.Linear(hidden_dim, hidden_dim))
        self.encoder = nn.ModuleList([
            ResidualBlock(input_dim, hidden_dim, hidden_dim),
            ResidualBlock(hidden_dim, hidden_dim * 2, hidden_dim),
            ResidualBlock(hidden_dim * 2, hidden_dim * 4, hidden_dim),
            ResidualBlock(hidden_dim * 4, hidden_dim * 8, hidden_dim),
        ])
        self.middle = nn.ModuleList([
            ResidualBlock(hidden_dim * 8, hidden_dim * 8, hidden_dim),
            ResidualBlock(hidden_dim * 8, hidden_dim * 8, hidden_dim),
        ])
        self.decoder = nn.ModuleList([
            ResidualBlock(hidden_dim * 16, hidden_dim * 4, hidden_dim),
            ResidualBlock(hidden_dim * 8, hidden_dim * 2, hidden_dim),
            ResidualBlock(hidden_dim * 4, hidden_dim, hidden_dim),
            ResidualBlock(hidden_dim * 2, hidden_dim, hidden_dim),
        ])
        self.output_norm = nn.GroupNorm(get_num_groups(hidden_dim), hidden_dim)
        self.output_layer = nn.Linear(hidden_dim, output_dim)
        self._initialize_weights()
        self._engines = {}
        self._grad_flat = None
        # True (train_step sets it around its backward; a caller may too): backward hands out views of ONE persistent flat
        # gradient buffer, so nothing is copied per step and a data-parallel all-reduce takes one collective -- but those
        # .grad tensors are overwritten by the next backward.  False: every backward returns gradients nothing overwrites.
        self.alias_gradients = False

    def _grad_arena(self, slots):
        """One persistent flat float32 buffer with a 16-byte aligned view per parameter gradient (allocated once)."""
        dev = self._device()
        total = sum((p.numel() + 3) // 4 * 4 for p, _ in slots)
        if self._grad_flat is None or self._grad_flat[0].numel() != total or self._grad_flat[0].device != dev:
            flat = torch.empty(total, device=dev, dtype=torch.float32)
            views, off = [], 0
            for p, _ in slots:
                views.append(flat[off:off + p.numel()].view_as(p))
                off += (p.numel() + 3) // 4 * 4
            self._grad_flat = (flat, views)
        return self._grad_flat

    def grad_arena_if_aliased(self):
        """The flat gradient buffer when every existing .grad is a view of it (one in-place all-reduce then averages them all)."""
        if self._grad_flat is None:
            return None
        flat, views = self._grad_flat
        ptrs = {v.data_ptr() for v in views}
        have = [p for p in self.parameters() if p.grad is not None]
        if have and all(p.grad.data_ptr() in ptrs for p in have) and len(have) == getattr(self, "_grad_live", len(views)):
            return flat
        return None

    def expand_condition(self, condition, batch):
        """[input_dim] or [1, input_dim] -> one row (broadcast later), [batch, input_dim] as is; anything else raises, as the
        reference's t_emb + c_emb broadcast would (diffusion_model.py:766)."""
        if condition.dim() == 1:
            condition = condition.unsqueeze(0)
        if condition.dim() != 2 or condition.shape[1] != self.input_dim:
            raise ValueError("condition must be [B, %d], [1, %d] or [%d]; got %s" % (self.input_dim, self.input_dim, self.input_dim,
                                                                                     tuple(condition.shape)))
        if condition.shape[0] not in (1, batch):
            raise ValueError("condition has %d rows for a batch of %d (1 or %d expected)" % (condition.shape[0], batch, batch))
        if condition.shape[0] > 1 and condition.stride(0) == 0:
            condition = condition[0].unsqueeze(0)   # one row expanded to the batch (MoleculeGenerator, sample_ddpm_sharded): still one row
        elif condition.shape[0] > 1 and self.detect_shared_condition and bool((condition == condition[:1]).all()):
            # opt-in for callers that materialise the repeated row, like the reference's own generator (molecule_generator.py:238-245):
            # one comparison and one host synchronisation per call; off by default because a data-dependent choice would let a
            # shard whose rows happen to agree differ in the last bits from the same rows sampled in a mixed batch
            condition = condition[:1].contiguous()
        return condition

    def _initialize_weights(self):
        """xavier_uniform_ weights, zero biases, GroupNorm weight 1 / bias 0 (diffusion_model.py:738-747)."""
        for module in self.modules():
            if isinstance(module, nn.Linear):
                nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.zeros_(module.bias)
            elif isinstance(module, nn.GroupNorm):
                nn.init.ones_(module.weight)
                nn.init.zeros_(module.bias)

    # ---- native plumbing -------------------------------------------------------------------
    def engine(self, compute_dtype=None):
        code = ops.dtype_code(compute_dtype if compute_dtype is not None else self.compute_dtype)
        if code not in self._engines:
            self._engines[code] = _Engine(self, code)
        return self._engines[code]

    def _device(self):
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise AdcdiffError("the denoiser runs on an MI355X only (parameters are on %s); no CPU fallback exists" % dev)
        return dev

    def condition_embedding(self, condition, compute_dtype=None, batch=None, broadcast_view=False):
        """condition_embed(condition) (diffusion_model.py:694-698, 765): computed once per sampling call.  With `batch`
        the condition is checked against it and a one-row condition's embedding is repeated to [batch, hidden] (rows of
        the embedding do not depend on each other, so this equals embedding the repeated condition); with broadcast_view
        it stays one row in memory, expanded with a row stride of 0, which is how the sampling loops are told that the
        whole batch shares the condition (they then turn the time projection into a per-timestep row bias)."""
        dev = self._device()
        eng = self.engine(compute_dtype)
        handle = eng.ensure(dev)
        if batch is not None:
            condition = self.expand_condition(condition, batch)
        cond = condition.to(device=dev, dtype=torch.float32).contiguous()
        B = cond.shape[0]
        eng.workspace_for(B, dev)
        cemb = torch.ops.adcdiff.condition_embed(handle.value, cond, self.hidden_dim, eng.workspace)
        if batch is not None and B == 1:
            if broadcast_view:   # row stride 0 for every batch size, 1 included: a row's bits must not depend on the batch
                cemb = cemb.as_strided((batch, cemb.shape[1]), (0, 1))
            elif batch > 1:
                cemb = cemb.expand(batch, -1).contiguous()
        return cemb

    def forward(self, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Predicted noise for x [B, input_dim], per-row timesteps t [B] (diffusion_model.py:749-789)."""
        dev = self._device()
        ops._need_gpu(x, t, condition)
        eng = self.engine()
        handle = eng.ensure(dev)
        x = x.to(torch.float32).contiguous()
        t = t.to(torch.int64).contiguous()
        B = x.shape[0]
        if x.shape[1] != self.input_dim or t.shape[0] != B:
            raise ValueError("x must be [B, %d] and t [B]" % self.input_dim)
        cemb = self.condition_embedding(condition, batch=B) if condition is not None else None
        eng.workspace_for(B, dev)
        return torch.ops.adcdiff.unet_forward(handle.value, x, t, cemb, self.output_dim, eng.workspace)


class DiffusionModel(nn.Module):
    """Mirror of the reference's DiffusionModel (diffusion_model.py:24-674)."""

    def __init__(self, config: Dict):
        super().__init__()
        self.config = config
        self.logger = logging.getLogger(__name__)

        self.input_dim = config.get('input_dim', 512)
        self.hidden_dim = config.get('hidden_dim', 256)
        self.output_dim = config.get('output_dim', 128)
        self.num_timesteps = config.get('num_timesteps', 1000)

        self.use_mixed_precision = config.get('use_mixed_precision', True)
        self.gradient_accumulation_steps = config.get('gradient_accumulation_steps', 1)
        self.max_batch_size = config.get('max_batch_size', 32)
        self.enable_memory_efficient_attention = config.get('enable_memory_efficient_attention', True)
        # mixed precision on MI355X = bfloat16 MFMA operands with float32 accumulation; 'compute_dtype'
        # ('bf16' / 'fp32') overrides the reference's boolean when given
        self.compute_dtype = config.get('compute_dtype', 'bf16' if self.use_mixed_precision else 'fp32')
        ops.dtype_code(self.compute_dtype)

        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')

        self.model = UNetModel(
            input_dim=self.input_dim,
            hidden_dim=self.hidden_dim,
            output_dim=self.output_dim,
            num_timesteps=self.num_timesteps,
            enable_memory_efficient_attention=self.enable_memory_efficient_attention,
            compute_dtype=self.compute_dtype,
            strict_shapes=config.get('strict_shapes', True),
        ).to(self.device)
        self.model.detect_shared_condition = bool(config.get('detect_shared_condition', False))

        # schedule tables: computed on the host with the reference's own expressions (:57-60), then moved
        self.beta_schedule = self._create_beta_schedule()
        self.alpha = 1.0 - self.beta_schedule
        self.alpha_cumprod = torch.cumprod(self.alpha, dim=0)
        self.alpha_cumprod_prev = F.pad(self.alpha_cumprod[:-1], (1, 0), value=1.0)
        self._ddim_plan_cache = {}   # (ddim_steps, eta) -> (timesteps, coefficient rows), host tensors
        self._host = dict(beta=self.beta_schedule.clone(), alpha=self.alpha.clone(), alpha_cumprod=self.alpha_cumprod.clone(),
                          alpha_cumprod_prev=self.alpha_cumprod_prev.clone())
        self.beta_schedule = self.beta_schedule.to(self.device)
        self.alpha = self.alpha.to(self.device)
        self.alpha_cumprod = self.alpha_cumprod.to(self.device)
        self.alpha_cumprod_prev = self.alpha_cumprod_prev.to(self.device)
        self._tables = {}

        optimizer_class = AdamW if config.get('use_adamw', True) else Adam
        self.optimizer = optimizer_class(self.model.parameters(), lr=config.get('learning_rate', 1e-4),
                                         weight_decay=config.get('weight_decay', 1e-2))
        self._fused_opt = None
        self._loss_host = None     # pinned scalar + event: how train_step reads the loss without waiting for the whole step
        self._loss_event = None
        if config.get('use_onecycle', False):
            self.scheduler = OneCycleLR(self.optimizer, max_lr=config.get('learning_rate', 1e-4),
                                        total_steps=config.get('total_steps', 10000))
        else:
            self.scheduler = CosineAnnealingLR(self.optimizer, T_max=config.get('max_epochs', 100))
        # diffusion_model.py:87 holds a GradScaler when use_mixed_precision is set.  bfloat16 operands with float32 accumulation and
        # float32 master weights need no loss scaling, so the attribute is a DISABLED scaler: scale() returns its argument, step()
        # calls optimizer.step(), unscale_() and update() do nothing -- code written against the reference's attribute keeps working.
        self.scaler = torch.amp.GradScaler("cuda", enabled=False) if self.use_mixed_precision else None

        self.memory_threshold = config.get('memory_threshold', 0.85)
        self.gc_frequency = config.get('gc_frequency', 50)
        self.operation_count = 0
        self.memory_monitor_enabled = config.get('memory_monitor_enabled', True)
        self.memory_check_interval = config.get('memory_check_interval', 10)
        self.logger.info("diffusion model ready, device: %s, compute dtype: %s", self.device, self.compute_dtype)

    # ---- schedule ---------------------------------------------------------------------------
    def _create_beta_schedule(self, schedule_type: str = 'linear') -> torch.Tensor:
        """diffusion_model.py:110-126 (always called with the default 'linear', :57)."""
        if schedule_type == 'linear':
            beta_start = self.config.get('beta_start', 1e-4)
            beta_end = self.config.get('beta_end', 2e-2)
            return torch.linspace(beta_start, beta_end, self.num_timesteps)
        elif schedule_type == 'cosine':
            # the reference's cosine table (:128-142) is unreachable there too: the constructor always asks for 'linear' (:57)
            raise NotImplementedError("only the linear schedule is reachable in the reference and built here")
        else:
            raise ValueError(f"unsupported schedule type: {schedule_type}")

    def _posterior_variance(self) -> torch.Tensor:
        """diffusion_model.py:509-519."""
        posterior_variance = self.beta_schedule * (1.0 - self.alpha_cumprod_prev) / (1.0 - self.alpha_cumprod)
        return torch.clamp(posterior_variance, min=1e-20)

    def _extract(self, a: torch.Tensor, t: torch.Tensor, x_shape: Tuple) -> torch.Tensor:
        """diffusion_model.py:494-507, gathering on the table's own device."""
        batch_size = t.shape[0]
        out = a.gather(-1, t.to(a.device))
        return out.reshape(batch_size, *((1,) * (len(x_shape) - 1))).to(t.device)

    def _host_tables(self):
        """Per-timestep scalars of the sampler, evaluated on the host with the reference's expressions
        (:197-201, :214-216) so they round exactly as they do there."""
        if "ddpm" not in self._tables:
            h = self._host
            pv = torch.clamp(h["beta"] * (1.0 - h["alpha_cumprod_prev"]) / (1.0 - h["alpha_cumprod"]), min=1e-20)
            sigma = torch.sqrt(pv)
            sigma[0] = 0.0   # the t == 0 branch returns the mean (:211-212)
            self._tables["ddpm"] = dict(beta=h["beta"], s1m=torch.sqrt(1.0 - h["alpha_cumprod"]),
                                        sra=torch.sqrt(1.0 / h["alpha"]), sigma=sigma)
        return self._tables["ddpm"]

    def _ddpm_coef_array(self):
        if "ddpm_c" not in self._tables:
            tb = self._host_tables()
            arr = (_lib.DdpmCoef * self.num_timesteps)()
            for i in range(self.num_timesteps):
                arr[i] = _lib.DdpmCoef(tb["sra"][i].item(), tb["beta"][i].item(), tb["s1m"][i].item(), tb["sigma"][i].item())
            self._tables["ddpm_c"] = arr
        return self._tables["ddpm_c"]

    def _ddpm_coef_tensor(self):
        """The per-timestep adcdiff_ddpm_coef structs as a CPU float32 [T, 4] tensor (what torch.ops.adcdiff.ddpm_sample_loop takes)."""
        if "ddpm_t" not in self._tables:
            tb = self._host_tables()
            self._tables["ddpm_t"] = torch.stack([tb["sra"], tb["beta"], tb["s1m"], tb["sigma"]], dim=1).to(torch.float32).contiguous()
        return self._tables["ddpm_t"]

    def _device_tables(self):
        if "dev" not in self._tables:
            self._tables["dev"] = dict(sqrt_ac=torch.sqrt(self._host["alpha_cumprod"]).to(self.device),
                                       sqrt_1mac=torch.sqrt(1.0 - self._host["alpha_cumprod"]).to(self.device))
        return self._tables["dev"]

    # ---- forward process and loss -----------------------------------------------------------
    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """diffusion_model.py:144-163."""
        dv = self._device_tables()
        seed = _seed_from_torch() if noise is None else 0
        return ops.q_sample(x_start.to(torch.float32).contiguous(), t.to(torch.int64).contiguous(), dv["sqrt_ac"], dv["sqrt_1mac"],
                            noise=noise.contiguous() if noise is not None else None, seed=seed)

    def p_losses(self, x_start: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None, row_offset: int = 0) -> torch.Tensor:
        """diffusion_model.py:165-184.  With gradients enabled the returned loss back-propagates through the HIP
        backward kernels, with a condition also into condition_embed (:764-766).  Dropout follows self.model.training with
        and without gradients, with and without a condition.  row_offset keys the counter-RNG noise by global row
        (data-parallel training)."""
        dv = self._device_tables()
        x_noisy, used = ops.q_sample(x_start.to(torch.float32).contiguous(), t.to(torch.int64).contiguous(), dv["sqrt_ac"],
                                     dv["sqrt_1mac"], noise=noise.contiguous() if noise is not None else None,
                                     seed=_seed_from_torch() if noise is None else 0, row_offset=row_offset, return_noise=True)
        wants_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.model.parameters())
        drop_p = self.model.encoder[0].dropout.p if self.model.training else 0.0
        B = x_start.shape[0]
        if condition is not None:
            # per-row conditions for the taped forward: a [D] or [1, D] condition broadcasts over the batch, as t_emb + c_emb does (:766)
            condition = self.model.expand_condition(condition, B).to(device=x_noisy.device, dtype=torch.float32)
            condition = condition.expand(B, -1).contiguous()
        if not wants_grad and drop_p == 0.0:
            predicted_noise = self.model(x_noisy, t, condition)
            return torch.ops.adcdiff.mse_loss_grad(predicted_noise, used, 1.0)[0]   # the same kernel as the differentiable path
        rank, _ = _dp_rank_world()
        drop_seed = (_seed_from_torch() + rank) if drop_p > 0 else 0     # ranks seeded alike still draw different masks
        t64 = t.to(torch.int64).contiguous()
        if not wants_grad:   # value only, train mode: the taped forward with dropout, no backward
            return _taped_forward_loss(self.model, x_noisy, t64, used, drop_p, drop_seed, condition)[0]
        return _NoisePredictionLoss.apply(self.model, x_noisy, t64, used, drop_p, drop_seed, condition, *self.model.parameters())

    # ---- reverse process --------------------------------------------------------------------
    def p_sample(self, x: torch.Tensor, t: torch.Tensor, condition: Optional[torch.Tensor] = None,
                 noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One reverse step (diffusion_model.py:186-216).  Like the reference, which branches on t[0],
        this expects one timestep for the whole batch."""
        ts = int(t[0].item())
        if not bool((t == ts).all().item()):
            raise ValueError("p_sample expects a batch-uniform timestep (the reference branches on t[0] == 0)")
        predicted_noise = self.model(x, t, condition)
        c = self._ddpm_coef_array()[ts]
        return ops.ddpm_step(x.to(torch.float32).contiguous(), predicted_noise,
                             (c.sqrt_recip_alpha, c.beta, c.sqrt_one_minus_ac, c.sigma),
                             noise=noise.contiguous() if noise is not None else None,
                             seed=_seed_from_torch() if noise is None else 0, step=ts)

    def _sampling_setup(self, batch, condition):
        dev = self.model._device()
        eng = self.model.engine(self.compute_dtype)
        handle = eng.ensure(dev)
        cemb = None
        if condition is not None:
            # not moved to the device here: condition_embedding first collapses a stride-0 (one row expanded) condition to its
            # one row, so a CPU-resident expanded condition takes the same shared-row path as a GPU one (and copies one row)
            cemb = self.model.condition_embedding(condition, self.compute_dtype, batch=batch, broadcast_view=True)
        ws, ws_bytes = eng.workspace_for(batch, dev)
        return dev, eng, handle, cemb, ws, ws_bytes

    def p_sample_loop(self, shape: Tuple, condition: Optional[torch.Tensor] = None, *, x_T: Optional[torch.Tensor] = None,
                      noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, row_offset: int = 0,
                      t_start: Optional[int] = None, t_stop: int = 0) -> torch.Tensor:
        """The T-step reverse loop (diffusion_model.py:218-261) as ONE native call.

        Keyword-only extensions (defaults reproduce the reference's call): x_T gives the initial latent,
        noise a [steps, B, D] tensor of injected per-step noise (execution order), seed / row_offset
        select the counter-RNG stream of this batch (row_offset = index of its first row in a sharded
        job), t_start / t_stop bound the timestep range."""
        b, d = int(shape[0]), int(shape[1])
        dev, eng, handle, cemb, ws, ws_bytes = self._sampling_setup(b, condition)
        if seed is None:
            seed = _seed_from_torch()
        img = x_T.to(device=dev, dtype=torch.float32).contiguous() if x_T is not None else \
            ops.philox_normal(b, d, seed, _lib.STREAM_XT, 0, row_offset, dev)
        first = self.num_timesteps - 1 if t_start is None else int(t_start)
        stride = 0
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            stride = noise.stride(0)
        return torch.ops.adcdiff.ddpm_sample_loop(handle.value, img, cemb, self._ddpm_coef_tensor(), first, int(t_stop), noise, int(seed),
                                                  int(row_offset), eng.workspace)

    def _p_sample_loop_chunk(self, img: torch.Tensor, condition: Optional[torch.Tensor] = None) -> torch.Tensor:
        """diffusion_model.py:263-281 (the reference's version reads a missing attribute)."""
        return self.p_sample_loop(tuple(img.shape), condition, x_T=img)

    def sample(self, batch_size: int, condition: Optional[torch.Tensor] = None, use_ddim: bool = True, ddim_steps: int = 50,
               eta: float = 0.0, **kw) -> torch.Tensor:
        """diffusion_model.py:283-304."""
        with torch.no_grad():
            shape = (batch_size, self.output_dim)
            self.operation_count += 1
            if use_ddim:
                return self.ddim_sample_loop(shape, condition, ddim_steps, eta, **kw)
            return self.p_sample_loop(shape, condition, **kw)

    def _make_ddim_timesteps(self, ddim_steps: int) -> torch.Tensor:
        """diffusion_model.py:371-377."""
        c = self.num_timesteps // ddim_steps
        return (torch.arange(0, ddim_steps) * c + 1).to(self.device)

    def _ddim_plan(self, ddim_steps, eta):
        """(timesteps, coefficient array) in execution order; scalars evaluated on the host with the
        reference's float32 expressions (diffusion_model.py:315-316, 325, 351-363)."""
        c = self.num_timesteps // ddim_steps
        ts = torch.arange(0, ddim_steps) * c + 1
        prev = torch.cat([torch.tensor([0]), ts[:-1]])
        ac = self._host["alpha_cumprod"]
        n = ddim_steps
        steps = (C.c_int * n)()
        coef = (_lib.DdimCoef * n)()
        for i, (t, tp) in enumerate(zip(ts.flip(0).tolist(), prev.flip(0).tolist())):
            alpha_t, alpha_t_prev = ac[t], ac[tp]
            c_eps = torch.sqrt(1 - alpha_t)
            dir_coef = torch.sqrt(1 - alpha_t_prev - eta ** 2 * (1 - alpha_t_prev) / (1 - alpha_t) * (1 - alpha_t / alpha_t_prev))
            sigma = eta * torch.sqrt((1 - alpha_t_prev) / (1 - alpha_t)) * torch.sqrt(1 - alpha_t / alpha_t_prev) if eta > 0 \
                else torch.tensor(0.0)
            steps[i] = t
            coef[i] = _lib.DdimCoef(c_eps.item(), torch.sqrt(alpha_t).item(), torch.sqrt(alpha_t_prev).item(), dir_coef.item(),
                                    float(sigma))
        return steps, coef

    def ddim_sample_loop(self, shape: Tuple, condition: Optional[torch.Tensor] = None, ddim_steps: int = 50, eta: float = 0.0, *,
                         x_T: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None,
                         row_offset: int = 0) -> torch.Tensor:
        """DDIM loop (diffusion_model.py:306-340), the generator's default sampler, as one native call."""
        b, d = int(shape[0]), int(shape[1])
        dev, eng, handle, cemb, ws, ws_bytes = self._sampling_setup(b, condition)
        if seed is None:
            seed = _seed_from_torch()
        img = x_T.to(device=dev, dtype=torch.float32).contiguous() if x_T is not None else \
            ops.philox_normal(b, d, seed, _lib.STREAM_XT, 0, row_offset, dev)
        # the plan depends on (ddim_steps, eta) and the fixed schedule only: ~1 ms of host scalar arithmetic per call otherwise,
        # a tenth of the generator's DDIM-50 call at its default batch
        key = (int(ddim_steps), float(eta))
        plan = self._ddim_plan_cache.get(key)
        if plan is None:
            steps, coef = self._ddim_plan(ddim_steps, eta)
            plan = (torch.tensor(list(steps), dtype=torch.int32),
                    torch.tensor([[c.c_eps, c.sqrt_ac_t, c.sqrt_ac_prev, c.dir_coef, c.sigma] for c in coef], dtype=torch.float32))
            self._ddim_plan_cache[key] = plan
        steps_t, coef_t = plan
        stride = 0
        if noise is not None:
            noise = noise.to(device=dev, dtype=torch.float32).contiguous()
            stride = noise.stride(0)
        return torch.ops.adcdiff.ddim_sample_loop(handle.value, img, cemb, steps_t, coef_t, noise, int(seed), int(row_offset), eng.workspace)

    def _ddim_sample_chunked(self, shape: Tuple, condition: Optional[torch.Tensor] = None, ddim_steps: int = 50,
                             eta: float = 0.0) -> torch.Tensor:
        """diffusion_model.py:379-399: chunking is unnecessary here, the whole batch runs at once."""
        return self.ddim_sample_loop(shape, condition, ddim_steps, eta)

    def ddim_step(self, x: torch.Tensor, t: torch.Tensor, t_prev: torch.Tensor, condition: Optional[torch.Tensor] = None,
                  eta: float = 0.0, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """diffusion_model.py:342-369 for batch-uniform t / t_prev."""
        ts, tp = int(t[0].item()), int(t_prev[0].item())
        noise_pred = self.model(x, t, condition)
        ac = self._host["alpha_cumprod"]
        alpha_t, alpha_t_prev = ac[ts], ac[tp]
        dir_coef = torch.sqrt(1 - alpha_t_prev - eta ** 2 * (1 - alpha_t_prev) / (1 - alpha_t) * (1 - alpha_t / alpha_t_prev))
        sigma = float(eta * torch.sqrt((1 - alpha_t_prev) / (1 - alpha_t)) * torch.sqrt(1 - alpha_t / alpha_t_prev)) if eta > 0 else 0.0
        coef = (torch.sqrt(1 - alpha_t).item(), torch.sqrt(alpha_t).item(), torch.sqrt(alpha_t_prev).item(), dir_coef.item(), sigma)
        return ops.ddim_step(x.to(torch.float32).contiguous(), noise_pred, coef, noise=noise,
                             seed=_seed_from_torch() if (noise is None and eta > 0) else 0, step=ts)

    # ---- training -----------------------------------------------------------------------------
    def train_step(self, x: torch.Tensor, step: int = 0) -> Dict[str, float]:
        """diffusion_model.py:401-471: random timesteps, noise-prediction loss, backward, clip_grad_norm_(1.0), optimizer step,
        with the reference's gradient accumulation and chunking above max_batch_size.  Forward and backward of the denoiser
        run in the HIP library, and so do gradient clipping and the AdamW arithmetic (optim.FusedClipAdamW) on the state
        tensors of the torch optimizer object, whose state_dict is part of the reference's checkpoint format."""
        self.operation_count += 1
        self.model.train()
        batch_size = x.shape[0]
        if batch_size > self.max_batch_size:
            return self._train_step_chunked(x, step)
        x = x.to(self.device)
        rank, world = _dp_rank_world()
        # data parallel: every rank draws the timesteps of the global batch and keeps its own slice, and the forward-diffusion
        # noise is keyed by global row, so ranks seeded alike train on independent samples (one process: the plain draw)
        t = torch.randint(0, self.num_timesteps, (world * batch_size,), device=self.device)[rank * batch_size:(rank + 1) * batch_size]
        alias_before, self.model.alias_gradients = self.model.alias_gradients, True   # .grad = views of the persistent arena
        try:
            loss = self.p_losses(x, t, row_offset=rank * batch_size)
            # The step returns the loss as a float.  Its value exists as soon as the forward pass has run, long before the
            # backward pass and the optimizer have: it is copied to pinned host memory right here, behind the forward on the
            # stream, and the step waits for THAT copy only -- not for the rest of the step, which the GPU finishes while the
            # host prepares the next call.
            if self._loss_host is None:
                self._loss_host = torch.empty((), dtype=torch.float32).pin_memory()
                self._loss_event = torch.cuda.Event()
            self._loss_host.copy_(loss.detach(), non_blocking=True)
            self._loss_event.record()
            loss = loss / self.gradient_accumulation_steps
            loss.backward()
        finally:
            self.model.alias_gradients = alias_before
        if (step + 1) % self.gradient_accumulation_steps == 0:
            # data parallel (one process per GPU): average the gradients over the ranks before clipping, so that the
            # clipped norm and the AdamW step are identical on every rank; a no-op in a single process
            from ..parallel import allreduce_gradients
            allreduce_gradients(self.model.parameters(), flat=self.model.grad_arena_if_aliased())
            if self._fused_opt is None or self._fused_opt.optimizer is not self.optimizer:
                from ..optim import FusedClipAdamW
                self._fused_opt = FusedClipAdamW(self.optimizer, max_norm=1.0)
            self._fused_opt.step()     # clip_grad_norm_(max_norm=1.0) + optimizer.step() in three launches
            self.optimizer.zero_grad()
        memory_info = self._get_memory_usage()
        self._loss_event.synchronize()
        return {
            'loss': float(self._loss_host),
            'lr': self.optimizer.param_groups[0]['lr'],
            'gpu_memory': memory_info.get('gpu_memory_used_gb', 0),
            'gpu_memory_percent': memory_info.get('gpu_memory_percent', 0),
            'system_memory_percent': memory_info.get('system_memory_percent', 0),
        }

    def _train_step_chunked(self, x: torch.Tensor, step: int) -> Dict[str, float]:
        """diffusion_model.py:473-492: sequential chunks of max_batch_size, each its own train_step."""
        total_loss = 0.0
        num_chunks = (x.shape[0] + self.max_batch_size - 1) // self.max_batch_size
        for i in range(num_chunks):
            chunk = x[i * self.max_batch_size:min((i + 1) * self.max_batch_size, x.shape[0])]
            result = self.train_step(chunk, step * num_chunks + i)
            total_loss += result['loss'] * chunk.shape[0]
        return {'loss': total_loss / x.shape[0], 'lr': self.optimizer.param_groups[0]['lr'],
                'gpu_memory': torch.cuda.memory_allocated() / 1024 ** 3 if torch.cuda.is_available() else 0}

    # ---- checkpoint ---------------------------------------------------------------------------
    def save_model(self, path: str):
        """Same checkpoint dict as the reference (diffusion_model.py:521-532)."""
        torch.save({'model_state_dict': self.model.state_dict(), 'optimizer_state_dict': self.optimizer.state_dict(),
                    'config': self.config}, path)
        self.logger.info("model saved to %s", path)

    def load_model(self, path: str):
        """diffusion_model.py:534-543."""
        checkpoint = torch.load(path, map_location=self.device)
        self.model.load_state_dict(checkpoint['model_state_dict'])
        self.optimizer.load_state_dict(checkpoint['optimizer_state_dict'])
        self.logger.info("model loaded from %s", path)

    # ---- housekeeping of the reference's surface ------------------------------------------------
    def _get_memory_usage(self) -> Dict[str, float]:
        """diffusion_model.py:629-647."""
        info = {'system_memory_percent': psutil.virtual_memory().percent / 100.0}
        if torch.cuda.is_available():
            used = torch.cuda.memory_allocated() / 1024 ** 3
            total = torch.cuda.get_device_properties(0).total_memory / 1024 ** 3
            info['gpu_memory_percent'] = used / total if total > 0 else 0.0
            info['gpu_memory_used_gb'] = used
        return info

    def cleanup_resources(self):
        """diffusion_model.py:649-667."""
        for eng in self.model._engines.values():
            eng.destroy()
            eng.arena = eng.workspace = None
        self.model._engines.clear()
        if torch.cuda.is_available():
            torch.cuda.empty_cache()
        gc.collect()
