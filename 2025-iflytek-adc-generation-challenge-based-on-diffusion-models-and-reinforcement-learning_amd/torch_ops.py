"""PyTorch custom operators ``torch.ops.adcdiff.*`` over the C ABI (include/adcdiff.h).

Every entry point of the hot path is registered with ``torch.library.custom_op`` (so the dispatcher, ``torch.compile``'s
tracer and ``torch.library.opcheck`` see an operator with a schema and a fake-tensor rule instead of an opaque ctypes
call); the implementations call libadcdiff.so through the ctypes wrappers of ``ops.py`` / ``_lib.py``.  The operators
take and return tensors, ints and floats only:

* single kernels: ``gemm``, ``groupnorm_silu``, ``philox_normal``, ``ddpm_step``, ``ddim_step``, ``q_sample``,
  ``latent_summary``;
* the denoiser through a native plan handle (an int: the ``adcdiff_unet*`` of ``_Engine``): ``condition_embed``,
  ``unet_forward``, ``ddpm_sample_loop``, ``ddim_sample_loop``, ``unet_train_forward``, ``mse_loss_grad``,
  ``unet_backward``.  ``ws`` is the engine's workspace (a uint8 tensor the operator scribbles in: declared as mutated).
  Per-timestep coefficient tables travel as small CPU float32 tensors laid out like the C structs.

The training loss stays a ``torch.autograd.Function`` (models/diffusion_model.py) over ``unet_train_forward`` /
``mse_loss_grad`` / ``unet_backward``: its backward writes all parameter gradients into one arena that ``.grad`` aliases,
which an autograd formula returning one tensor per parameter could not do without a copy.
"""
import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib, ops
from ._lib import BF16, check

_op = torch.library.custom_op


def _td(code):
    return torch.bfloat16 if code == BF16 else torch.float32


# ---------------------------------------------------------------------------------------------- single kernels
@_op("adcdiff::gemm", mutates_args=())
def gemm(a: List[Tensor], w: Tensor, bias: Optional[Tensor], rowbias: Optional[Tensor], rowbias_idx: Optional[Tensor],
         residual: Optional[Tensor], act: int, out_f32: bool, out_op: bool, gn_gamma: List[Tensor], gn_beta: List[Tensor],
         gn_group_size: List[int], m_valid: int, dtype: int, eps: float, tile_hint: int, split_k: int) -> List[Tensor]:
    res = ops._raw_gemm(a, w, bias=bias, rowbias=rowbias, rowbias_idx=rowbias_idx, residual=residual, act=act, out_f32=out_f32,
                        out_op=out_op, gn=list(zip(gn_gamma, gn_beta, gn_group_size)), m_valid=m_valid or None,
                        dtype="bf16" if dtype == BF16 else "fp32", eps=eps, tile_hint=tile_hint, split_k=split_k)
    return list(res) if isinstance(res, tuple) else [res]


@gemm.register_fake
def _(a, w, bias, rowbias, rowbias_idx, residual, act, out_f32, out_op, gn_gamma, gn_beta, gn_group_size, m_valid, dtype, eps,
      tile_hint, split_k):
    M, N = a[0].shape[0], w.shape[0]
    outs = []
    if out_f32:
        outs.append(w.new_empty((split_k, M, N) if split_k > 1 else (M, N), dtype=torch.float32))
    if out_op:
        outs.append(w.new_empty((M, N), dtype=_td(dtype)))
    outs += [w.new_empty((M, N), dtype=_td(dtype)) for _ in gn_gamma]
    return outs


@_op("adcdiff::groupnorm_silu", mutates_args=())
def groupnorm_silu(x: Tensor, groups: int, gamma: Tensor, beta: Tensor, eps: float, out_dtype: int, with_raw: bool) -> List[Tensor]:
    res = ops._raw_groupnorm_silu(x, groups, gamma, beta, eps=eps, out_dtype="bf16" if out_dtype == BF16 else "fp32", with_raw=with_raw)
    return list(res) if with_raw else [res]


@groupnorm_silu.register_fake
def _(x, groups, gamma, beta, eps, out_dtype, with_raw):
    return [x.new_empty(x.shape, dtype=_td(out_dtype)) for _ in range(2 if with_raw else 1)]


@_op("adcdiff::philox_normal", mutates_args=())
def philox_normal(rows: int, cols: int, seed: int, stream_id: int, step: int, row_offset: int, device: torch.device) -> Tensor:
    return ops._raw_philox_normal(rows, cols, seed, stream_id, step, row_offset, device)


@philox_normal.register_fake
def _(rows, cols, seed, stream_id, step, row_offset, device):
    return torch.empty((rows, cols), device=device, dtype=torch.float32)


@_op("adcdiff::ddpm_step", mutates_args=())
def ddpm_step(x: Tensor, eps: Tensor, noise: Optional[Tensor], sqrt_recip_alpha: float, beta: float, sqrt_one_minus_ac: float,
              sigma: float, seed: int, step: int, row_offset: int) -> Tensor:
    return ops._raw_ddpm_step(x, eps, (sqrt_recip_alpha, beta, sqrt_one_minus_ac, sigma), noise=noise, seed=seed, step=step,
                              row_offset=row_offset)


@ddpm_step.register_fake
def _(x, eps, noise, sqrt_recip_alpha, beta, sqrt_one_minus_ac, sigma, seed, step, row_offset):
    return torch.empty_like(x)


@_op("adcdiff::ddim_step", mutates_args=())
def ddim_step(x: Tensor, eps: Tensor, noise: Optional[Tensor], c_eps: float, sqrt_ac_t: float, sqrt_ac_prev: float, dir_coef: float,
              sigma: float, seed: int, step: int, row_offset: int) -> Tensor:
    return ops._raw_ddim_step(x, eps, (c_eps, sqrt_ac_t, sqrt_ac_prev, dir_coef, sigma), noise=noise, seed=seed, step=step,
                              row_offset=row_offset)


@ddim_step.register_fake
def _(x, eps, noise, c_eps, sqrt_ac_t, sqrt_ac_prev, dir_coef, sigma, seed, step, row_offset):
    return torch.empty_like(x)


@_op("adcdiff::q_sample", mutates_args=())
def q_sample(x0: Tensor, t: Tensor, sqrt_ac: Tensor, sqrt_one_minus_ac: Tensor, noise: Optional[Tensor], seed: int, step: int,
             row_offset: int) -> Tuple[Tensor, Tensor]:
    """(x_t, the noise that was used)"""
    return ops._raw_q_sample(x0, t, sqrt_ac, sqrt_one_minus_ac, noise=noise, seed=seed, step=step, row_offset=row_offset, return_noise=True)


@q_sample.register_fake
def _(x0, t, sqrt_ac, sqrt_one_minus_ac, noise, seed, step, row_offset):
    return torch.empty_like(x0), torch.empty_like(x0)


@_op("adcdiff::latent_summary", mutates_args=())
def latent_summary(x: Tensor) -> Tensor:
    return ops._raw_latent_summary(x)


@latent_summary.register_fake
def _(x):
    return x.new_empty((x.shape[0], 5))


# ---------------------------------------------------------------------------------------------- the denoiser
def _h(handle):
    return C.c_void_p(handle)


def _ws(ws):
    return C.c_void_p(ws.data_ptr()), ws.numel()


@_op("adcdiff::condition_embed", mutates_args=("ws",))
def condition_embed(handle: int, cond: Tensor, hidden_dim: int, ws: Tensor) -> Tensor:
    B = cond.shape[0]
    cemb = torch.empty((B, hidden_dim), device=cond.device, dtype=torch.float32)
    p, n = _ws(ws)
    check(_lib.load().adcdiff_unet_condition_embed(_h(handle), ops._ptr(cond), cond.stride(0), B, ops._ptr(cemb), hidden_dim, p, n, ops._stream()))
    return cemb


@condition_embed.register_fake
def _(handle, cond, hidden_dim, ws):
    return cond.new_empty((cond.shape[0], hidden_dim))


@_op("adcdiff::unet_forward", mutates_args=("ws",))
def unet_forward(handle: int, x: Tensor, t: Tensor, cemb: Optional[Tensor], output_dim: int, ws: Tensor) -> Tensor:
    B = x.shape[0]
    eps = torch.empty((B, output_dim), device=x.device, dtype=torch.float32)
    p, n = _ws(ws)
    check(_lib.load().adcdiff_unet_forward(_h(handle), ops._ptr(x), x.stride(0), B, ops._ptr(t), 0, ops._ptr(cemb),
                                           cemb.stride(0) if cemb is not None else 0, ops._ptr(eps), output_dim, p, n, ops._stream()))
    return eps


@unet_forward.register_fake
def _(handle, x, t, cemb, output_dim, ws):
    return x.new_empty((x.shape[0], output_dim))


@_op("adcdiff::ddpm_sample_loop", mutates_args=("ws",))
def ddpm_sample_loop(handle: int, x_T: Tensor, cemb: Optional[Tensor], coef: Tensor, t_start: int, t_stop: int, noise: Optional[Tensor],
                     seed: int, row_offset: int, ws: Tensor) -> Tensor:
    """coef: CPU float32 [T, 4] = adcdiff_ddpm_coef per timestep; noise: [steps, B, D] or None.  Returns x after step t_stop."""
    if coef.device.type != "cpu" or coef.dtype != torch.float32 or not coef.is_contiguous() or coef.shape[1] != 4:
        raise ValueError("coef must be a contiguous CPU float32 [T, 4] tensor")
    img = x_T.clone()
    p, n = _ws(ws)
    check(_lib.load().adcdiff_ddpm_sample_loop(_h(handle), ops._ptr(img), img.stride(0), img.shape[0], ops._ptr(cemb),
                                               cemb.stride(0) if cemb is not None else 0,
                                               C.cast(C.c_void_p(coef.data_ptr()), C.POINTER(_lib.DdpmCoef)), t_start, t_stop, ops._ptr(noise),
                                               noise.stride(0) if noise is not None else 0, seed, row_offset, p, n, ops._stream()))
    return img


@ddpm_sample_loop.register_fake
def _(handle, x_T, cemb, coef, t_start, t_stop, noise, seed, row_offset, ws):
    return torch.empty_like(x_T)


@_op("adcdiff::ddim_sample_loop", mutates_args=("ws",))
def ddim_sample_loop(handle: int, x_T: Tensor, cemb: Optional[Tensor], timesteps: Tensor, coef: Tensor, noise: Optional[Tensor], seed: int,
                     row_offset: int, ws: Tensor) -> Tensor:
    """timesteps: CPU int32 [n] in execution order; coef: CPU float32 [n, 5] = adcdiff_ddim_coef per executed step."""
    if timesteps.device.type != "cpu" or timesteps.dtype != torch.int32 or coef.device.type != "cpu" or coef.dtype != torch.float32 or \
            not coef.is_contiguous() or coef.shape != (timesteps.numel(), 5):
        raise ValueError("timesteps must be CPU int32 [n] and coef contiguous CPU float32 [n, 5]")
    img = x_T.clone()
    p, n = _ws(ws)
    check(_lib.load().adcdiff_ddim_sample_loop(_h(handle), ops._ptr(img), img.stride(0), img.shape[0], ops._ptr(cemb),
                                               cemb.stride(0) if cemb is not None else 0,
                                               C.cast(C.c_void_p(timesteps.data_ptr()), C.POINTER(C.c_int)),
                                               C.cast(C.c_void_p(coef.data_ptr()), C.POINTER(_lib.DdimCoef)), timesteps.numel(), ops._ptr(noise),
                                               noise.stride(0) if noise is not None else 0, seed, row_offset, p, n, ops._stream()))
    return img


@ddim_sample_loop.register_fake
def _(handle, x_T, cemb, timesteps, coef, noise, seed, row_offset, ws):
    return torch.empty_like(x_T)


@_op("adcdiff::unet_train_forward", mutates_args=("ws",))
def unet_train_forward(handle: int, x: Tensor, t: Tensor, cond: Optional[Tensor], dropout_p: float, dropout_seed: int, output_dim: int,
                       ws: Tensor) -> Tensor:
    """Training-mode forward with per-row timesteps (and, optionally, per-row conditions [B, input_dim]); leaves the activations the
    backward pass needs in ws."""
    B = x.shape[0]
    eps = torch.empty((B, output_dim), device=x.device, dtype=torch.float32)
    p, n = _ws(ws)
    check(_lib.load().adcdiff_unet_train_forward(_h(handle), ops._ptr(x), x.stride(0), B, ops._ptr(t), ops._ptr(cond) if cond is not None else None,
                                                 cond.stride(0) if cond is not None else 0, dropout_p, dropout_seed, ops._ptr(eps), output_dim, p, n,
                                                 ops._stream()))
    return eps


@unet_train_forward.register_fake
def _(handle, x, t, cond, dropout_p, dropout_seed, output_dim, ws):
    return x.new_empty((x.shape[0], output_dim))


@_op("adcdiff::mse_loss_grad", mutates_args=())
def mse_loss_grad(pred: Tensor, target: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
    """(scale * mean((target - pred)^2), its gradient with respect to pred)"""
    B, D = pred.shape
    loss = torch.zeros((), device=pred.device, dtype=torch.float32)
    d = torch.empty_like(pred)
    check(_lib.load().adcdiff_mse_loss_grad(ops._ptr(pred), pred.stride(0), ops._ptr(target), target.stride(0), ops._ptr(d), d.stride(0), B, D,
                                            scale, ops._ptr(loss), ops._stream()))
    return loss, d


@mse_loss_grad.register_fake
def _(pred, target, scale):
    return pred.new_empty(()), torch.empty_like(pred)


@_op("adcdiff::unet_backward", mutates_args=("ws", "grad_arena"))
def unet_backward(handle: int, d_eps: Tensor, grad_pointers: Tensor, grad_arena: Tensor, ws: Tensor) -> None:
    """grad_pointers: CPU int64 tensor holding an adcdiff_unet_grads struct (device addresses inside grad_arena)."""
    p, n = _ws(ws)
    check(_lib.load().adcdiff_unet_backward(_h(handle), ops._ptr(d_eps), d_eps.stride(0),
                                            C.cast(C.c_void_p(grad_pointers.data_ptr()), C.POINTER(_lib.UnetGrads)), p, n, ops._stream()))


def names():
    return sorted(n for n in dir(torch.ops.adcdiff) if not n.startswith("_"))
