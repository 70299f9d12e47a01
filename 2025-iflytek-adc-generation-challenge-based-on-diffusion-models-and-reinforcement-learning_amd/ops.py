"""Tensor-level wrappers over the C ABI (include/adcdiff.h).

torch is used here for device memory and the current stream only; every arithmetic step is a call
into libadcdiff.so.  All functions require CUDA (ROCm) tensors and raise otherwise: there is no CPU
path.

The public functions (gemm, groupnorm_silu, philox_normal, ddpm_step, ddim_step, q_sample, latent_summary) go through
the PyTorch dispatcher: they call the custom operators ``torch.ops.adcdiff.*`` registered in ``torch_ops.py``
(torch.library.custom_op with fake-tensor rules), whose implementations are the ``_raw_*`` ctypes wrappers below.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import BF16, F32, AdcdiffError, check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise AdcdiffError("adcdiff kernels run on an MI355X only; got a %s tensor (no CPU fallback exists)" % t.device)


def dtype_code(name):
    if name in (BF16, "bf16", "bfloat16", torch.bfloat16):
        return BF16
    if name in (F32, "fp32", "f32", "float32", torch.float32):
        return F32
    raise ValueError("compute dtype must be 'bf16' or 'fp32', got %r" % (name,))


def _pad_rows(n):
    return (n + 127) // 128 * 128


def _operand(x, code):
    """Copy a [B, K] float32 tensor into a row-padded operand buffer of the GEMM's type."""
    B, K = x.shape
    buf = torch.zeros((_pad_rows(B), K), device=x.device, dtype=torch.bfloat16 if code == BF16 else torch.float32)
    buf[:B].copy_(x)
    return buf


def _raw_gemm(a_list, w, bias=None, rowbias=None, rowbias_idx=None, residual=None, act=0, out_f32=True, out_op=False, gn=(),
         m_valid=None, dtype="fp32", eps=1e-5, tile_hint=0, split_k=0, update=None, splitk_scratch=None):
    """adcdiff_gemm: v = [A0|A1|A2] @ w.T + bias + rowbias + residual.

    splitk_scratch: a uint8 device tensor whose first 4096 bytes are zero (adcdiff_gemm_args.splitk_scratch): lets a bf16 launch
    with at most 64 live rows split K over workgroups.

    a_list: operand tensors [M, K_s] (M a multiple of 128) already in the operand dtype; w [N, sum K_s]; residual in the
    operand dtype; gn: up to two (gamma, beta, group_size) for fused SiLU(GroupNorm(v)) outputs in the operand dtype.
    split_k = S > 1: K is cut into S slices computed by separate workgroups; the float32 result then has shape [S, M, N]
    (the slices' partial products, to be summed by the caller).
    update = dict(kind=1 | 2, c=coefficients, x=latent [M, N] float32, noise=None | [M, N], seed, step, row_offset): the
    sampler's reverse step applied to v before the outputs are formed (adcdiff_step_update in include/adcdiff.h).
    Returns the requested outputs in the order (v float32, v operand type, gn outputs...), a single tensor if only one."""
    lib = _lib.load()
    code = dtype_code(dtype)
    td = torch.bfloat16 if code == BF16 else torch.float32
    _need_gpu(w, *a_list)
    M = a_list[0].shape[0]
    N = w.shape[0]
    args = _lib.GemmArgs()
    args.dtype, args.M, args.N, args.nseg = code, M, N, len(a_list)
    args.m_valid = 0 if m_valid is None else int(m_valid)
    for i, a in enumerate(a_list):
        args.A[i], args.lda[i], args.K[i] = a.data_ptr(), a.stride(0), a.shape[1]
    args.W, args.ldw = w.data_ptr(), w.stride(0)
    args.bias = bias.data_ptr() if bias is not None else None
    if rowbias is not None:
        args.rowbias = rowbias.data_ptr()
        args.ldrb = 0 if rowbias.dim() == 1 else rowbias.stride(0)
    args.rowbias_idx = rowbias_idx.data_ptr() if rowbias_idx is not None else None
    args.rowbias_rows = rowbias.shape[0] if (rowbias_idx is not None and rowbias is not None and rowbias.dim() == 2) else 0
    if residual is not None:
        args.residual, args.ldr = residual.data_ptr(), residual.stride(0)
    args.act = act
    args.tile_hint = tile_hint
    if splitk_scratch is not None:
        args.splitk_scratch, args.splitk_scratch_bytes = splitk_scratch.data_ptr(), splitk_scratch.numel()
    if tile_hint:   # forcing a kernel is a test / measurement request: the library honours hints only behind its debug gate
        _lib.debug_set("tile_hints", 1)
    outs = []
    if out_f32 is not None and out_f32 is not False:
        # a tensor is written in place (the latent of a fused schedule update)
        o = out_f32 if torch.is_tensor(out_f32) else \
            torch.empty((split_k, M, N) if split_k > 1 else (M, N), device=w.device, dtype=torch.float32)
        args.out_f32, args.ldo_f32 = o.data_ptr(), (o.stride(0) if torch.is_tensor(out_f32) else N)
        args.split_k, args.split_stride = int(split_k), M * N
        outs.append(o)
    if out_op:
        o = torch.empty((M, N), device=w.device, dtype=td)
        args.out_op, args.ldo_op = o.data_ptr(), N
        outs.append(o)
    args.n_gn = len(gn)
    args.gn_eps = eps
    keep = []
    for i, (gamma, beta, gs) in enumerate(gn):
        o = torch.empty((M, N), device=w.device, dtype=td)
        keep += [gamma, beta]
        args.gn[i].gamma, args.gn[i].beta, args.gn[i].out, args.gn[i].ldo, args.gn[i].group_size = \
            gamma.data_ptr(), beta.data_ptr(), o.data_ptr(), N, int(gs)
        outs.append(o)
    if update is not None:
        u = args.update
        u.kind = int(update["kind"])
        for i, cv in enumerate(update["c"]):
            u.c[i] = float(cv)
        ux, un = update["x"], update.get("noise")
        _need_gpu(ux)
        u.x, u.ldx = ux.data_ptr(), ux.stride(0)
        if un is not None:
            u.noise, u.ldn = un.data_ptr(), un.stride(0)
        u.seed, u.step, u.row_offset = int(update.get("seed", 0)), int(update.get("step", 0)), int(update.get("row_offset", 0))
    check(lib.adcdiff_gemm(C.byref(args), _stream()))
    return outs[0] if len(outs) == 1 else tuple(outs)


def _raw_gemm_tr(a, b, ta, slices=1):
    """adcdiff_gemm_tr: bf16 operands stored contraction-major.  b [K, N]; a [K, M] when ta else [M, K].
    Returns float32 [M, N], or [slices, M, N] partial products when slices > 1."""
    lib = _lib.load()
    _need_gpu(a, b)
    K, N = b.shape
    M = a.shape[1] if ta else a.shape[0]
    out = torch.empty((slices, M, N) if slices > 1 else (M, N), device=a.device, dtype=torch.float32)
    check(lib.adcdiff_gemm_tr(int(bool(ta)), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), N, M, N, K, int(slices),
                              M * N, _stream()))
    return out


def linear(x, weight, bias, act=0, dtype="fp32"):
    """nn.Linear on [B, K] float32 input (any B); returns [B, N] float32."""
    code = dtype_code(dtype)
    B = x.shape[0]
    a = _operand(x.contiguous(), code)
    w = weight.detach().to(torch.bfloat16 if code == BF16 else torch.float32).contiguous()
    out = gemm([a], w, bias=bias.detach() if bias is not None else None, act=act, m_valid=B, dtype=dtype)
    return out[:B]


def _raw_groupnorm_silu(x, groups, gamma, beta, eps=1e-5, out_dtype="fp32", with_raw=False):
    lib = _lib.load()
    _need_gpu(x, gamma, beta)
    code = dtype_code(out_dtype)
    M, Cc = x.shape
    td = torch.bfloat16 if code == BF16 else torch.float32
    out = torch.empty((M, Cc), device=x.device, dtype=td)
    raw = torch.empty((M, Cc), device=x.device, dtype=td) if with_raw else None
    check(lib.adcdiff_groupnorm_silu(_ptr(x), x.stride(0), M, Cc, groups, _ptr(gamma), _ptr(beta), eps, code, _ptr(out), Cc,
                                     _ptr(raw), Cc, _stream()))
    return (out, raw) if with_raw else out


def _raw_philox_normal(rows, cols, seed, stream_id, step, row_offset=0, device="cuda"):
    lib = _lib.load()
    out = torch.empty((rows, cols), device=device, dtype=torch.float32)
    _need_gpu(out)
    check(lib.adcdiff_philox_normal(_ptr(out), cols, rows, cols, seed, stream_id, step, row_offset, _stream()))
    return out


def _raw_ddpm_step(x, eps, coef, noise=None, seed=0, step=0, row_offset=0, out=None):
    lib = _lib.load()
    _need_gpu(x, eps, noise)
    B, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    c = _lib.DdpmCoef(*[float(v) for v in coef])
    check(lib.adcdiff_ddpm_step(_ptr(x), x.stride(0), _ptr(eps), eps.stride(0), _ptr(noise), noise.stride(0) if noise is not None else 0,
                                _ptr(out), out.stride(0), B, D, c, seed, step, row_offset, _stream()))
    return out


def _raw_ddim_step(x, eps, coef, noise=None, seed=0, step=0, row_offset=0, out=None):
    lib = _lib.load()
    _need_gpu(x, eps, noise)
    B, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    c = _lib.DdimCoef(*[float(v) for v in coef])
    check(lib.adcdiff_ddim_step(_ptr(x), x.stride(0), _ptr(eps), eps.stride(0), _ptr(noise), noise.stride(0) if noise is not None else 0,
                                _ptr(out), out.stride(0), B, D, c, seed, step, row_offset, _stream()))
    return out


def _raw_q_sample(x0, t, sqrt_ac, sqrt_1mac, noise=None, seed=0, step=0, row_offset=0, return_noise=False):
    lib = _lib.load()
    _need_gpu(x0, t, sqrt_ac, sqrt_1mac, noise)
    B, D = x0.shape
    out = torch.empty_like(x0)
    nz_out = torch.empty_like(x0) if return_noise else None
    check(lib.adcdiff_q_sample(_ptr(x0), x0.stride(0), _ptr(t), _ptr(sqrt_ac), _ptr(sqrt_1mac), sqrt_ac.numel(), _ptr(noise),
                               noise.stride(0) if noise is not None else 0, _ptr(out), D, _ptr(nz_out), D, B, D, seed, step,
                               row_offset, _stream()))
    return (out, nz_out) if return_noise else out


def _raw_latent_summary(x):
    """[B, D] float32 latents -> [B, 5] float32 (mean, sum, x[:, 3], x[:, 5], x[:, 8]) with numpy's float32 summation order."""
    lib = _lib.load()
    _need_gpu(x)
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("latent_summary expects a [B, D] float32 tensor with contiguous rows")
    B, D = x.shape
    out = torch.empty((B, 5), device=x.device, dtype=torch.float32)
    if B:
        check(lib.adcdiff_latent_summary(_ptr(x), x.stride(0), B, D, _ptr(out), 5, _stream()))
    return out


class timing:
    """Context manager over adcdiff_timing_begin/_end: ``with ops.timing(4096) as t: ...; t.records``."""

    def __init__(self, max_records):
        self.max_records = max_records
        self.records = []

    def __enter__(self):
        check(_lib.load().adcdiff_timing_begin(self.max_records))
        return self

    def __exit__(self, *exc):
        buf = (_lib.TimingRecord * self.max_records)()
        n = C.c_int(0)
        check(_lib.load().adcdiff_timing_end(buf, self.max_records, C.byref(n)))
        self.records = [dict(kind=r.kind, M=r.M, N=r.N, K=r.K, ms=r.ms, flops=r.flops, bytes=r.bytes) for r in buf[:n.value]]
        return False


# ---- public entry points: through torch.ops.adcdiff.* (registered in torch_ops.py) ----------------------------------
def gemm(a_list, w, bias=None, rowbias=None, rowbias_idx=None, residual=None, act=0, out_f32=True, out_op=False, gn=(),
         m_valid=None, dtype="fp32", eps=1e-5, tile_hint=0, split_k=0):
    """adcdiff_gemm through the dispatcher; arguments and results as _raw_gemm."""
    from . import torch_ops  # noqa: F401  (registers the operators)
    outs = torch.ops.adcdiff.gemm(list(a_list), w, bias, rowbias, rowbias_idx, residual, int(act), bool(out_f32), bool(out_op),
                                  [g[0] for g in gn], [g[1] for g in gn], [int(g[2]) for g in gn], 0 if m_valid is None else int(m_valid),
                                  dtype_code(dtype), float(eps), int(tile_hint), int(split_k))
    return outs[0] if len(outs) == 1 else tuple(outs)


def groupnorm_silu(x, groups, gamma, beta, eps=1e-5, out_dtype="fp32", with_raw=False):
    from . import torch_ops  # noqa: F401
    outs = torch.ops.adcdiff.groupnorm_silu(x, int(groups), gamma, beta, float(eps), dtype_code(out_dtype), bool(with_raw))
    return (outs[0], outs[1]) if with_raw else outs[0]


def philox_normal(rows, cols, seed, stream_id, step, row_offset=0, device="cuda"):
    from . import torch_ops  # noqa: F401
    return torch.ops.adcdiff.philox_normal(int(rows), int(cols), int(seed), int(stream_id), int(step), int(row_offset), torch.device(device))


def ddpm_step(x, eps, coef, noise=None, seed=0, step=0, row_offset=0, out=None):
    from . import torch_ops  # noqa: F401
    c = [float(v) for v in coef]
    res = torch.ops.adcdiff.ddpm_step(x, eps, noise, c[0], c[1], c[2], c[3], int(seed), int(step), int(row_offset))
    if out is not None:
        out.copy_(res)
        return out
    return res


def ddim_step(x, eps, coef, noise=None, seed=0, step=0, row_offset=0, out=None):
    from . import torch_ops  # noqa: F401
    c = [float(v) for v in coef]
    res = torch.ops.adcdiff.ddim_step(x, eps, noise, c[0], c[1], c[2], c[3], c[4], int(seed), int(step), int(row_offset))
    if out is not None:
        out.copy_(res)
        return out
    return res


def q_sample(x0, t, sqrt_ac, sqrt_1mac, noise=None, seed=0, step=0, row_offset=0, return_noise=False):
    from . import torch_ops  # noqa: F401
    out, used = torch.ops.adcdiff.q_sample(x0, t, sqrt_ac, sqrt_1mac, noise, int(seed), int(step), int(row_offset))
    return (out, used) if return_noise else out


def latent_summary(x):
    """[B, D] float32 latents -> [B, 5] float32 (mean, sum, x[:, 3], x[:, 5], x[:, 8]) with numpy's float32 summation order."""
    from . import torch_ops  # noqa: F401
    if x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("latent_summary expects a [B, D] float32 tensor with contiguous rows")
    return torch.ops.adcdiff.latent_summary(x)
