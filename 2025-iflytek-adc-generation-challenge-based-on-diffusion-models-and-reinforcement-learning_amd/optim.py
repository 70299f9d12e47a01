"""clip_grad_norm_ + AdamW / Adam step over all parameters in three launches of the HIP library.

Mirrors what the reference's train_step does with torch (diffusion_model.py:69-74, :434-436, :456-459):
``torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0); optimizer.step()``.  The torch optimizer object stays
the owner of the state (``exp_avg``, ``exp_avg_sq``, ``step`` per parameter, same ``state_dict`` as the reference's
checkpoints); this class only performs the arithmetic on those tensors.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class FusedClipAdamW:
    def __init__(self, optimizer, max_norm=1.0):
        if type(optimizer) not in (torch.optim.AdamW, torch.optim.Adam):
            raise TypeError("FusedClipAdamW drives torch.optim.AdamW / Adam, got %s" % type(optimizer).__name__)
        for g in optimizer.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                raise ValueError("amsgrad / maximize / capturable / differentiable are not supported")
        self.optimizer = optimizer
        self.max_norm = float(max_norm)
        self.decoupled = isinstance(optimizer, torch.optim.AdamW) or bool(optimizer.param_groups[0].get("decoupled_weight_decay", False))
        self._cache = {}           # param group index -> (key, descriptor table, workspace, chunks, grad_norm)
        self.grad_norm = None      # device scalar of the last group stepped: the norm before clipping, as clip_grad_norm_ returns it

    def _state_for(self, p):
        st = self.optimizer.state[p]
        if len(st) == 0:   # as torch's _init_group creates it on the first step
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _tables(self, gi, group):
        params = [p for p in group["params"] if p.grad is not None]
        if not params:
            return None
        for p in params:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise ValueError("parameters and gradients must be contiguous float32 tensors on the GPU")
        states = [self._state_for(p) for p in params]
        key = tuple((p.data_ptr(), p.grad.data_ptr(), s["exp_avg"].data_ptr(), s["exp_avg_sq"].data_ptr(), p.numel())
                    for p, s in zip(params, states))
        cached = self._cache.get(gi)
        if cached is None or cached[0] != key:
            rows = np.zeros((len(params), 6), dtype=np.int64)
            chunk = 0
            for i, k in enumerate(key):
                rows[i, :5] = k
                rows[i, 5] = chunk
                chunk += (k[4] + _lib.OPT_CHUNK - 1) // _lib.OPT_CHUNK
            assert C.sizeof(_lib.OptTensor) == 48
            dev = params[0].device
            need = _lib.load().adcdiff_clip_adamw_workspace_bytes(int(chunk))
            cached = (key, torch.from_numpy(rows).to(dev), torch.empty(need, dtype=torch.uint8, device=dev), int(chunk),
                      torch.zeros(1, dtype=torch.float32, device=dev))
            self._cache[gi] = cached
        return params, states, cached

    @torch.no_grad()
    def step(self):
        """One clipped update of every parameter group (the reference has one group)."""
        lib = _lib.load()
        for gi, group in enumerate(self.optimizer.param_groups):
            got = self._tables(gi, group)
            if got is None:
                continue
            params, states, (_, table, workspace, chunks, self.grad_norm) = got
            steps = [s["step"] for s in states]
            torch._foreach_add_(steps, 1)
            b1, b2 = group["betas"]
            args = _lib.AdamWArgs(float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                  self.max_norm, int(steps[0].item()), 1 if self.decoupled else 0)
            check(lib.adcdiff_clip_adamw_step(table.data_ptr(), len(params), chunks, C.byref(args), self.grad_norm.data_ptr(),
                                              workspace.data_ptr(), workspace.numel(), _stream()))
            # the library wrote the tensors behind torch's back: bump their version counters, which is how everything that
            # caches derived data (the packed operand copies of the denoiser, autograd's saved-tensor checks) sees a change
            torch.autograd.graph.increment_version(params)
            torch.autograd.graph.increment_version([s["exp_avg"] for s in states] + [s["exp_avg_sq"] for s in states])
        return self.grad_norm
