"""Row sharding of one sampling job over the GPUs of a node (one process per GPU).

Rows (samples) are independent through the whole reverse loop -- GroupNorm is per row and nothing in
sampling reduces over the batch -- so the only exchange is the collection of the finished latents: one
all-gather (RCCL over xGMI when the process group is "nccl", gloo in the CPU tests).  Every rank draws its
noise from the shared counter RNG at its GLOBAL row indices, so the gathered result equals the unsharded
run of the same seed.
"""
from typing import Callable, Optional, Tuple

import torch


def shard_rows(total: int, world: int, rank: int) -> Tuple[int, int]:
    """(first_row, count) of `rank`'s contiguous shard; the first total % world ranks hold one extra row."""
    if total < 0 or world <= 0 or not 0 <= rank < world:
        raise ValueError("bad shard request: total=%d world=%d rank=%d" % (total, world, rank))
    base, extra = divmod(total, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All-gather the ranks' shards (shapes from shard_rows) into the full [total, D] tensor on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # gloo (tests, rehearsals of the rank logic on one GPU) gathers device tensors through the host; RCCL takes them as they are
        return gather_rows(local.cpu(), total, group).to(local.device)
    d = local.shape[1]
    longest = shard_rows(total, world, 0)[1]
    if total % world == 0:
        out = torch.empty((total, d), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((longest, d), device=local.device, dtype=local.dtype)
    padded[:local.shape[0]] = local
    pieces = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(pieces, padded, group=group)
    return torch.cat([pieces[r][:shard_rows(total, world, r)[1]] for r in range(world)], dim=0)


def sample_sharded(sample_fn: Callable[[int, int], torch.Tensor], total: int, group=None) -> torch.Tensor:
    """Run sample_fn(count, first_row) on this rank's shard and gather all shards.

    sample_fn must key its randomness by global row (DiffusionModel.p_sample_loop / ddim_sample_loop do, via
    their row_offset argument); an empty shard (total < world) still takes part in the collective."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    first, count = shard_rows(total, world, rank)
    local = sample_fn(count, first)
    if world == 1:
        return local
    return gather_rows(local, total, group)


def sample_ddpm_sharded(model, total: int, condition: Optional[torch.Tensor] = None, seed: int = 0, use_ddim: bool = False,
                        ddim_steps: int = 50, eta: float = 0.0, group=None, summaries: bool = False) -> torch.Tensor:
    """Shard `total` samples of model.sample over the process group; condition is [total, D], [1, D] or [D].

    summaries=True gathers, instead of the [total, D] latents, the five numbers per row that the reference's
    latent -> SMILES heuristic reads (ops.latent_summary: mean, sum, x[3], x[5], x[8]): 20 bytes per sample over xGMI
    instead of 4 * D, and the same 20 bytes for the host copy that follows."""
    def run(count, first):
        if count == 0:
            return torch.empty((0, 5 if summaries else model.output_dim), device=model.device, dtype=torch.float32)
        if summaries:
            from . import ops
            return ops.latent_summary(run_latents(count, first))
        return run_latents(count, first)

    def run_latents(count, first):
        cond = None
        if condition is not None:
            cond = condition
            if cond.dim() == 1:
                cond = cond.unsqueeze(0)
            cond = cond.expand(count, -1) if cond.shape[0] == 1 else cond[first:first + count]   # a view: sample() sees the shared row
        return model.sample(count, condition=cond, use_ddim=use_ddim, ddim_steps=ddim_steps, eta=eta, seed=seed, row_offset=first)
    return sample_sharded(run, total, group)


def allreduce_gradients(parameters, group=None, bucket_bytes: int = 256 << 20, flat: Optional[torch.Tensor] = None) -> None:
    """Data-parallel gradient averaging: sum the ranks' `.grad` tensors and divide by the world size.

    flat: a buffer that every `.grad` is a view of (UNetModel.grad_arena_if_aliased(): the HIP backward writes all
    gradients into one arena) -- it is then reduced in place by ONE collective, with no packing or copy-back.

    Gradients are packed into flat float32 buckets (default 256 MiB: the whole 44.3 M-parameter denoiser is 177 MB, i.e.
    one all-reduce; xGMI is point-to-point, so few large ring collectives beat many small ones) and reduced with one
    collective per bucket.  Every rank must hold gradients for the same parameters.  No-op without a process group."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    if flat is not None:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        return
    params = [p for p in parameters if p.grad is not None]
    bucket, size = [], 0
    buckets = []
    for p in params:
        nbytes = p.grad.numel() * p.grad.element_size()
        if bucket and size + nbytes > bucket_bytes:
            buckets.append(bucket)
            bucket, size = [], 0
        bucket.append(p)
        size += nbytes
    if bucket:
        buckets.append(bucket)
    for bucket in buckets:
        flat = torch.cat([p.grad.reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.grad.numel()
            p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n
