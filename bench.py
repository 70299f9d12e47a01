#!/usr/bin/env python3
"""Benchmark of the DDPM sampling hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload bf16_b65536|fp32_b4096]

One "step" is one full pass of the hot path over one batch: the T = 1000 reverse loop
(p_sample_loop: denoiser forward + fused schedule update per timestep) on `batch` synthetic samples per
GPU, emb 512 / hidden 256, random weights of that architecture, inputs generated on the device.  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank samples its own shard of rows
(counter-RNG row offset = rank * batch, so the job equals one N*batch-row run) and the latents are
collected by one RCCL all-gather inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import adcdiff  # noqa: E402
from adcdiff import _lib, ops  # noqa: E402

WORKLOADS = {
    # name: (compute dtype, per-GPU batch, description of the configuration)
    "bf16_b65536": ("bf16", 65536, "configs[2]: bf16 MFMA denoiser GEMMs, batch 65536 on 1 MI355X"),
    "fp32_b4096": ("fp32", 4096, "configs[1]: same denoiser, fp32 on 1 MI355X, batch 4096"),
}
DIMS = (512, 256, 512)
ALGO_FLOP_PER_SAMPLE_STEP = 81.92e6     # 2 * 40 960 000 MACs: denoiser linears with the time term hoisted
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
PEAK_HBM_GBS = 8000.0
SEED = 2025


def build_model(dtype, device):
    """The denoiser at emb 512 / hidden 256 with synthetic weights drawn on the device from the library's counter RNG
    (stream 100 + parameter index, so every rank builds the same tensors without a checkpoint): Linear weights
    N(0, 2 / (fan_in + fan_out)) -- the variance of the reference's xavier_uniform_ init --, biases N(0, 0.05^2), GroupNorm
    weight 1 + N(0, 0.1^2)."""
    cfg = dict(input_dim=DIMS[0], hidden_dim=DIMS[1], output_dim=DIMS[2], num_timesteps=1000, beta_start=1e-4, beta_end=2e-2,
               compute_dtype=dtype, memory_monitor_enabled=False)
    model = adcdiff.DiffusionModel(cfg)
    with torch.no_grad():
        for i, (name, p) in enumerate(model.model.named_parameters()):
            n = p.numel()
            z = ops.philox_normal(1, (n + 3) // 4 * 4, 1234, 100 + i, 0, device=device).reshape(-1)[:n].view_as(p)
            if p.dim() == 2:
                p.copy_(z * (2.0 / (p.shape[0] + p.shape[1])) ** 0.5)
            elif ".norm" in name and name.endswith("weight") or name == "output_norm.weight":
                p.copy_(1.0 + 0.1 * z)
            else:
                p.copy_(0.05 * z)
    return model


def measured_traffic(kernel_key, dtype, batch):
    """HBM bytes per launch of a kernel from the newest committed counter summary of this same workload (profiles/*_traffic.json,
    produced by profiles/collect.sh + summarize.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this benchmark)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
            line = d.get("bench_line_under_profiler", {})
            if line.get("dtype") != dtype or line.get("config", {}).get("per_gpu_batch") != batch:
                continue
            v = d["kernels"].get(kernel_key, {}).get("hbm_bytes_per_launch")
            if v:
                best = (v, os.path.basename(f))
        except Exception:
            pass
    return best


def roofline(model, dtype, batch, window=8):
    """Live per-launch hipEvent timing (on the stream the kernels run on) of `window` denoising steps.

    `dominant`: the kernel with the largest share of the step -- the 256 x 256-tile bf16 GEMM (bf16 workload) or the
    float32 GEMM -- achieved = FLOPs of its launches (2 M N K each, the layer as executed) / their device time.
    `gemms`: every GEMM launch of the step against the algorithmic 81.92 MFLOP per sample and step.
    Schedule-update and GroupNorm kernels: algorithmic bytes / device time."""
    with ops.timing(64 * window + 64) as tm:
        model.p_sample_loop((batch, DIMS[2]), seed=SEED, t_start=999, t_stop=1000 - window)
        torch.cuda.synchronize()
    gemm_kinds = (_lib.KIND_GEMM_BF16, _lib.KIND_GEMM_BF16_T256) if dtype == "bf16" else (_lib.KIND_GEMM_F32,)
    gemm = [r for r in tm.records if r["kind"] in gemm_kinds]
    big = [r for r in gemm if r["kind"] == _lib.KIND_GEMM_BF16_T256]
    dom, dom_name, dom_key = (big, "gemm256_bf16_kernel", "gemm256_kernel<bf16>") if big else \
        (gemm, "gemm_nt_kernel<%s>" % ("bf16" if dtype == "bf16" else "float"), "gemm_nt_kernel<%s>" % ("bf16" if dtype == "bf16" else "float"))
    upd = [r for r in tm.records if r["kind"] == _lib.KIND_UPDATE]
    gn = [r for r in tm.records if r["kind"] == _lib.KIND_GROUPNORM]
    peak = PEAK_TFLOPS[dtype]

    dom_ms = sum(r["ms"] for r in dom)
    dom_fl = sum(r["flops"] for r in dom)
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12
    main = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None, "launches": len(dom), "avg_launch_ms": round(dom_ms / max(1, len(dom)), 4),
            "flop_per_launch": dom_fl / max(1, len(dom)), "algorithmic_bytes_per_launch": sum(r["bytes"] for r in dom) / max(1, len(dom)),
            "share_of_gemm_time": round(dom_ms / max(1e-9, sum(r["ms"] for r in gemm)), 3)}
    tr = measured_traffic(dom_key, dtype, batch)
    if tr:
        main["traffic"], main["traffic_source"] = tr[0], "profiles/" + tr[1]
    gemm_ms = sum(r["ms"] for r in gemm)
    algo = ALGO_FLOP_PER_SAMPLE_STEP * batch * window
    allg = {"bound": "mfma", "kernel": "all denoiser GEMM launches of a step", "achieved": round(algo / (gemm_ms * 1e-3) / 1e12, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(algo / (gemm_ms * 1e-3) / 1e12 / peak, 4), "launches_per_step": len(gemm) // window,
            "algorithmic_flop_per_sample_step": ALGO_FLOP_PER_SAMPLE_STEP, "executed_flop_per_sample_step": sum(r["flops"] for r in gemm) / window / batch,
            "gemm_ms_per_denoise_step": round(gemm_ms / window, 4)}
    upd_ms = sum(r["ms"] for r in upd)
    upd_bytes = sum(r["bytes"] for r in upd)
    update = {"bound": "hbm", "kernel": "ddpm_step_kernel<false>", "achieved": round(upd_bytes / (upd_ms * 1e-3) / 1e9, 1),
              "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(upd_bytes / (upd_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
              "traffic": None, "launches": len(upd), "avg_launch_ms": round(upd_ms / max(1, len(upd)), 4),
              "bytes_per_launch": upd_bytes / max(1, len(upd))}
    tr = measured_traffic("ddpm_step_kernel", dtype, batch)
    if tr:
        update["traffic"], update["traffic_source"] = tr[0], "profiles/" + tr[1]
    gn_ms = sum(r["ms"] for r in gn)
    gn_bytes = sum(r["bytes"] for r in gn)
    norm = {"bound": "hbm", "kernel": "gn_silu_vec_kernel", "achieved": round(gn_bytes / (gn_ms * 1e-3) / 1e9, 1),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gn_bytes / (gn_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "launches": len(gn), "ms_per_denoise_step": round(gn_ms / window, 4)}
    return main, allg, update, norm


def cpu_baseline(weights, target_seconds=15.0):
    """The CPU restatement (oracle.ddpm, numpy float32 + OpenBLAS) timed on this host on a bounded
    sample of configs[0]: batch 32 (the reference's generation_batch_size), the first n of the 1000
    reverse steps, scaled to samples/s at T = 1000."""
    from oracle import ddpm
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    w = dict(weights)
    sched = ddpm.make_schedule(1000)
    B, D = 32, DIMS[0]
    x = ddpm.philox_x_T(SEED, 0, B, D)
    nf = ddpm.philox_noise_fn(SEED, 0, B, D)

    def run(n):
        t0 = time.perf_counter()
        ddpm.p_sample_loop(w, sched, x, None, nf, t_start=999, t_stop=1000 - n)
        return time.perf_counter() - t0

    run(3)
    probe = run(10) / 10
    n = int(max(20, min(1000, target_seconds / probe)))
    per_step = run(n) / n
    return {"value": round(B / (per_step * 1000), 4), "unit": "samples/s", "cores": int(cores), "kind": "port",
            "sample": "oracle.ddpm.p_sample_loop (numpy float32), emb 512/hidden 256, batch 32, %d of 1000 reverse steps "
                      "timed (%.1f ms/step) and scaled to T=1000" % (n, per_step * 1e3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="bf16_b65536")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--timesteps", type=int, default=1000, help="reverse steps per pass (1000 = the metric's T)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    dtype, batch, config_name = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    model = build_model(dtype, dev)
    T = args.timesteps
    gathered = torch.empty((world * batch, DIMS[2]), device=dev, dtype=torch.float32) if world > 1 else None

    def one_pass(k):
        out = model.p_sample_loop((batch, DIMS[2]), seed=SEED + k, row_offset=rank * batch, t_start=999, t_stop=1000 - T)
        if world > 1:
            dist.all_gather_into_tensor(gathered, out)
        return out

    for k in range(args.warmup):
        one_pass(-1 - k)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = one_pass(k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(torch.isfinite(out).all().item())

    if rank == 0:
        main_roof, all_roof, upd_roof, gn_roof = roofline(model, dtype, batch)
        samples = world * batch * args.steps * (T / 1000.0)
        line = {
            "metric": "DDPM samples/sec (T=1000)", "value": round(samples / elapsed, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": config_name, "denoiser": "emb 512 / hidden 256 / 44.3M params", "per_gpu_batch": batch,
                       "global_batch": world * batch, "timesteps": T, "sampler": "p_sample_loop (DDPM, linear beta 1e-4..2e-2)",
                       "parallelism": "rows sharded over %d GPU(s), RCCL all-gather of latents" % world,
                       "weights": "random (counter RNG, xavier variance), no checkpoint ships"},
            "roofline": main_roof, "roofline_all_gemms": all_roof, "roofline_schedule_update": upd_roof, "groupnorm_kernel": gn_roof,
            "ms_per_denoise_step": round(elapsed / args.steps / T * 1e3, 4), "outputs_finite": finite,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline({k: v.detach().cpu().numpy() for k, v in model.model.state_dict().items()})
            line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
