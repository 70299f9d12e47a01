#!/usr/bin/env python3
"""Benchmark of the DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME] [--condition] [--sampler ddpm|ddim50]

Workloads (emb 512 / hidden 256, random weights of that architecture, inputs generated on the device):

* ``bf16_b65536`` (default; BASELINE configs[2]), ``fp32_b4096`` (configs[1]) and ``bf16_b32`` (the generator's own call): one "step" is one full pass of the hot
  path over one batch -- the T = 1000 reverse loop (p_sample_loop: denoiser forward + fused schedule update per timestep)
  on `batch` samples per GPU.  With N > 1 every rank samples its own rows (counter-RNG row offset = rank * batch, so the
  job equals one N * batch-row run) and the latents are collected by one RCCL all-gather inside the timed region
  (weak scaling).  ``--condition`` adds the generator's one-row condition repeated to the batch
  (molecule_generator.py:238-245); ``--sampler ddim50`` times the generator's default sampler instead (50 DDIM steps).
* ``shard_1m`` (configs[3]): 1 048 576 samples in total, sharded by rows over the N ranks (parallel.shard_rows), bf16,
  all-gather of the [1048576, 512] latents in the timed region; strong scaling.
* ``train_dp`` (configs[4]): one "step" is one train_step (q_sample, denoiser forward + backward in the HIP library,
  gradient all-reduce over the ranks, fused clip + AdamW) at batch 4096 per GPU, bf16 operands / fp32 master weights.

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# torch, numpy and the package are imported by load_runtime(), in the worker processes only: the process that starts the
# ranks of a multi-GPU run (launch_ranks) must never initialise a GPU.
np = torch = adcdiff = _lib = ops = parallel = None


def load_runtime(need_library=True):
    global np, torch, adcdiff, _lib, ops, parallel
    import numpy as np_
    import torch as torch_
    np, torch = np_, torch_
    if need_library:
        import adcdiff as adcdiff_
        from adcdiff import _lib as lib_, ops as ops_, parallel as parallel_
        adcdiff, _lib, ops, parallel = adcdiff_, lib_, ops_, parallel_
    else:
        # the launcher self-test needs the row-sharding helpers only (pure torch.distributed, no native library)
        import importlib.util
        pkg = os.path.join(ROOT, "2025-iflytek-adc-generation-challenge-based-on-diffusion-models-and-reinforcement-learning_amd")
        spec = importlib.util.spec_from_file_location("adcdiff_parallel_only", os.path.join(pkg, "parallel.py"))
        parallel = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(parallel)

WORKLOADS = {
    # name: (compute dtype, per-GPU batch, description of the configuration)
    "bf16_b65536": ("bf16", 65536, "configs[2]: bf16 MFMA denoiser GEMMs, batch 65536 on 1 MI355X"),
    "fp32_b4096": ("fp32", 4096, "configs[1]: same denoiser, fp32 on 1 MI355X, batch 4096"),
    "shard_1m": ("bf16", 1048576, "configs[3]: 1M samples sharded over the GPUs, RCCL all-gather of latents"),
    "train_dp": ("bf16", 4096, "configs[4]: denoiser train step (q_sample noise-pred loss fwd+bwd), data parallel, batch 4096 per GPU"),
    "bf16_b32": ("bf16", 32, "the generator's own call (molecule_generator.py:238-262): 32 latents per call, bf16"),
    "launcher_selftest": ("f32", 1000, "rank start-up, row sharding and gather on the CPU (gloo); no GPU work"),
}
DIMS = (512, 256, 512)
ALGO_FLOP_PER_SAMPLE_STEP = 81.92e6     # 2 * 40 960 000 MACs: denoiser linears with the time term hoisted
ALGO_FLOP_PER_SAMPLE_STEP_COND = 87.03e6   # + 2 * 2 555 904: the ten time_proj layers are per row with a condition
TRAIN_FLOP_PER_SAMPLE = 3 * 88.08e6     # forward as executed (per-row timesteps: nothing hoisted) + two backward GEMMs per layer
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}
PEAK_HBM_GBS = 8000.0
SEED = 2025


def build_model(dtype, device, **extra):
    """The denoiser at emb 512 / hidden 256 with synthetic weights drawn on the device from the library's counter RNG
    (stream 100 + parameter index, so every rank builds the same tensors without a checkpoint): Linear weights
    N(0, 2 / (fan_in + fan_out)) -- the variance of the reference's xavier_uniform_ init --, biases N(0, 0.05^2), GroupNorm
    weight 1 + N(0, 0.1^2)."""
    cfg = dict(input_dim=DIMS[0], hidden_dim=DIMS[1], output_dim=DIMS[2], num_timesteps=1000, beta_start=1e-4, beta_end=2e-2,
               compute_dtype=dtype, memory_monitor_enabled=False, **extra)
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


def measured_traffic(kernel_key, workload):
    """HBM bytes per launch of a kernel from the newest committed counter summary of this same workload (profiles/*_traffic.json,
    produced by profiles/collect.sh + summarize.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this benchmark)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
            line = d.get("bench_line_under_profiler", {})
            cfg = line.get("config", {})
            name = cfg.get("workload_name") or ("bf16_b65536" if line.get("dtype") == "bf16" and cfg.get("per_gpu_batch") == 65536 else
                                                "fp32_b4096" if line.get("dtype") == "fp32" and cfg.get("per_gpu_batch") == 4096 else None)
            if name != workload or cfg.get("condition") or cfg.get("sampler_name", "ddpm") != "ddpm":
                continue
            v = d["kernels"].get(kernel_key, {}).get("hbm_bytes_per_launch")
            if v:
                best = (v, os.path.basename(f))
        except Exception:
            pass
    return best


def timed_window(fn):
    """Run fn() twice on the current stream: once between ONE pair of events (the window's true device time) and once with
    an event in front of every launch (ops.timing).  Recording an event costs the stream a few microseconds, so the second
    run's spans add up to more than the first run's time; the excess is taken off every span in equal parts (it is a
    per-event cost), which makes the records add up to the un-instrumented window exactly."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    clean_ms = e0.elapsed_time(e1)
    with ops.timing(4096) as tm:
        fn()
        torch.cuda.synchronize()
    recs = tm.records
    over = (sum(r["ms"] for r in recs) - clean_ms) / max(1, len(recs))
    for r in recs:
        r["ms"] = max(r["ms"] - over, 1e-6)
    return recs, clean_ms, over


def summarize_records(records, dtype, units, algo_flop_per_unit, workload):
    """Roofline objects from per-launch records (timed_window): `units` = samples x denoising steps (or training samples)
    inside the window.  The record times add up to the window's device time."""
    gemm_kinds = (_lib.KIND_GEMM_BF16, _lib.KIND_GEMM_BF16_T256, _lib.KIND_GEMM_BF16_W256, _lib.KIND_GEMM_F32)
    gemm = [r for r in records if r["kind"] in gemm_kinds]
    w4 = [r for r in gemm if r["kind"] == _lib.KIND_GEMM_BF16_W256]
    big = [r for r in gemm if r["kind"] == _lib.KIND_GEMM_BF16_T256]
    small_name = "gemm_nt_kernel<%s>" % ("bf16" if dtype == "bf16" else "float")
    if workload == "bf16_b32":
        # a batch of 32 streams the weights once per step through the few-row kernels: bound by bytes, not by the matrix cores
        dom_ms, dom_by = sum(r["ms"] for r in gemm), sum(r["bytes"] for r in gemm)
        window_ms = sum(r["ms"] for r in records)
        main = {"bound": "hbm", "kernel": "gemm_w1_bf16_kernel / gemm_cl_bf16_kernel", "achieved": round(dom_by / (dom_ms * 1e-3) / 1e9, 1),
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(dom_by / (dom_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "traffic": None,
                "launches": len(gemm), "avg_launch_ms": round(dom_ms / max(1, len(gemm)), 4),
                "algorithmic_bytes_per_launch": dom_by / max(1, len(gemm)), "share_of_window": round(dom_ms / max(1e-9, window_ms), 3),
                "timing": "hipEvent in front of every launch on the launch stream, span to the next launch's event, minus the per-event cost "
                          "measured against the same window between one event pair"}
        return {"roofline": main}
    if w4:
        dom, dom_name, dom_key = w4, "gemm256w_bf16_kernel", "gemm256w_kernel<bf16>"
    elif big:
        dom, dom_name, dom_key = big, "gemm256_bf16_kernel", "gemm256_kernel<bf16>"
    else:
        dom, dom_name, dom_key = gemm, small_name, small_name
    peak = PEAK_TFLOPS[dtype]
    window_ms = sum(r["ms"] for r in records)
    dom_ms = sum(r["ms"] for r in dom)
    dom_fl = sum(r["flops"] for r in dom)
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12
    main = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None, "launches": len(dom), "avg_launch_ms": round(dom_ms / max(1, len(dom)), 4),
            "flop_per_launch": dom_fl / max(1, len(dom)), "algorithmic_bytes_per_launch": sum(r["bytes"] for r in dom) / max(1, len(dom)),
            "share_of_window": round(dom_ms / max(1e-9, window_ms), 3),
            "timing": "hipEvent in front of every launch on the launch stream, span to the next launch's event, minus the per-event cost "
                      "measured against the same window between one event pair"}
    # the dominant kernel by layer shape (it runs layers of very different arithmetic intensity): the three shapes that take the most time
    shapes = {}
    for r in dom:
        k = (r["N"], r["K"])
        a = shapes.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1; a[1] += r["ms"]; a[2] += r["flops"]
    main["by_shape"] = [{"N": k[0], "K": k[1], "launches": a[0], "avg_launch_ms": round(a[1] / a[0], 4),
                         "achieved": round(a[2] / (a[1] * 1e-3) / 1e12, 1), "frac": round(a[2] / (a[1] * 1e-3) / 1e12 / peak, 4),
                         "share_of_window": round(a[1] / max(1e-9, window_ms), 3)}
                        for k, a in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:3]]
    tr = measured_traffic(dom_key, workload)
    if tr:
        main["traffic"], main["traffic_source"] = tr[0], "profiles/" + tr[1]
    gemm_ms = sum(r["ms"] for r in gemm)
    algo = algo_flop_per_unit * units
    allg = {"bound": "mfma", "kernel": "all GEMM launches of the window", "achieved": round(algo / (gemm_ms * 1e-3) / 1e12, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(algo / (gemm_ms * 1e-3) / 1e12 / peak, 4), "launches": len(gemm),
            "algorithmic_flop_per_unit": algo_flop_per_unit, "executed_flop_per_unit": sum(r["flops"] for r in gemm) / units,
            "gemm_ms": round(gemm_ms, 4), "window_ms": round(window_ms, 4), "gemm_share_of_window": round(gemm_ms / max(1e-9, window_ms), 3)}
    out = {"roofline": main, "roofline_all_gemms": allg}
    for kind, key, kname in ((_lib.KIND_UPDATE, "roofline_schedule_update", "ddpm_step_kernel / ddim_step_kernel"),
                             (_lib.KIND_GROUPNORM, "groupnorm_kernel", "gn_silu_vec_kernel"), (_lib.KIND_OTHER, "other_kernels", "casts, transposes, element-wise backward, optimizer")):
        rs = [r for r in records if r["kind"] == kind]
        if not rs:
            continue
        ms, by = sum(r["ms"] for r in rs), sum(r["bytes"] for r in rs)
        out[key] = {"bound": "hbm", "kernel": kname, "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(by / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4), "launches": len(rs), "avg_launch_ms": round(ms / len(rs), 4),
                    "share_of_window": round(ms / max(1e-9, window_ms), 3)}
    return out


def fp32_side_result(dev):
    """BASELINE configs[1] (the same denoiser in exact float32, batch 4096, T = 1000) beside the default line: one untimed
    100-step run, then ONE full reverse loop timed between synchronisations, and the live roofline of its GEMMs."""
    model = build_model("fp32", dev)
    B = 4096
    model.p_sample_loop((B, DIMS[2]), None, seed=SEED, t_start=999, t_stop=900)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.p_sample_loop((B, DIMS[2]), None, seed=SEED + 1)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    recs, clean_ms, over = timed_window(lambda: model.p_sample_loop((B, DIMS[2]), None, seed=SEED, t_start=999, t_stop=992))
    roofs = summarize_records(recs, "fp32", B * 8, ALGO_FLOP_PER_SAMPLE_STEP, "fp32_b4096")
    return {"metric": "DDPM samples/sec (T=1000)", "value": round(B / el, 2), "unit": "samples/s", "dtype": "fp32", "steps": 1,
            "ms_per_step": round(el * 1e3, 2), "config": {"workload": WORKLOADS["fp32_b4096"][2], "per_gpu_batch": B, "timesteps": 1000},
            "outputs_finite": bool(torch.isfinite(out).all().item()), "roofline": roofs["roofline"],
            "roofline_all_gemms": roofs["roofline_all_gemms"]}


def cpu_model_name():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def cpu_baseline(weights, target_seconds=12.0):
    """The CPU restatement (oracle.ddpm, numpy float32 + OpenBLAS) timed on this host on a bounded sample of configs[0]:
    batch 32 (the reference's generation_batch_size), the first n of the 1000 reverse steps, scaled to samples/s at
    T = 1000; plus a batch-4096 run of a few steps as the best case a CPU gets from batching."""
    from oracle import ddpm
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    w = dict(weights)
    sched = ddpm.make_schedule(1000)
    D = DIMS[0]

    def rate(B, budget, min_steps):
        x = ddpm.philox_x_T(SEED, 0, B, D)
        nf = ddpm.philox_noise_fn(SEED, 0, B, D)

        def run(n):
            t0 = time.perf_counter()
            ddpm.p_sample_loop(w, sched, x, None, nf, t_start=999, t_stop=1000 - n)
            return time.perf_counter() - t0
        run(2)
        probe = run(4) / 4
        n = int(max(min_steps, min(1000, budget / probe)))
        per_step = run(n) / n
        return B / (per_step * 1000), n, per_step

    v32, n32, ps32 = rate(32, target_seconds, 20)
    v4k, n4k, ps4k = rate(4096, target_seconds * 0.6, 3)
    return {"value": round(v32, 4), "unit": "samples/s", "cores": int(cores), "kind": "port", "cpu_model": cpu_model_name(),
            "sample": "oracle.ddpm.p_sample_loop (numpy float32), emb 512/hidden 256, batch 32, %d of 1000 reverse steps "
                      "timed (%.1f ms/step) and scaled to T=1000" % (n32, ps32 * 1e3),
            "best_case_batch_4096": {"value": round(v4k, 3), "unit": "samples/s",
                                     "sample": "same, batch 4096, %d steps timed (%.0f ms/step)" % (n4k, ps4k * 1e3)}}


def launch_ranks(n):
    """`python bench.py --gpus N` with no rendezvous in the environment: start the N ranks as a CHILD process
    (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) with this command line, pass their output through and
    return their exit code.  This process has imported neither torch nor the library and never touches a GPU; it is not
    replaced by the child (no exec)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "gloo" in sys.argv:
        env.setdefault("GLOO_SOCKET_IFNAME", "lo")   # the rehearsal backend picks its interface by host name, which may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def launcher_selftest(args, rank, world):
    """The rank plumbing of a multi-GPU run on the CPU (gloo), with no GPU and no native library: every rank fills its row
    shard with a function of the GLOBAL row index, the shards are gathered as the sampling workloads gather their latents,
    and the line has the same barrier / max-over-ranks timing.  Not a measurement; tests/test_bench_launcher_cpu.py runs it."""
    load_runtime(need_library=False)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    total, d = args.batch or 1000, 8
    first, count = parallel.shard_rows(total, world, rank)

    def one_pass(k):
        rows = (torch.arange(first, first + count, dtype=torch.float32)[:, None] * d + torch.arange(d, dtype=torch.float32)[None, :]) + k
        return parallel.gather_rows(rows, total) if world > 1 else rows
    for k in range(args.warmup):
        one_pass(-1 - k)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = one_pass(k)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    want = torch.arange(total * d, dtype=torch.float32).reshape(total, d) + (args.steps - 1)
    ok = bool(out.shape == want.shape and torch.equal(out, want))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test rows/sec (CPU, gloo; not a measurement)", "value": round(total * args.steps / elapsed, 1),
                          "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "launcher_selftest", "rows": total},
                          "ranks": dist.get_world_size() if world > 1 else 1, "gathered_rows_match": ok}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="bf16_b65536")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch override (shard_1m: total rows)")
    ap.add_argument("--timesteps", type=int, default=1000, help="reverse steps per pass (1000 = the metric's T)")
    ap.add_argument("--condition", action="store_true", help="conditional sampling: a one-row condition repeated to the batch")
    ap.add_argument("--sampler", choices=["ddpm", "ddim50"], default="ddpm")
    ap.add_argument("--debug", action="append", default=[], metavar="KEY=VALUE", help="a measurement switch of the library (adcdiff_debug_set), for A/B runs")
    ap.add_argument("--backend", default="nccl", help="process-group backend: nccl (= RCCL, the measurement) or gloo (rehearsal of the rank "
                    "logic: the ranks may then share one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32", action="store_true", help="default workload only: skip the float32 batch-4096 (configs[1]) side result")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(args.gpus))
    if os.environ.get("BENCH_WATCHDOG"):   # all threads' stacks to stderr after that many seconds (a hung rehearsal says where)
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["BENCH_WATCHDOG"]), exit=False)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the two must agree (run `python bench.py --gpus N`, which starts its own "
                         "ranks, or torch.distributed.run --nproc-per-node N)" % (args.gpus, world))
    if args.workload == "launcher_selftest":
        return launcher_selftest(args, rank, world)
    load_runtime()
    for kv in args.debug:
        _lib.debug_set(kv.split("=")[0], int(kv.split("=")[1]))
    dev_index = local_rank if args.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    def mark(what):
        if os.environ.get("BENCH_PROGRESS"):
            print("[bench rank %d] %s" % (rank, what), file=sys.stderr, flush=True)
    mark("process group up")
    dtype, batch, config_name = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    T = args.timesteps
    training = args.workload == "train_dp"
    model = build_model(dtype, dev, **(dict(max_batch_size=batch, learning_rate=1e-4) if training else {}))
    cond = ops.philox_normal(1, DIMS[0], SEED, 51, 0, device=dev) if args.condition else None   # [1, 512]: broadcast to the batch

    if args.workload == "shard_1m":
        total = batch
        first, count = parallel.shard_rows(total, world, rank)
        scaling = "strong"
    else:
        total, first, count, scaling = world * batch, rank * batch, batch, "weak"

    if training:
        x_train = ops.philox_normal(count, DIMS[0], 7, 52, 0, first, dev)
        torch.manual_seed(1234)

        def one_pass(k):
            return model.train_step(x_train, k)
    else:
        def one_pass(k):
            if args.sampler == "ddim50":
                out = model.ddim_sample_loop((count, DIMS[2]), cond, 50, 0.0, seed=SEED + k, row_offset=first)
            else:
                out = model.p_sample_loop((count, DIMS[2]), cond, seed=SEED + k, row_offset=first, t_start=999, t_stop=1000 - T)
            if world > 1:
                out = parallel.gather_rows(out, total)
            return out

    mark("model built")
    for k in range(args.warmup):
        one_pass(-1 - k)
    mark("warm-up done")
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    mark("barrier passed")
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = one_pass(k)
        mark("step %d enqueued" % k)
    torch.cuda.synchronize()
    mark("device idle")
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    mark("timed steps done")
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    finite = bool(np.isfinite(out["loss"])) if training else bool(torch.isfinite(out).all().item())
    collective = None
    if world > 1:
        # the job's one exchange, timed alone on every rank (max over ranks): the all-gather of the latents or the all-reduce
        # of the gradient arena; it is inside `elapsed` above as well
        if training:
            flat = model.model._grad_flat[0] if model.model._grad_flat is not None else None   # the arena every .grad aliased during the steps
            fn_c = (lambda: parallel.allreduce_gradients(model.model.parameters(), flat=flat)) if flat is not None else None
            nbytes = flat.numel() * 4 if flat is not None else 0
            name = "RCCL all-reduce of the float32 gradient arena"
        else:
            local = out[first:first + count].contiguous()
            fn_c = lambda: parallel.gather_rows(local, total)
            nbytes, name = total * DIMS[2] * 4, "RCCL all-gather of the [global_batch, 512] float32 latents"
        if fn_c is not None:
            fn_c()
            dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                fn_c()
            e1.record()
            torch.cuda.synchronize()
            tc = torch.tensor([e0.elapsed_time(e1) / 3], device=dev, dtype=torch.float64)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            collective = {"what": name, "ms": round(float(tc.item()), 4), "bytes_result_per_rank": nbytes,
                          "share_of_step": round(float(tc.item()) * 1e-3 / (elapsed / args.steps), 5)}

    # live roofline: per-launch events over a short window of the same work.  A training step holds a collective (the gradient
    # all-reduce), so with several ranks every rank runs the window's steps; rank 0 alone keeps the records
    train_window = timed_window(lambda: model.train_step(x_train, 10 ** 6)) if training else None
    if rank == 0:
        if training:
            recs, clean_ms, over = train_window
            roofs = summarize_records(recs, dtype, count, TRAIN_FLOP_PER_SAMPLE, args.workload)
            samples = total * args.steps
            metric, sampler_desc = "train samples/sec (noise-prediction loss fwd+bwd + clip + AdamW)", "train_step"
            per_unit_ms = None
        else:
            window = 50 if args.sampler == "ddim50" else 8
            wb = min(count, 65536)
            if args.sampler == "ddim50":
                fn = lambda: model.ddim_sample_loop((wb, DIMS[2]), cond, 50, 0.0, seed=SEED, row_offset=first)
            else:
                fn = lambda: model.p_sample_loop((wb, DIMS[2]), cond, seed=SEED, row_offset=first, t_start=999, t_stop=1000 - window)
            recs, clean_ms, over = timed_window(fn)
            # --condition passes ONE condition row for the batch (the generator's call): its time projection is a row bias, the
            # step executes the unconditional 81.92 MFLOP and is credited with that; 87.03 applies to B distinct condition rows
            roofs = summarize_records(recs, dtype, wb * window, ALGO_FLOP_PER_SAMPLE_STEP, args.workload)
            nsteps = 50 if args.sampler == "ddim50" else T
            samples = total * args.steps * (1.0 if args.sampler == "ddim50" else T / 1000.0)
            metric = "DDPM samples/sec (T=1000)" if args.sampler == "ddpm" else "DDIM-50 samples/sec"
            sampler_desc = "p_sample_loop (DDPM, linear beta 1e-4..2e-2)" if args.sampler == "ddpm" else "ddim_sample_loop (50 steps, eta 0)"
            per_unit_ms = round(elapsed / args.steps / nsteps * 1e3, 4)
        roofs["roofline"]["window"] = {"clean_ms": round(clean_ms, 4), "launches": len(recs), "per_event_cost_us": round(over * 1e3, 2)}
        if roofs["roofline"]["bound"] == "mfma" and dtype == "bf16":
            # the data sheet's peak assumes the top clock; what the matrix cores sustain inside the power limit on register operands
            # with random bits (no memory traffic) is measured here, on this GPU, right behind the timed region
            tf, mhz = _lib.matrix_core_probe(20000, 24)
            roofs["roofline"]["sustained_matrix_rate"] = {
                "what": "v_mfma_f32_16x16x32_bf16 on register operands, one wave per SIMD on every CU, ~0.4 s",
                "tflops": round(tf, 1), "shader_clock_mhz": round(mhz), "frac_of_peak": round(tf / roofs["roofline"]["peak"], 4),
                "achieved_over_sustained": round(roofs["roofline"]["achieved"] / tf, 4)}
        line = {
            "metric": metric, "value": round(samples / elapsed, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": config_name, "workload_name": args.workload, "denoiser": "emb 512 / hidden 256 / 44.3M params",
                       "per_gpu_batch": count, "global_batch": total, "timesteps": T, "sampler": sampler_desc, "sampler_name": args.sampler,
                       "condition": bool(args.condition),
                       "parallelism": ("rows sharded over %d GPU(s), RCCL all-gather of latents" % world) if not training else
                                      ("data parallel over %d GPU(s), RCCL all-reduce of the gradient arena" % world),
                       "weights": "random (counter RNG, xavier variance), no checkpoint ships"},
            "outputs_finite": finite, "rccl_ranks": dist.get_world_size() if world > 1 else 1, "backend": args.backend if world > 1 else None,
        }
        if collective:
            line["collective"] = collective
        line.update(roofs)
        if per_unit_ms is not None:
            line["ms_per_denoise_step"] = per_unit_ms
        if world == 1 and not args.no_cpu_baseline and not training:
            line["cpu_baseline"] = cpu_baseline({k: v.detach().cpu().numpy() for k, v in model.model.state_dict().items()})
            if args.sampler == "ddpm":
                line["speedup_vs_cpu_baseline"] = round(line["value"] / line["cpu_baseline"]["value"], 1)
        if world == 1 and args.workload == "bf16_b65536" and not args.condition and args.sampler == "ddpm" and T == 1000 and not args.no_fp32:
            line["configs_1_fp32_b4096"] = fp32_side_result(dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
