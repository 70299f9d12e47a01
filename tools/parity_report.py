#!/usr/bin/env python3
"""Observed parity of the GPU path against the reference's golden vectors, as numbers (run on the GPU box):

    python tools/parity_report.py gpurun_out/parity.json

For {tiny 128/128/128, Config A 512/256/512, Config B 512/128/512} x {fp32, bf16} x {DDPM T = 1000, DDIM 50 steps}
the final latents of the drop-in class (in-kernel Philox noise) against what the reference computed with the same noise
definition injected: per-element |d_ij| / max(1, |ref_ij|), maximum and 99.9th percentile, plus the tensor-relative
max |d| / max(1, max|ref|) and the magnitude of the latents.  The full tensors are compared for the tiny configuration, the
64 probe columns per row that the golden files hold for Configs A and B.  Also one denoiser forward per configuration."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import adcdiff  # noqa: E402
from conftest import elem_err, load_golden, rel_err, row_scaled_err  # noqa: E402
from oracle import philox  # noqa: E402
from oracle.weights import make_weights  # noqa: E402

SEED = 2025


def build(dims, dtype):
    m = adcdiff.DiffusionModel(dict(input_dim=dims[0], hidden_dim=dims[1], output_dim=dims[2], num_timesteps=1000, compute_dtype=dtype,
                                    memory_monitor_enabled=False))
    m.model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in make_weights(1234, *dims).items()})
    return m


def entry(out, ref):
    mx, p = elem_err(out, ref)
    return {"elem_max": mx, "elem_p999": p, "row_scaled_max": row_scaled_err(out, ref), "abs_max": float(np.abs(np.asarray(out, dtype=np.float64) - ref).max()),
            "tensor_rel": rel_err(out, ref), "ref_abs_max": float(np.abs(ref).max()),
            "ref_abs_median": float(np.median(np.abs(ref))), "n": int(np.asarray(ref).size)}


def main():
    report = {"measure": "elem_*: |d_ij| / max(1, |ref_ij|) over the compared elements; row_scaled_max: |d_ij| / max(1, rms of ref row i); abs_max: |d_ij|; tensor_rel: max|d| / max(1, max|ref|)", "device": torch.cuda.get_device_name(0), "cases": {}}
    cu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).to("cuda").to(dt)
    for name, dims, B in (("tiny", (128, 128, 128), 8), ("config_a", (512, 256, 512), 32), ("config_b", (512, 128, 512), 32)):
        g = load_golden(name)
        for dtype in ("fp32", "bf16"):
            m = build(dims, dtype)
            x = philox.normal(SEED, 50, 0, 0, B, dims[0])
            y = m.model(cu(x), cu(g["fwd_t"], torch.int64)).cpu().numpy()
            ddpm = m.sample(B, use_ddim=False, seed=SEED).cpu().numpy()
            ddim = m.sample(B, use_ddim=True, ddim_steps=50, seed=SEED).cpu().numpy()
            if name == "tiny":
                fwd_ref, ddpm_ref, ddim_ref = g["fwd_uncond"], g["loop_T1000"], g["ddim50_eta0"]
            else:
                cols = g["fwd.cols"]
                y, ddpm, ddim = y[:, cols], ddpm[:, cols], ddim[:, cols]
                fwd_ref, ddpm_ref, ddim_ref = g["fwd.values"], g["loop_T1000.values"], g["ddim50.values"]
            report["cases"]["%s/%s/forward" % (name, dtype)] = entry(y, fwd_ref)
            report["cases"]["%s/%s/ddpm1000" % (name, dtype)] = entry(ddpm, ddpm_ref)
            report["cases"]["%s/%s/ddim50" % (name, dtype)] = entry(ddim, ddim_ref)
            m.cleanup_resources()
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity.json"
    json.dump(report, open(out, "w"), indent=1)
    for k, v in report["cases"].items():
        print("%-28s elem max %.3e  p99.9 %.3e  row-scaled %.3e  abs %.3e  tensor-rel %.3e  |ref| max %.1f median %.1f" % (k, v["elem_max"], v["elem_p999"], v["row_scaled_max"], v["abs_max"], v["tensor_rel"],
                                                                                        v["ref_abs_max"], v["ref_abs_median"]))


if __name__ == "__main__":
    main()
