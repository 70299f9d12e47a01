#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (written by profiles/collect.sh) into profiles/<tag>_*.{csv,json,md}."""
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s" % tag
dst = "profiles"


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern), recursive=True)
    return hits[0] if hits else None


def short(name):
    if "gemm_nt_kernel" in name:
        return "gemm_nt_kernel<bf16>" if "DF16b" in name or "bf16" in name else "gemm_nt_kernel<float>"
    if "gemm256w" in name:
        return "gemm256w_kernel<bf16>"
    if "gemm256" in name:
        return "gemm256_kernel<bf16>"
    for key in ("gn_silu_vec_kernel", "gn_silu_generic_kernel", "ddpm_step_kernel", "ddim_step_kernel", "philox_normal_kernel",
                "cast_pack_kernel", "temb_silu_kernel", "sinusoid_kernel", "add_vec_kernel", "q_sample_kernel"):
        if key in name:
            return key
    return name[:60]


stats = one("trace/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
    for r in rows:
        label = short(r["Name"])
        if "gemm256w" in r["Name"]:   # template arguments <epilogue batch, side inputs known at compile time>
            label += " general epilogue" if ", 0>" in r["Name"] or "Li0EE" in r["Name"] else \
                     " residual look-ahead" if ", 2>" in r["Name"] or "Li2EE" in r["Name"] else " no side inputs"
        w.writerow([label, r["Calls"], "%.3f" % (int(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
                    r["Percentage"]])

traffic = {}
for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    f = one("%s/**/*counter_collection.csv" % sub)
    if not f:
        continue
    agg = {}
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter:
            continue
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        traffic.setdefault(k, {})[counter] = {"launches": n, "mean_value_kb": v / n}
out = {"tag": tag, "note": "FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; on gfx950 FETCH_SIZE tallies 128-byte read requests at "
                            "64 bytes, so bytes read = 2 * FETCH_SIZE * 1024 for wide streaming reads", "kernels": {}}
for k, d in traffic.items():
    fetch = d.get("FETCH_SIZE", {}).get("mean_value_kb")
    write = d.get("WRITE_SIZE", {}).get("mean_value_kb")
    entry = dict(d)
    if fetch is not None and write is not None:
        entry["hbm_bytes_per_launch"] = (2.0 * fetch + write) * 1024.0
    out["kernels"][k] = entry
# matrix-core counters: per kernel, sums over the dispatches of the pass, joined with the dispatch durations of the same pass
mf = one("pmc_mfma/**/*counter_collection.csv")
if mf:
    agg = {}
    seen = set()
    for r in csv.DictReader(open(mf)):
        k = short(r["Kernel_Name"])
        a = agg.setdefault(k, {"dispatches": 0, "ns": 0.0})
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        did = r.get("Dispatch_Id")
        if did not in seen:
            seen.add(did)
            a["dispatches"] += 1
            if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                a["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    mfma = {}
    for k, a in agg.items():
        busy = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if busy <= 0:
            continue
        e = {"dispatches": a["dispatches"], "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_WAVE_CYCLES": a.get("SQ_WAVE_CYCLES"),
             "SQ_WAIT_ANY": a.get("SQ_WAIT_ANY"), "SQ_WAIT_INST_ANY": a.get("SQ_WAIT_INST_ANY"),
             "SQ_VALU_MFMA_COEXEC_CYCLES": a.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
             "mfma_mops_bf16": a.get("SQ_INSTS_VALU_MFMA_MOPS_BF16"), "mfma_mops_f32": a.get("SQ_INSTS_VALU_MFMA_MOPS_F32")}
        if a["ns"] > 0:
            # busy cycles summed over the chip's 1024 SIMDs against the kernel's wall time: the clock-equivalent each SIMD's matrix pipe
            # was busy at, and that as a fraction of the 2.4 GHz the dense peak is quoted at
            ghz = busy / 1024.0 / a["ns"]
            e["mfma_busy_ghz_per_simd"] = ghz
            e["mfma_busy_frac_of_2p4ghz"] = ghz / 2.4
            e["avg_dispatch_us_under_counters"] = a["ns"] / a["dispatches"] / 1e3
        mfma[k] = e
    out["mfma_counters"] = {"note": "rocprofv3 --pmc pass of its own (profiled clocks run a few per cent lower than unprofiled ones); "
                                    "SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles per wave",
                            "kernels": mfma}
line = os.path.join(src, "bench_line.json")
if os.path.exists(line) and os.path.getsize(line):
    out["bench_line_under_profiler"] = json.loads(open(line).read().strip().splitlines()[-1])
json.dump(out, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
print(open(os.path.join(dst, "%s_kernel_stats.csv" % tag)).read())
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out["kernels"].items()}, indent=1))
