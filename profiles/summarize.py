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
        w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (int(r["TotalDurationNs"]) / 1e6), "%.2f" % (float(r["AverageNs"]) / 1e3),
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
line = os.path.join(src, "bench_line.json")
if os.path.exists(line) and os.path.getsize(line):
    out["bench_line_under_profiler"] = json.loads(open(line).read().strip().splitlines()[-1])
json.dump(out, open(os.path.join(dst, "%s_traffic.json" % tag), "w"), indent=1)
print(open(os.path.join(dst, "%s_kernel_stats.csv" % tag)).read())
print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in out["kernels"].items()}, indent=1))
