"""Median device duration per (shape, variant) of tools/splitk_probe.py from a rocprofv3 kernel trace of it: the probe launches
320 GEMMs per (shape, variant) in a fixed order, so the n-th group of 320 GEMM dispatches belongs to the n-th combination.
python tools/trace_by_order.py <dir with *_kernel_trace.csv> [launches per group]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
per = int(sys.argv[2]) if len(sys.argv) > 2 else 320
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        if "gemm" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
rows.sort()
out = []
for g in range(len(rows) // per):
    grp = rows[g * per + 20:(g + 1) * per]
    d = sorted(x[1] for x in grp)
    name = grp[0][2]
    short = name[name.index("gemm"):][:44]
    out.append((d[len(d) // 2] / 1e3, short))
tot = [0.0, 0.0]
for i in range(0, len(out) - 1, 2):
    print("%2d  default %6.2f us (%s)   variant %6.2f us (%s)" % (i // 2, out[i][0], out[i][1], out[i + 1][0], out[i + 1][1]))
    tot[0] += out[i][0]; tot[1] += out[i + 1][0]
print("sum: default %.1f us, variant %.1f us" % tuple(tot))
