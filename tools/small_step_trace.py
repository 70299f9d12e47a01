"""One denoising step at batch 32 (bf16), launch by launch, from a rocprofv3 kernel trace of a DDIM-50 call:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/small_step_trace.py run
    python3 tools/small_step_trace.py DIR
The second form prints the median device time of every launch position of a step (the steps of the last call)."""
import csv, glob, sys
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, ".")
    import bench
    bench.load_runtime()
    model = bench.build_model("bf16", torch.device("cuda", 0))
    for _ in range(4):
        model.ddim_sample_loop((32, 512), None, 50, 0.0, seed=1)
    torch.cuda.synchronize()
    sys.exit(0)
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
names = [r[2] for r in rows]
# the last call: 50 steps; a step starts at the launch after the one that carries the update (the only N = 512 / K = 256 launch is not
# recognisable by name, so find the period of the name sequence from the end)
tail = rows[-50 * 40:]
n = len(tail)
for per in range(15, 40):
    seq = [r[2] for r in tail[n - per:]]
    if all([r[2] for r in tail[n - (k + 1) * per:n - k * per]] == seq for k in range(1, 20)):
        break
else:
    raise SystemExit("no period found")
steps = [tail[n - (k + 1) * per:n - k * per] for k in range(1, 41)]
tot = 0.0
gaps = 0.0
for i in range(per):
    d = sorted(s[i][1] - s[i][0] for s in steps)
    g = sorted((s[i + 1][0] - s[i][1]) for s in steps) if i + 1 < per else [0]
    med = d[len(d) // 2] / 1e3
    tot += med
    gaps += g[len(g) // 2] / 1e3
    nm = steps[0][i][2]
    nm = nm[nm.index("adcdiff"):][:60] if "adcdiff" in nm else nm[:60]
    print("%2d %6.2f us  gap after %5.2f us  %s" % (i, med, g[len(g) // 2] / 1e3, nm))
print("launches per step %d, kernel time %.1f us, gaps %.1f us" % (per, tot, gaps))
