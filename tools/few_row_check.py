import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import adcdiff, bench
bench.load_runtime()
m = bench.build_model("bf16", torch.device("cuda", 0))
for B in (1, 7, 32, 33, 48, 64, 65):
    for cond in (None, torch.randn(1, 512, device="cuda").expand(B, -1)):
        a = m.p_sample_loop((B, 512), cond, seed=3, t_start=999, t_stop=990)
        a2 = m.p_sample_loop((B, 512), cond, seed=3, t_start=999, t_stop=990)
        adcdiff._lib.debug_set("gemm_splitk", 0)
        b = m.p_sample_loop((B, 512), cond, seed=3, t_start=999, t_stop=990)
        adcdiff._lib.debug_set("gemm_splitk", 1)
        d = m.ddim_sample_loop((B, 512), cond, 10, 0.0, seed=3)
        print(B, cond is not None, "repro", torch.equal(a, a2), "finite", bool(torch.isfinite(a).all() and torch.isfinite(d).all()),
              "vs plain %.2e" % (float((a - b).abs().max()) / float(b.abs().max())), flush=True)
