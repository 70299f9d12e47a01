"""Where the host spends a training step (cProfile over 50 steps at batch 4096, emb 512 / hidden 256, bf16).  python tools/train_host_profile.py"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import bench

bench.load_runtime()
dev = torch.device("cuda", 0)
model = bench.build_model("bf16", dev, max_batch_size=4096, learning_rate=1e-4)
x = bench.ops.philox_normal(4096, 512, 7, 52, 0, 0, dev)
for k in range(5):
    model.train_step(x, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(50):
    model.train_step(x, k)
torch.cuda.synchronize()
print("ms per step %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
# the same without the host waiting for the loss value at the end of every step (what the returned dict's float costs)
real_item = torch.Tensor.item
torch.Tensor.item = lambda self: real_item(self) if not self.is_cuda else 0.0
t0 = time.perf_counter()
for k in range(50):
    model.train_step(x, k)
torch.cuda.synchronize()
print("ms per step without the loss synchronisation %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
t0 = time.perf_counter()
for k in range(50):
    model.train_step(x, k)
print("host ms per step (no synchronisation at all) %.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
torch.cuda.synchronize()
torch.Tensor.item = real_item
pr = cProfile.Profile()
pr.enable()
for k in range(50):
    model.train_step(x, k)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
st.print_callers("parameters")
st.print_callers("_param_version")

# host time of the step's parts (no synchronisation: the queue absorbs the launches), averaged over 50 steps
torch.Tensor.item = lambda self: real_item(self) if not self.is_cuda else 0.0
import collections
acc = collections.defaultdict(float)
m = model
for k in range(50):
    t0 = time.perf_counter()
    m.model.train()
    t = torch.randint(0, m.num_timesteps, (4096,), device=m.device)
    t1 = time.perf_counter()
    m.model.alias_gradients = True
    loss = m.p_losses(x, t)
    t2 = time.perf_counter()
    loss.backward()
    t3 = time.perf_counter()
    flat = m.model.grad_arena_if_aliased()
    t4 = time.perf_counter()
    m._fused_opt.step()
    t5 = time.perf_counter()
    m.optimizer.zero_grad()
    t6 = time.perf_counter()
    m._get_memory_usage()
    t7 = time.perf_counter()
    for name, a, b in (("train()+randint", t0, t1), ("p_losses fwd", t1, t2), ("backward", t2, t3), ("arena check", t3, t4), ("optimizer", t4, t5),
                       ("zero_grad", t5, t6), ("memory usage", t6, t7)):
        acc[name] += (b - a) / 50 * 1e3
torch.cuda.synchronize()
torch.Tensor.item = real_item
print("host ms per part:", {k: round(v, 3) for k, v in acc.items()}, "sum", round(sum(acc.values()), 3))
