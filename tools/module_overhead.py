"""Where the SSGLoss module's extra time over the engine step goes: host time per call (no GPU wait) beside the GPU
time per call, at C2.   PYTHONPATH=. python tools/module_overhead.py"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import SSGLoss, engine
cfg = bench.CONFIGS["c2"]
sr_np, gt_np, mask_np = bench.make_inputs(cfg, 0, 1, "weak")
dev = torch.device("cuda:0")
sr, gt, mask = (torch.as_tensor(np.ascontiguousarray(a), device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
crit = SSGLoss(cfg["ks"], cfg["kw"], cfg["sigma"], True, bench.W_L1, bench.W_KL, capacity=n + 1024)
x = sr.clone().requires_grad_(True)
def one():
    x.grad = None
    a, b = crit(x, gt, mask)
    (a + b).backward()
def fwd_only():
    with torch.no_grad():
        crit(x, gt, mask)
for f, name in ((one, "forward + backward"), (fwd_only, "forward under no_grad")):
    for _ in range(3): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): f()
    host = (time.perf_counter() - t0) / 100 * 1e3
    torch.cuda.synchronize()
    total = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{name:24s}: host {host:.3f} ms per call, with the GPU drained {total:.3f} ms")
step = engine.LossStep(sr.shape[0], 3, sr.shape[2], sr.shape[3], cfg["ks"], cfg["kw"], cfg["sigma"], 1e-10, True, bench.W_L1, bench.W_KL, device=dev, capacity=n + 1024)
for _ in range(3): step(sr, gt, mask)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): step(sr, gt, mask)
host = (time.perf_counter() - t0) / 100 * 1e3
torch.cuda.synchronize()
print(f"{'LossStep':24s}: host {host:.3f} ms per call, with the GPU drained {(time.perf_counter() - t0) / 100 * 1e3:.3f} ms")
