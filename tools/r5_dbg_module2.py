import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from ssl_amd import SSGLoss, synth, engine, _lib
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
B, H, W, ks, kw, sigma, stride, eps, w = 2, 512, 512, 25, 9, 0.004, 3, 1e-20, 5e2
sr, gt, m = synth.make_batch(B, H, W, seed0=2000)
n = int((m[:, :1] * ((np.add.outer(np.arange(H), -np.arange(W)) % 3) == 0)).sum())
x = T(sr).clone().requires_grad_(True); g = T(gt); mk = T(m[:, :1])
def measure(tag):
    crit = SSGLoss(ks, kw, sigma, True, w, w, mask_stride=stride, eps=eps, capacity=n + 1024)
    def one():
        x.grad = None
        a, b = crit(x, g, mk)
        (a + b).backward()
    for _ in range(3): one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): one()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(tag, "host %.3f total %.3f ms/call; assignment %d" % ((t1 - t0) * 20, (t2 - t0) * 20, _lib.lib().ssg_last_overlap_assignment()))
measure("before table")
step = engine.LossStep(B, 3, H, W, ks, kw, sigma, eps, True, w, w, device=dev, capacity=n + 1024, mask_stride=stride)
step(T(sr), g, mk); torch.cuda.synchronize()
bench.C = 3
out = bench.stage_times(step, T(sr), g, mk, n, 20)
print({k[:20]: round(v, 3) for k, v in out.items()})
measure("after table ")
measure("again       ")
