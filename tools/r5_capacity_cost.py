"""What a generous `capacity` costs: the C2 step (LossStep and the SSGLoss module) at capacity n + 1024 (bench.py), at a
quarter of the pixels (the module's default) and at every pixel (LossStep's default).   python tools/r5_capacity_cost.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth, SSGLoss
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
def timed(fn, iters):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters): fn()
    en.record(); en.synchronize()
    return st.elapsed_time(en) / iters
for name, (B, H, W, seed, sigma, stride, eps, w) in {"c2": (16, 256, 256, 100, 1.0, 0, 1e-10, 1e3), "c4": (2, 512, 512, 2000, 0.004, 3, 1e-20, 5e2)}.items():
    sr, gt, m = synth.make_batch(B, H, W, seed0=seed)
    a, b, mm = T(sr), T(gt), T(m[:, :1])
    n = int(engine.edge_list(mask=mm, mask_stride=stride).counts[0])
    for cap in (n + 1024, B * H * W // 4, B * H * W):
        step = engine.LossStep(B, 3, H, W, 25, 9, sigma, eps, True, w, w, device=dev, capacity=cap, mask_stride=stride)
        t_step = timed(lambda: step(a, b, mm), 100)
        del step
        crit = SSGLoss(25, 9, sigma, True, w, w, mask_stride=stride, eps=eps, capacity=cap)
        x = a.clone().requires_grad_(True)
        def one():
            x.grad = None
            l1, kl = crit(x, b, mm)
            (l1 + kl).backward()
        t_mod = timed(one, 50)
        del crit
        torch.cuda.empty_cache()
        print("%s n %6d capacity %8d (%.1fx)  LossStep %.4f ms   module %.4f ms" % (name, n, cap, cap / n, t_step, t_mod))
