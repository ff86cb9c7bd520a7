import sys, time, torch, numpy as np
sys.path.insert(0, "/root/repo")
import bench
from ssl_amd import synth
cfg = bench.CONFIGS["c2"]; dev = torch.device("cuda:0")
sr_np, gt_np, mask_np = bench.make_inputs(cfg, 0, 1, "weak")
n = int(mask_np.sum())
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
step = bench.make_step(cfg, 16, dev, n + 1024)
def block(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): step(sr, gt, mask)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
print("blocks of 50:", [round(block(50), 4) for _ in range(8)])
print("10 then 50 x3:", [(round(block(10), 4), round(block(50), 4)) for _ in range(3)])
time.sleep(0.01)
print("after 10 ms sleep, 50:", [round(block(50), 4) for _ in range(3)])
# bench.py's own sequence: pre-warm blocks, W warm-up steps, sync, then K-step blocks
def seq(tag, pre):
    pre()
    for _ in range(10): step(sr, gt, mask)
    torch.cuda.synchronize()
    out = []
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(50): step(sr, gt, mask)
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / 50 * 1e3, 4))
        torch.cuda.synchronize()
    print(tag, out)
seq("after 3 synced blocks:", lambda: [block(50) for _ in range(3)])
seq("after nothing:", lambda: None)
import gc
def pre_gc():
    gc.collect(); gc.disable()
seq("gc off:", pre_gc)
gc.enable()
