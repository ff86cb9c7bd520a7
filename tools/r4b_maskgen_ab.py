"""C2 step with the fp32 mask against the same step with mask=None (Laplacian mask of GT generated inside the edge-list
builder), alternating in one process: python tools/r4b_maskgen_ab.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
cfg = bench.CONFIGS["c2"]
sr_np, gt_np, mask_np = bench.make_inputs(cfg, 0, 1, "weak")
n = int(bench.effective_mask(cfg, mask_np).sum())
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
step = bench.make_step(cfg, sr.shape[0], dev, n + 1024, materialise=True)
def run(m, steps=100):
    for _ in range(10):
        step(sr, gt, m)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        step(sr, gt, m)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
run(mask); run(None)
for _ in range(4):
    print("fp32 mask %.4f ms   mask=None (Laplacian on the device) %.4f ms" % (run(mask), run(None)), flush=True)
assert int(step.counts[0]) == n
