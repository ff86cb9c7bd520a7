import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from oracle import ssg_oracle as orc
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
B, H, W = 2, 512, 512
gt = np.stack([synth.natural_like(2000 + i, H, W) for i in range(B)])
sr = np.stack([synth.degrade(gt[i], 2100 + i) for i in range(B)])
mask = np.stack([orc.mask_stride(synth.laplacian_edge_mask(gt[i]), 3) for i in range(B)]).astype(np.float32)
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
a, b, c = T(sr), T(gt), T(mask[:, None])
n = int(mask.sum())
for cap in (n + 1024, 4 * n, 16 * n, None):
    step = engine.LossStep(B, 3, H, W, 25, 9, 0.004, 1e-20, True, 5e2, 5e2, device=dev, capacity=cap)
    step(a, b, c); torch.cuda.synchronize()
    ms = bench.event_time_ms(lambda: step(a, b, c), 10)
    print(f"capacity {step.capacity:7d}: {ms:.3f} ms/step")
