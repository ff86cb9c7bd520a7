"""Cost of sizing the launches for the worst case: C2 step with capacity = N + 1024 (bench) vs capacity = B*H*W
(the default of SSGLoss / LossStep: every pixel may be an edge pixel, no host round trip to size anything)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr_np, gt_np, m_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, m_np))
n = int(m_np.sum())
for cap in (n + 1024, 4 * n, None):
    step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=cap)
    for _ in range(3):
        step(sr, gt, mask)
    ms = bench.event_time_ms(lambda: step(sr, gt, mask), 10)
    print(f"capacity {step.capacity:8d} rows: {ms:.3f} ms/step")
