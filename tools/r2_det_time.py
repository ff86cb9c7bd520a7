import sys, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
for det in (False, True):
    step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024, deterministic=det)
    step(sr, gt, mask); torch.cuda.synchronize()
    print("deterministic", det, "step ms", [round(bench.event_time_ms(lambda: step(sr, gt, mask), 20), 4) for _ in range(3)])
