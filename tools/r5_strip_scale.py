"""Fused k_s = 49 step on dense masks of 1 x 3 x H x 512 for several H: how the strip forward's duration scales with
the number of strips (workgroups).  Run under rocprofv3 --kernel-trace --output-format csv; tools/r5_strip_scale.sh
prints mean duration per (kernel, grid size)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
for H in (int(a) for a in sys.argv[1:]):
    W = 512
    gt = synth.natural_like(300, 512, 512)[None][:, :, :H]
    sr = synth.degrade(gt[0], 7)[None]
    m = np.ones((1, 1, H, W), np.float32)
    step = engine.LossStep(1, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=H * W + 64, materialise=False)
    a, b, mm = (torch.as_tensor(np.ascontiguousarray(x), device=dev) for x in (sr, gt, m))
    for _ in range(6):
        step(a, b, mm)
    torch.cuda.synchronize()
    print("H", H, "N", int(step.counts[0]), "loss", step.loss.cpu().numpy())
    del step
    torch.cuda.empty_cache()
