"""Dense-tile kernels against the number of dense tiles: the C4 mask (Laplacian x stride 3, 2 x 512 x 512: 183 dense tiles) on
B = 2, 4, 8, 16 images, every launch on the caller's stream (ssg_set_overlap(0)); run under rocprofv3 --kernel-trace --stats
(tools/r6_dense_fill.sh) the kernel durations show where a launch stops being one under-filled round.
   python tools/r6_dense_fill.py <B> [iters]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth

B = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
sr, gt, m = synth.make_batch(B, 512, 512, seed0=2000)
n = int(engine.edge_list(mask=torch.as_tensor(m, device=dev), mask_stride=3).counts[0]) if False else None
engine.set_overlap(0)
step = engine.LossStep(B, 3, 512, 512, 25, 9, 0.004, 1e-20, True, 5e2, 5e2, device=dev, capacity=B * 8000, mask_stride=3)
a, b, mm = (torch.as_tensor(x, device=dev) for x in (sr, gt, m))
for _ in range(iters):
    step(a, b, mm)
torch.cuda.synchronize()
plan = None
print("B=%d N=%d" % (B, int(step.counts[0])))
