#!/bin/bash
# per-kernel times of the stress config C5 (1x3x512x512, dense mask, k_s=49, k_w=13)
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/c5_time.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import engine, synth
bench.H = bench.W = 512; bench.KS, bench.KW, bench.C = 49, 13, 3
dev = torch.device("cuda:0")
g = synth.natural_like(500, 512, 512)[None]
sr_np, gt_np = synth.degrade(g[0], 501)[None], g
mask_np = np.ones((1, 1, 512, 512), np.float32)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = 512 * 512
step = engine.LossStep(1, 3, 512, 512, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n)
step(sr, gt, mask); torch.cuda.synchronize()
t = bench.stage_times(step, sr, gt, mask, n, 3)
print("C5  " + "  ".join(f"{k.split('(')[0][:30]}={v:.2f}" for k, v in t.items()))
PY
python /tmp/c5_time.py 2>&1 | tail -2
