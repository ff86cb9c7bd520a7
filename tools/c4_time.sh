#!/bin/bash
# per-kernel times of the C4-shaped step (2 x 3x512x512, mask_stride 3, sigma 0.004, eps 1e-20)
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/c4_time.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from oracle import ssg_oracle as orc
from ssl_amd import engine, synth
bench.H = bench.W = 512; bench.SIGMA = 0.004; bench.EPS = 1e-20
dev = torch.device("cuda:0")
B = 2
import os; CAPFULL = os.environ.get("CAPFULL") == "1"
gt_np = np.stack([synth.natural_like(2000 + i, 512, 512) for i in range(B)])
sr_np = np.stack([synth.degrade(gt_np[i], 2100 + i) for i in range(B)])
for stride in (3, 0):
    m = np.stack([synth.laplacian_edge_mask(gt_np[i]) for i in range(B)]).astype(np.float32)
    if stride: m = np.stack([orc.mask_stride(m[i], stride) for i in range(B)]).astype(np.float32)
    sr, gt, mask = (torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev) for a in (sr_np, gt_np, m[:, None]))
    n = int(m.sum())
    step = engine.LossStep(B, 3, 512, 512, 25, 9, 0.004, 1e-20, True, 5e2, 5e2, device=dev, capacity=(None if CAPFULL else n + 1024))
    step(sr, gt, mask); torch.cuda.synchronize()
    t = bench.stage_times(step, sr, gt, mask, n, 5)
    print(f"stride {stride} N={n}: " + "  ".join(f"{k.split('(')[0][:22]}={v:.3f}" for k, v in t.items()))
PY
python /tmp/c4_time.py 2>&1 | tail -2
