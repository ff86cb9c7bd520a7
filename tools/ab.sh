#!/bin/bash
# A/B timing of the C2 loss step stages under SSG_DEBUG_SKIP masks: tools/ab.sh MASK [MASK...]
# (each mask is run 3 times; results are wrong under ablation masks, this only measures time)
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/ab_time.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024)
step(sr, gt, mask); torch.cuda.synchronize()
for rep in range(3):
    t = bench.stage_times(step, sr, gt, mask, n, 20)
    print("  ".join(f"{k.split('(')[0][:24]}={v:.3f}" for k, v in t.items()))
PY
for m in "$@"; do echo "SSG_DEBUG_SKIP=$m"; SSG_DEBUG_SKIP=$m python /tmp/ab_time.py 2>&1 | tail -3; done
