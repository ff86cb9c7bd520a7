#!/bin/bash
# phase ablation of the dense forward kernel (all non-empty tiles routed to it)
cd "${GRAFT_REPO_ROOT:-.}"
cat > /tmp/fwd_time.py <<'PY'
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
t = bench.stage_times(step, sr, gt, mask, n, 10)
print("  ".join(f"{k.split('(')[0][:28]}={v:.3f}" for k, v in t.items()))
PY
for thr in 1 28; do for m in 0 1 2 4 8 16 32 64 127; do
  echo -n "thr=$thr dense_dbg=$m: "; SSG_DENSE_THR=$thr SSG_DEBUG_SKIP=$((m<<16)) python /tmp/fwd_time.py 2>&1 | tail -1
done; done
