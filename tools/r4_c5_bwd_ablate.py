"""Fused C5 step (tile-major rows) under the profiling ablations of ssg_bwd_dense<49,13,TM>: what do the HBM loads and the
gradient atomics cost?  Step times on the profiling build; the kernel's own share = the differences.
   python tools/r4_c5_bwd_ablate.py [iters]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import _lib
dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = bench.CONFIGS["c5"]
sr_np, gt_np, mask_np = bench.make_inputs(cfg, 0, 1, "weak")
n = int(mask_np.sum())
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
with _lib.profile_build() as L:
    step = bench.make_step(cfg, 1, dev, n + 1024, materialise=False)
    def timed():
        for _ in range(3):
            step(sr, gt, mask)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            step(sr, gt, mask)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    for nm, bits in (("full", 0), ("dense backward not launched", 1 << 27), ("tile-major loads from slot 0 (L2 hits)", 16),
                     ("no gradient atomics", 8), ("L2 hits + no atomics", 24)):
        L.ssg_set_profile_mask(bits)
        print(f"{nm:45s} {timed():.4f} ms/step", flush=True)
    L.ssg_set_profile_mask(0)
