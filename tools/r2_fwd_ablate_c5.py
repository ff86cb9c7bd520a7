"""Time the dense forward at C5 (1x3x512x512, dense mask, (49,13)) under its profiling ablations (results wrong)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import engine, synth, _lib
dev = torch.device("cuda:0")
L = _lib.lib_prof()   # profiling build (-DSSG_PROFILE): the product library has no ablation switch
H = W = 512
gt_np = synth.natural_like(300, H, W)[None]
sr_np = synth.degrade(gt_np[0], 7)[None]
mask_np = np.ones((1, 1, H, W), np.float32)
sr, gt, mask = (torch.as_tensor(np.ascontiguousarray(a), device=dev) for a in (sr_np, gt_np, mask_np))
n = H * W
step = engine.LossStep(1, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024)
step(sr, gt, mask); torch.cuda.synchronize()
p = engine._ptr
st = torch.cuda.current_stream().cuda_stream
el = engine.edge_list(mask=mask, capacity=n + 1024, ks=49)
rsc = torch.zeros(2 * (n + 1024), dtype=torch.float64, device=dev)
def fwd():
    _lib.check(L.ssg_map_forward(p(sr), p(gt), 1, 3, H, W, p(el.edges), p(el.order), p(el.rank), p(el.plan), p(el.counts),
                                 n, 49, 13, 1.0, 1e-10, 1, p(step.ssg_sr), p(step.ssg_gt), p(rsc), st))
base = 1 << 26   # skip the direct forward launches
for name, bits in (("full (deferred normalisation)", 0), ("no stores", 1), ("no edge stage", 2), ("no main loop", 64)):
    L.ssg_set_profile_mask(base | (bits << 16))
    fwd(); torch.cuda.synchronize()
    print(f"dense forward C5, {name:30s}: {bench.event_time_ms(fwd, 5):.3f} ms")
L.ssg_set_profile_mask(0)
