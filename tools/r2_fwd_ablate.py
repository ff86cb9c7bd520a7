"""Time the dense forward at C2 under its profiling ablations (results wrong, timing only)."""
import sys, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import engine, synth, _lib
dev = torch.device("cuda:0")
L = _lib.lib_prof()   # profiling build (-DSSG_PROFILE): the product library has no ablation switch
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024)
step(sr, gt, mask); torch.cuda.synchronize()
p = engine._ptr
st = torch.cuda.current_stream().cuda_stream
el = engine.edge_list(mask=mask, capacity=n + 1024, ks=25)
rsc = torch.empty(2 * (n + 1024), dtype=torch.float64, device=dev) if "--deferred" in sys.argv else None
def fwd():
    _lib.check(L.ssg_map_forward(p(sr), p(gt), 16, 3, 256, 256, p(el.edges), p(el.order), p(el.rank), p(el.plan), p(el.counts),
                                 n, 25, 9, 1.0, 1e-10, 1, p(step.ssg_sr), p(step.ssg_gt), p(rsc), st))
base = 1 << 26   # skip the direct forward launches
for name, bits in (("full", 0), ("no stores", 1), ("no edge stage", 2), ("no rescale", 8), ("no stores+no rescale", 9),
                   ("no edge, no rescale", 10), ("no main loop", 64)):
    L.ssg_set_profile_mask(base | (bits << 16))
    fwd(); torch.cuda.synchronize()
    print(f"dense forward, {name:24s}: {bench.event_time_ms(fwd, 20):.3f} ms")
L.ssg_set_profile_mask(0)
