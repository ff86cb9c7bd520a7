"""Time the dense backward at C2 under profiling ablations (results wrong, timing only)."""
import sys, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import engine, synth, _lib
dev = torch.device("cuda:0")
L = _lib.lib_prof()   # profiling build (-DSSG_PROFILE): the product library has no ablation switch
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024, deterministic=False)
step(sr, gt, mask); torch.cuda.synchronize()
p = engine._ptr
st = torch.cuda.current_stream().cuda_stream
el = engine.edge_list(mask=mask, capacity=n + 1024, ks=25)
scratch = torch.empty(L.ssg_loss_scratch_bytes(16, 256, 256, n + 1024, 25), dtype=torch.uint8, device=dev)
def bwd():
    _lib.check(L.ssg_loss_backward(p(sr), 16, 3, 256, 256, p(el.edges), p(el.order), p(el.rank), p(el.plan), p(el.counts), n,
                                   25, 9, 1.0, 1, p(step.ssg_sr), p(step.ssg_gt), 1e3, 1e3, None, p(step.loss), p(step.grad),
                                   p(scratch), None, None, 0, st))
base = (1 << 28) | (1 << 29)   # skip direct backward and G rows: the dense backward (+finalize) alone
for name, bits in (("full", 0), ("no atomics", 8), ("all lanes stream G row 0", 16), ("both", 24)):
    L.ssg_set_profile_mask(base | (bits << 8))
    bwd(); torch.cuda.synchronize()
    print(f"dense backward, {name:28s}: {bench.event_time_ms(bwd, 20):.3f} ms")
L.ssg_set_profile_mask(0)
