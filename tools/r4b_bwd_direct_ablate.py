"""Direct backward (ssg_bwd_tiled<25,9>) alone under its profiling ablations (results wrong, timing only):
   python tools/r4b_bwd_direct_ablate.py [c2|b1|b4]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import engine, synth, _lib
dev = torch.device("cuda:0")
L = _lib.lib_prof()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
if cfg == "c2":
    B = 16
    sr_np, gt_np, mask_np = synth.make_batch(B, 256, 256)
else:
    B = 4
    sr_np, gt_np, _ = synth.make_batch(B, 256, 256)
    mask_np = (np.random.default_rng(7).random((B, 1, 256, 256)) < (0.01 if cfg == "b1" else 0.04)).astype(np.float32)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
step = engine.LossStep(B, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024, deterministic=True)
step(sr, gt, mask); torch.cuda.synchronize()
p = engine._ptr
st = torch.cuda.current_stream().cuda_stream
el = engine.edge_list(mask=mask, capacity=n + 1024, ks=25, order=False)
scratch = torch.empty(L.ssg_loss_scratch_bytes(B, 256, 256, n + 1024, 25), dtype=torch.uint8, device=dev)
fix = torch.zeros(B * 3 * 256 * 256 + 8, dtype=torch.int64, device=dev)
def bwd():
    _lib.check(L.ssg_loss_backward(p(sr), B, 3, 256, 256, p(el.edges), None, p(el.rank), p(el.plan), p(el.counts), n,
                                   25, 9, 1.0, 1, p(step.ssg_sr), p(step.ssg_gt), 1e3, 1e3, None, p(step.loss), p(step.grad),
                                   p(scratch), p(fix), None, 0, st))
L.ssg_set_profile_mask(0); bwd(); torch.cuda.synchronize()          # real G rows in the scratch
base = (1 << 27) | (1 << 29)   # skip the dense backward and the row pass: the direct backward (+ reduce / flush) alone
print(cfg, "N =", n, "sparse rows =", int(el.plan[0]))
for name, bits in (("full", 0), ("no gather-merge / atomics", 8), ("gather-merge without its atomics", 32), ("no pass B", 4), ("no pass A", 2), ("no pass A, B", 6), ("nothing but prologue", 14)):
    L.ssg_set_profile_mask(base | (bits << 8))
    bwd(); torch.cuda.synchronize()
    print(f"direct backward, {name:28s}: {bench.event_time_ms(bwd, 30):.4f} ms")
L.ssg_set_profile_mask(0)
