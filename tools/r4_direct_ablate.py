"""Direct forward / backward kernels under their profiling ablations at a SPARSE configuration (4x3x256x256 Bernoulli
1 % or 4 %, or C4): where does a launch's time go when the call has too few jobs to fill the chip?
   python tools/r4_direct_ablate.py <b1|b4|c4>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import engine, synth, _lib
dev = torch.device("cuda:0")
L = _lib.lib_prof()
name = sys.argv[1] if len(sys.argv) > 1 else "b1"
stride = 0
if name in ("b1", "b4"):
    sr_np, gt_np, _ = synth.make_batch(4, 256, 256)
    rng = np.random.default_rng(0)
    for d in (0.01, 0.04):
        mm = (rng.random((4, 1, 256, 256)) < d).astype(np.float32)
        if d == {"b1": 0.01, "b4": 0.04}[name]:
            mask_np = mm
else:
    sr_np, gt_np, mask_np = synth.make_batch(2, 512, 512, seed0=2000)
    stride = 3
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
B, C, H, W = sr_np.shape
el = engine.edge_list(mask=mask, mask_stride=stride, ks=25)
n = int(el.counts[0])
cap = B * H * W
el = engine.edge_list(mask=mask, mask_stride=stride, ks=25, capacity=n + 64)
print(f"{name}: N = {n}, plan header (sparse rows, heavy tiles, tile rows, light tiles) = {el.plan[:4].tolist()}")
p = engine._ptr
st = torch.cuda.current_stream().cuda_stream
ssg_sr = torch.empty((n + 64, 625), device=dev)
ssg_gt = torch.empty((n + 64, 625), device=dev)
def fwd():
    _lib.check(L.ssg_map_forward(p(sr), p(gt), B, 3, H, W, p(el.edges), p(el.order), p(el.rank), p(el.plan), p(el.counts),
                                 n, 25, 9, 1.0, 1e-10, 1, p(ssg_sr), p(ssg_gt), None, st))
base = 1 << 25   # skip the dense forward launch
for nm, bits in (("full", 0), ("no fill", 1), ("no main loop", 2), ("no epilogue", 4), ("no fill, no main loop", 3),
                 ("only launch + job setup", 7)):
    L.ssg_set_profile_mask(base | bits)
    fwd(); torch.cuda.synchronize()
    print(f"direct forward (launches of the plan's sparse rows), {nm:26s}: {bench.event_time_ms(fwd, 30):.4f} ms")
L.ssg_set_profile_mask(1 << 26)
fwd(); torch.cuda.synchronize()
print(f"dense forward alone: {bench.event_time_ms(fwd, 30):.4f} ms")
L.ssg_set_profile_mask(0)
fwd(); torch.cuda.synchronize()
print(f"forward, both (side stream): {bench.event_time_ms(fwd, 30):.4f} ms")
