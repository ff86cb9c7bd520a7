"""The reference operator on one 3x512x512 image (18 k positions) with the plan built inside the call and without:
HIP-event times of forward and backward alone (no pad / nonzero), for rocprofv3 kernel traces."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import _lib, synth
dev = torch.device("cuda:0")
L = _lib.lib()
H = W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gt = synth.natural_like(2000, H, W)
mask = synth.laplacian_edge_mask(gt)
hp = 12
pad = np.pad(gt, ((0, 0), (hp, hp), (hp, hp)), mode="reflect")
ys, xs = np.nonzero(mask)
pos = torch.as_tensor(np.stack([ys + hp, xs + hp], 1).astype(np.int32), device=dev)
mc = pos.shape[0]
image = torch.as_tensor(pad, device=dev)
out = torch.zeros(mc, 25, 25, device=dev)
cot = torch.rand(mc, 25, 25, device=dev)
g = torch.zeros_like(image)
st = torch.cuda.current_stream().cuda_stream
Hp, Wp = H + 2 * hp, W + 2 * hp
fwd = lambda: _lib.check(L.ssg_compute_similarity(image.data_ptr(), pos.data_ptr(), out.data_ptr(), mc, 25, 9, Hp, Wp, 3, st))
bwd = lambda: _lib.check(L.ssg_compute_similarity_backward(image.data_ptr(), cot.data_ptr(), pos.data_ptr(), g.data_ptr(), mc, 25, 9, Hp, Wp, 3, st))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
def ev(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); a.record()
    for _ in range(it):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / it
for name, thr in (("direct kernels in pos order", 0), ("plan built inside the call", 1)):
    L.ssg_set_operator_plan_threshold(thr)
    print(f"{H}x{W}, {mc} positions, {name}: forward {ev(fwd):.4f} ms, backward {ev(bwd):.4f} ms")
