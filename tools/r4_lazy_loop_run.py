"""A few deferred caller loops at C2 (for a rocprofv3 kernel trace: tools/r4_lazy_timeline.sh)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import synth
from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
from ssl_amd.reference_loop import gan_selfsim_block
dev = torch.device("cuda:0")
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256, seed0=100)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
cri1, cri2 = L1Loss(1e3), KLDistanceLoss(1e3)
x = sr.clone().requires_grad_(True)
setting = dict(ssl_mode=os.environ.get("MODE", "cuda"), kernel_size_search=25, generalization=True, kernel_size_window=9, sigma=1.0)
set_lazy(True)
def loop():
    x.grad = None
    out = x * 1.0
    l1, kl = gan_selfsim_block(similarity_map, cri1, cri2, out, gt, mask, setting)
    (l1 + kl).backward()
for _ in range(5):
    loop()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    loop()
torch.cuda.synchronize(); print("ms per loop %.3f" % ((time.perf_counter() - t0) * 100))
