"""Throughput of the run-time-sized generic kernels vs the tiled instantiations (operator forward + backward,
one 3x256x256 image, C2-like mask)."""
import sys, torch
sys.path.insert(0, ".")
import bench
from ssl_amd import compute_similarity, synth
dev = torch.device("cuda:0")
sr_np, gt_np, m_np = synth.make_batch(1, 256, 256)
img = torch.as_tensor(gt_np[0], device=dev).requires_grad_(True)
mask = torch.as_tensor(m_np[0, 0], device=dev)
for ks, kw in ((25, 9), (25, 5), (25, 13), (11, 5), (13, 5), (49, 13)):
    def f():
        img.grad = None
        q = compute_similarity(img, mask, ks, kw)
        q.backward(torch.ones_like(q))
    f(); torch.cuda.synchronize()
    ms = bench.event_time_ms(f, 5)
    print(f"k_s={ks:2d} k_w={kw:2d}: {ms:8.3f} ms fwd+bwd for N={int(m_np.sum())} edge px")
