"""C4-shaped check (LDM-SR: 512x512 crops, eps 1e-20, mask_stride 3, sigma 0.004, w = 5e2): loss step vs the oracle
and its time, 2 images per GPU."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from oracle import ssg_oracle as orc
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
B, H, W = 2, 512, 512
gt = np.stack([synth.natural_like(2000 + i, H, W) for i in range(B)])
sr = np.stack([synth.degrade(gt[i], 2100 + i) for i in range(B)])
mask = np.stack([orc.mask_stride(synth.laplacian_edge_mask(gt[i]), 3) for i in range(B)]).astype(np.float32)
step = engine.LossStep(B, 3, H, W, 25, 9, 0.004, 1e-20, True, 5e2, 5e2, device=dev)
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
a, b, c = T(sr), T(gt), T(mask[:, None])
loss, grad = step(a, b, c)
n = int(step.counts[0])
ref = orc.ssg_loss(sr.astype(np.float64), gt.astype(np.float64), mask, 25, 9, 0.004, 5e2, 5e2, eps=1e-20)
l = loss.cpu().numpy()
ge = float(np.abs(grad.cpu().numpy() - ref["grad"]).max() / np.abs(ref["grad"]).max())
ms = bench.event_time_ms(lambda: step(a, b, c), 10)
print(f"C4-shaped: N={n} (oracle {ref['n_edges']})  l1 {l[0]:.6g} vs {ref['l1']:.6g}  kl {l[1]:.6g} vs {ref['kl']:.6g}  grad rel err {ge:.2e}  {ms:.3f} ms/step")
# the same with the mask at full density (no stride): dense tiles appear
mask2 = np.stack([synth.laplacian_edge_mask(gt[i]) for i in range(B)]).astype(np.float32)
c2 = T(mask2[:, None])
loss, grad = step(a, b, c2); n2 = int(step.counts[0])
ms2 = bench.event_time_ms(lambda: step(a, b, c2), 10)
print(f"  without mask_stride: N={n2}  {ms2:.3f} ms/step")
