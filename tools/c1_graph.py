"""C1 (1x3x64x64, 209 edge px, k_s=11, k_w=5): the step is launch-bound (17 launches for ~25 us of kernel work);
per-kernel launches vs HIP-graph replay of the recorded step."""
import sys, time, torch
sys.path.insert(0, ".")
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr_np, gt_np, m_np = synth.uniform_case()
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, m_np))
for graph in (False, True, False, True):
    step = engine.LossStep(1, 3, 64, 64, 11, 5, 1.0, 1e-10, True, 1e3, 1e3, device=dev, graph=graph)
    for _ in range(20):
        step(sr, gt, mask)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500):
        step(sr, gt, mask)
    torch.cuda.synchronize()
    print(f"C1 graph={graph}: {(time.perf_counter() - t0) / 500 * 1e6:.1f} us/step  loss {step.loss.cpu().numpy()}")
