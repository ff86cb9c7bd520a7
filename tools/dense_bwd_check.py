"""GPU check of the dense-tile backward: split backward (plan with threshold T) vs direct kernels (T = 0) on the
same inputs, and against the fp64 oracle on a small case; stage timings at C2."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from ssl_amd import engine, synth
from oracle import ssg_oracle as orc

dev = torch.device("cuda:0")


def run(sr, gt, mask, ks, kw, sigma, thr):
    engine.set_dense_threshold(thr)
    B, C, H, W = sr.shape
    step = engine.LossStep(B, C, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev)
    loss, grad = step(sr, gt, mask)
    torch.cuda.synchronize()
    return loss.cpu().numpy().copy(), grad.cpu().numpy().copy()


# small odd-sized case vs oracle (fp64)
for (H, W, dens, sigma) in [(45, 70, 0.3, 1.0), (37, 41, 1.0, 0.05), (64, 96, 0.15, 0.004)]:
    rng = np.random.default_rng(5)
    g = synth.natural_like(11, H, W)[None]
    s = synth.degrade(g[0], 3)[None]
    m = (rng.random((1, 1, H, W)) < dens).astype(np.float32)
    ref = orc.ssg_loss(s.astype(np.float64), g.astype(np.float64), m[:, 0], 25, 9, sigma, 1e3, 1e3)
    T = lambda a: torch.as_tensor(a, device=dev)
    out = {}
    for thr in (0, 1, 28):
        l, gr = run(T(s), T(g), T(m), 25, 9, sigma, thr)
        out[thr] = gr
        e = np.abs(gr - ref["grad"]).max() / np.abs(ref["grad"]).max()
        print(f"{H}x{W} dens {dens} sigma {sigma} thr {thr}: loss {l} (ref {ref['l1']:.6f} {ref['kl']:.6g}) grad err vs fp64 oracle {e:.2e}")
    print("   dense-all vs direct:", np.abs(out[1] - out[0]).max() / np.abs(out[0]).max())

# C2 batch
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
res = {}
for thr in (0, 28, 16, 8, 1):
    l, gr = run(sr, gt, mask, 25, 9, 1.0, thr)
    res[thr] = gr
    if thr:
        print(f"C2 thr {thr}: loss {l} grad diff vs direct {np.abs(gr - res[0]).max() / np.abs(res[0]).max():.2e}")
import bench
n = int(mask_np.sum())
for thr in (0, 28, 16, 8):
    engine.set_dense_threshold(thr)
    step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024)
    step(sr, gt, mask); torch.cuda.synchronize()
    t = bench.stage_times(step, sr, gt, mask, n, 20)
    tot = bench.event_time_ms(lambda: step(sr, gt, mask), 20)
    print(f"thr {thr}: step {tot:.3f} ms  " + "  ".join(f"{k.split('(')[0][:28]}={v:.3f}" for k, v in t.items()))
