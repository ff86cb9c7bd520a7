"""One sparse / small configuration of the loss step, for rocprofv3 kernel traces and launch-overhead measurements:
   python tools/sparse_step.py <c1|b1|b4|c4|i1|c2> [iters] [--graph]
c1 = BASELINE configs[0] (1x3x64x64, 209 px, (11,5)); b1 / b4 = 4x3x256x256 Bernoulli 1 % / 4 % masks, (25,9);
c4 = BASELINE configs[3] per GPU (2x3x512x512, Laplacian x stride 3, eps 1e-20).  Prints ms/step (HIP events)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth

name = sys.argv[1]
iters = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 50
graph = "--graph" in sys.argv
dev = torch.device("cuda:0")
kw = {}
if name == "c1":
    sr, gt, m = synth.uniform_case()
    ks, kwin, sigma = 11, 5, 1.0
elif name in ("b1", "b4"):
    sr, gt, _ = synth.make_batch(4, 256, 256)
    rng = np.random.default_rng(0)
    dens = {"b1": 0.01, "b4": 0.04}[name]
    for d in (0.01, 0.04):          # (the same generator stream as tools/sweep.py)
        mm = (rng.random((4, 1, 256, 256)) < d).astype(np.float32)
        if d == dens:
            m = mm
    ks, kwin, sigma = 25, 9, 1.0
elif name == "c2":     # BASELINE configs[1]: the headline batch (for kernel timelines; bench.py is the measurement)
    sr, gt, m = synth.make_batch(16, 256, 256, seed0=100)
    ks, kwin, sigma = 25, 9, 0.004
elif name == "i1":     # one image of the C2 batch (what a per-image caller hands over)
    sr, gt, m = synth.make_batch(1, 256, 256, seed0=7)
    ks, kwin, sigma = 25, 9, 0.004
else:
    sr, gt, m = synth.make_batch(2, 512, 512, seed0=2000)
    ks, kwin, sigma = 25, 9, 0.004
    kw = dict(mask_stride=3)
B, C, H, W = sr.shape
n = int(m.sum())
eps, w = (1e-20, 5e2) if name == "c4" else (1e-10, 1e3)
step = engine.LossStep(B, C, H, W, ks, kwin, sigma, eps, True, w, w, device=dev, capacity=n + 64, graph=graph, **kw)
a, b, mm = (torch.as_tensor(x, device=dev) for x in (sr, gt, m))
for _ in range(10):
    step(a, b, mm)
torch.cuda.synchronize()
st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
st.record()
for _ in range(iters):
    step(a, b, mm)
en.record(); en.synchronize()
print(f"{name}{' graph' if graph else ''}: N={int(step.counts[0])} {st.elapsed_time(en) / iters:.4f} ms/step")
