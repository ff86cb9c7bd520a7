"""Throughput sweep over BASELINE.json's configs and mask densities (run on the GPU box):
   python tools/sweep.py > gpurun_out/sweep.txt
Reports ms/step and edge-px/s of the whole loss step (engine.LossStep) per configuration."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth

dev = torch.device("cuda:0")


def run(name, sr, gt, mask, ks, kw, sigma, iters):
    B, C, H, W = sr.shape
    n = int(mask.sum())
    step = engine.LossStep(B, C, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 64)
    a, b, m = (torch.as_tensor(x, device=dev) for x in (sr, gt, mask))
    for _ in range(max(3, iters // 5)):     # warm-up like bench.py (the first steps after an idle GPU run ~15 % slower)
        step(a, b, m)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        step(a, b, m)
    en.record(); en.synchronize()
    ms = st.elapsed_time(en) / iters
    alg = (8.0 * ks * ks + (12.0 * C + 4.0) * H * W * B / max(n, 1)) * n
    print(f"{name:58s} N={n:7d}  {ms:9.3f} ms/step  {n / ms / 1e3:8.2f} M edge-px/s  HBM(alg) {alg / ms / 1e6:7.1f} GB/s "
          f"({alg / ms / 1e6 / 8000 * 100:4.1f} %)", flush=True)


sr, gt, m = synth.uniform_case()
run("C1  1x3x64x64 5% fixed mask k_s=11 k_w=5 sigma=1", sr, gt, m, 11, 5, 1.0, 50)
sr, gt, m = synth.make_batch(16, 256, 256)
run("C2  16x3x256x256 Laplacian mask k_s=25 k_w=9 sigma=1.0", sr, gt, m, 25, 9, 1.0, 50)
run("C2' same, sigma=0.004 (training value)", sr, gt, m, 25, 9, 0.004, 50)
rng = np.random.default_rng(0)
for dens in (0.01, 0.04, 0.16, 0.5, 1.0):
    mm = (rng.random((4, 1, 256, 256)) < dens).astype(np.float32)
    run(f"density sweep 4x3x256x256 Bernoulli {dens:4.2f} k_s=25 k_w=9", sr[:4], gt[:4], mm, 25, 9, 1.0, 20 if dens > 0.3 else 50)
g = synth.natural_like(2000, 512, 512)[None]
s = synth.degrade(g[0], 2001)[None]
run("C5  1x3x512x512 DENSE mask k_s=49 k_w=13 sigma=1", s, g, np.ones((1, 1, 512, 512), np.float32), 49, 13, 1.0, 10)
