"""C5 (1x3x512x512, dense mask, k_s 49, k_w 13) and C2 stage timings + C5 parity on sampled rows vs the oracle."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from ssl_amd import engine, synth, _lib
from oracle import ssg_oracle as orc
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(a, device=dev)

def ev(fn, it):
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); st.record()
    for _ in range(it): fn()
    en.record(); en.synchronize(); return st.elapsed_time(en) / it

# small (49,13) parity: dense mask 96x128 vs oracle
H, W = 72, 100
g = synth.natural_like(21, H, W)[None]; s = synth.degrade(g[0], 5)[None]
for dens in (1.0, 0.3):
    rng = np.random.default_rng(3)
    m = (rng.random((1, 1, H, W)) < dens).astype(np.float32)
    ref = orc.ssg_loss(s.astype(np.float64), g.astype(np.float64), m[:, 0], 49, 13, 1.0, 1e3, 1e3)
    for thr in (0, 1):
        engine.set_dense_threshold(thr)
        step = engine.LossStep(1, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev)
        loss, grad = step(T(s), T(g), T(m)); torch.cuda.synchronize()
        n = int(step.counts[0])
        se = np.abs(step.ssg_sr[:n].cpu().numpy() - ref["s_sr"]).max()
        ge = np.abs(grad.cpu().numpy() - ref["grad"]).max() / np.abs(ref["grad"]).max()
        print(f"(49,13) {H}x{W} dens {dens} thr {thr}: N {n} loss {loss.cpu().numpy()} ref {ref['l1']:.6f} {ref['kl']:.6g} max|dSSG| {se:.2e} grad err {ge:.2e}")

# C5 timing
g = synth.natural_like(300, 512, 512)[None]; s = synth.degrade(g[0], 7)[None]
m = np.ones((1, 1, 512, 512), np.float32)
sr, gt, mask = T(s), T(g), T(m)
for thr in (1, 0):
    engine.set_dense_threshold(thr)
    step = engine.LossStep(1, 3, 512, 512, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev)
    t = ev(lambda: step(sr, gt, mask), 3)
    print(f"C5 thr {thr}: {t:.2f} ms/step  -> {512*512/t/1e3:.2f} M edge-px/s   loss {step.loss.cpu().numpy()}")
