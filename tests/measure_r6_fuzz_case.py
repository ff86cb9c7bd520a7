"""One case of tools/r5_fuzz_paths.py (seed, case index) replayed and priced against the fp64 ORACLE: the all-direct step,
the mixed dense + direct step and the oracle's gradient with the GPU's own L1 signs at near-ties (tests/test_gpu_parity.py:
ref_grad_with_gpu_signs).  Round 6: case 659 of seed 61 was flagged by the fuzz (gradient of the two paths 2.5e-2 of its maximum
apart at sigma 0.004) -- which of the two, if any, is wrong?     python tests/measure_r6_fuzz_case.py 61 659"""
import importlib.util, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ssl_amd import engine, synth
from oracle import ssg_oracle as orc
import test_gpu_parity as tp

seed, target = int(sys.argv[1]), int(sys.argv[2])
sys.argv = [sys.argv[0], "0", str(seed)]           # the fuzz tool with zero cases: its generators and its rng
spec = importlib.util.spec_from_file_location("fuzz", os.path.join(ROOT, "tools", "r5_fuzz_paths.py"))
fz = importlib.util.module_from_spec(spec)
try:
    spec.loader.exec_module(fz)
except SystemExit:
    pass
rng = fz.rng
dev = torch.device("cuda:0")
for case in range(target + 1):      # the same draws in the same order as fz.run()
    ks, kw = (49, 13) if rng.random() < 0.12 else (25, 9)
    B = int(rng.integers(1, 5))
    lo = ks // 2 + 2
    H = int(rng.integers(lo + 8, 140 if ks == 49 else 330))
    W = int(rng.integers(lo + 8, 140 if ks == 49 else 600 if rng.random() < 0.2 else 300))
    if B * H * W > 300000:
        B = 1
    kind = rng.choice(["bernoulli", "laplacian", "strokes", "blobs", "right_edge"])
    stride = int(rng.choice([0, 0, 0, 2, 3]))
    gt = np.stack([synth.natural_like(int(rng.integers(1 << 20)), H, W) for _ in range(B)])
    sr = np.stack([synth.degrade(gt[i], int(rng.integers(1 << 20))) for i in range(B)])
    m = fz.make_mask(kind, B, H, W, gt)
    thr = int(rng.choice([1, 4, 12, 18, 18, 18, 30, 64]))
    sigma = float(rng.choice([0.004, 0.05, 1.0]))
print("case", target, "ks", ks, "B", B, "H", H, "W", W, kind, "stride", stride, "thr", thr, "sigma", sigma, "n", int(m.sum()))
eff = np.stack([orc.mask_stride(m[i].astype(np.float32), stride) for i in range(B)]) if stride > 1 else m.astype(np.float32)
ref = orc.ssg_loss(sr.astype(np.float64), gt.astype(np.float64), eff, ks, kw, sigma, 1e3, 1e3)
mx = np.abs(ref["grad"]).max()
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
for t in (0, thr):
    prev = engine.set_dense_threshold(t)
    try:
        step = engine.LossStep(B, 3, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, mask_stride=stride)
        loss, grad = step(T(sr), T(gt), T(m[:, None].astype(np.float32)))
        n = int(step.counts[0])
        g = grad.cpu().numpy().astype(np.float64)
        gref, nflip = tp.ref_grad_with_gpu_signs(sr, eff, ks, kw, sigma, ref, step.ssg_sr[:n].cpu().numpy(), step.ssg_gt[:n].cpu().numpy())
        print("threshold %2d: n %d  |grad - oracle| / max %.2e   with the GPU's signs at %d near-ties: %.2e   rows %.2e / %.2e   loss rel %.2e %.2e"
              % (t, n, np.abs(g - ref["grad"]).max() / mx, nflip, np.abs(g - gref).max() / mx,
                 np.abs(step.ssg_sr[:n].cpu().numpy() - ref["s_sr"]).max(), np.abs(step.ssg_gt[:n].cpu().numpy() - ref["s_gt"]).max(),
                 abs(float(loss[0]) - ref["l1"]) / ref["l1"], abs(float(loss[1]) - ref["kl"]) / max(ref["kl"], 1e-30)))
    finally:
        engine.set_dense_threshold(prev)
