"""Tile-major fused step (k_s 49) against the materialising step on the same inputs: loss / gradient deviations for
all-dense, ragged, half-dense (mixed tile-major / row-major tiles: capacity = N) and two-image cases; run-to-run bits."""
import numpy as np
import torch
from ssl_amd import engine, synth

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(a, device=dev)


def case(name, B, H, W, density, cap_is_n=False, sigma=1.0, det=True):
    gt = np.stack([synth.natural_like(900 + i, H, W) for i in range(B)])
    sr = np.stack([synth.degrade(gt[i], 950 + i) for i in range(B)])
    rng = np.random.default_rng(3)
    mask = (rng.random((B, 1, H, W)) < density).astype(np.float32)
    cap = int(mask.sum()) if cap_is_n else None
    a = engine.LossStep(B, 3, H, W, 49, 13, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=det, capacity=cap)
    b = engine.LossStep(B, 3, H, W, 49, 13, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=det, capacity=cap,
                        materialise=False)
    la, ga = a(T(sr), T(gt), T(mask))
    la, ga = la.clone(), ga.clone()
    lb, gb = b(T(sr), T(gt), T(mask))
    lb, gb = lb.clone(), gb.clone()
    lb2, gb2 = b(T(sr), T(gt), T(mask))
    torch.cuda.synchronize()
    gm = float(ga.abs().max())
    print(f"{name}: N={int(a.counts[0])}/{int(b.counts[0])} l1 {float(la[0]):.6g} vs {float(lb[0]):.6g} rel {abs(float(la[0]-lb[0]))/abs(float(la[0])):.2e}  "
          f"kl {float(la[1]):.6g} vs {float(lb[1]):.6g} rel {abs(float(la[1]-lb[1]))/abs(float(la[1])):.2e}  "
          f"grad max {gm:.3e} dev {float((ga-gb).abs().max())/gm:.2e} finite {bool(torch.isfinite(gb).all())} "
          f"rerun equal {torch.equal(gb, gb2) and torch.equal(lb, lb2)}", flush=True)


case("dense 64x96", 1, 64, 96, 1.1)
case("dense ragged 70x100", 1, 70, 100, 1.1)
case("half 96x160 cap=N (row-major: too many tiles)", 1, 96, 160, 0.5, cap_is_n=True)
case("0.8 96x160 (tile-major with holes)", 1, 96, 160, 0.8)
case("0.8 ragged 70x100 B=2", 2, 70, 100, 0.8)
case("0.45 96x160 (row-major: tiles < 60 % full)", 1, 96, 160, 0.45)
case("B=2 0.7", 2, 64, 96, 0.7)
case("dense sigma .05", 1, 64, 96, 1.1, sigma=0.05)
case("dense fp32 atomics", 1, 64, 96, 1.1, det=False)
