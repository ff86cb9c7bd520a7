"""Gradient / SSG errors of the k_s 49 steps against the fp64 oracle on one dense 64 x 96 case: the materialising step
(row-major rows) and the fused step (tile-major rows; SSG_STRIPS=0/1 selects the forward that computes them)."""
import os
import sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import ssg_oracle as orc
from ssl_amd import engine, synth
from test_gpu_parity import tile_major_ssg, ref_grad_with_gpu_signs

dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(a, device=dev)
for sigma in (1.0, 0.05, 0.004):
    B, H, W, ks, kw = 1, 64, 96, 49, 13
    gt = np.stack([synth.natural_like(700 + i, H, W) for i in range(B)])
    sr = np.stack([synth.degrade(gt[i], 750 + i) for i in range(B)])
    mask = np.ones((B, 1, H, W), np.float32)
    ref = orc.ssg_loss(sr.astype(np.float64), gt.astype(np.float64), mask[:, 0], ks, kw, sigma, 1e3, 1e3)
    a = engine.LossStep(B, 3, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=True)
    b = engine.LossStep(B, 3, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=True, materialise=False)
    la, ga = a(T(sr), T(gt), T(mask))
    lb, gb = b(T(sr), T(gt), T(mask))
    torch.cuda.synchronize()
    n = int(a.counts[0])
    sa = (a.ssg_sr[:n].cpu().numpy(), a.ssg_gt[:n].cpu().numpy())
    sb = tile_major_ssg(b)
    mx = np.abs(ref["grad"]).max()
    for name, g, s in (("materialised", ga, sa), ("fused", gb, sb)):
        gref, nflip = ref_grad_with_gpu_signs(sr, mask[:, 0], ks, kw, sigma, ref, s[0], s[1])
        rel = np.abs(s[0].astype(np.float64) / ref["s_sr"] - 1)
        rows = np.abs(s[0].astype(np.float64).sum(1) - ref["s_sr"].sum(1)).max()
        print(f"sigma {sigma} {name:13s} STRIPS={os.environ.get('SSG_STRIPS', '1')}: grad err {np.abs(g.cpu().numpy() - gref).max() / mx:.2e} "
              f"(sign ties {nflip})  SSG abs {np.abs(s[0] - ref['s_sr']).max():.2e} rel max {rel.max():.2e} rel rms {np.sqrt((rel ** 2).mean()):.2e} "
              f"row-sum err {rows:.2e}  kl {float((la if name[0] == 'm' else lb)[1]):.6g} (ref {ref['kl']:.6g})", flush=True)
