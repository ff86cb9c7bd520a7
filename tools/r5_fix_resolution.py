"""Deterministic (fixed-point, a-priori scale) against fp32-atomic accumulation of the fused step's gradient, and both
against an fp64 evaluation of the same G rows is not available here -- so: |g_det - g_atomic| / max|g| per configuration.
The fp32 atomics' own run-to-run spread is printed beside it (two atomic runs).   python tools/r5_fix_resolution.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
sr, gt, m = synth.make_batch(4, 128, 128)
n = int(m.sum())
for ks, kw in ((25, 9), (11, 5)):
    for sigma in (0.004, 0.05, 1.0):
        for w in (1e3, 1.0):
            g = {}
            for tag, det in (("det", True), ("atomic", False), ("atomic2", False)):
                step = engine.LossStep(4, 3, 128, 128, ks, kw, sigma, 1e-10, True, w, w, device=dev, capacity=n + 64, deterministic=det)
                loss, grad = step(T(sr), T(gt), T(m))
                g[tag] = grad.double().clone()
            mx = float(g["atomic"].abs().max())
            print("(%d,%d) sigma %-5g w %-6g max|g| %.3e   det vs atomic %.2e   atomic vs atomic %.2e" % (
                ks, kw, sigma, w, mx, float((g["det"] - g["atomic"]).abs().max()) / mx, float((g["atomic2"] - g["atomic"]).abs().max()) / mx))
