"""Randomised hunt: shared-term kernels (threshold 1 or 28) vs direct kernels (threshold 0) over random shapes,
densities and capacities -- SSG rows (2e-6), losses (1e-5), gradients (deterministic mode, 2e-5 of max), and the
fused step against the materialising one (bits; k_s 49: tolerances).  Usage: python tests/measure_fuzz_dense_vs_direct.py [cases] [seed]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(ncases):
    ks, kw = ((25, 9), (49, 13))[int(rng.integers(2))]
    B = int(rng.integers(1, 4))
    H = int(rng.integers(ks // 2 + 1 + 8, 90)); W = int(rng.integers(ks // 2 + 1 + 8, 130))
    dens = float(rng.choice([0.03, 0.1, 0.3, 0.7, 1.0]))
    sigma = float(rng.choice([1.0, 0.05, 0.004]))
    thr = int(rng.choice([1, 8, 28]))
    gt = np.stack([synth.natural_like(int(rng.integers(1 << 20)), H, W) for _ in range(B)])
    sr = np.stack([synth.degrade(gt[i], int(rng.integers(1 << 20))) for i in range(B)])
    kind = int(rng.integers(3))
    if kind == 0:
        mask = (rng.random((B, 1, H, W)) < dens).astype(np.float32)
    elif kind == 1:
        mask = np.stack([synth.laplacian_edge_mask(gt[i]) for i in range(B)])[:, None].astype(np.float32)
    else:   # blocks of full density
        mask = np.zeros((B, 1, H, W), np.float32)
        for b in range(B):
            y0, x0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
            mask[b, 0, y0:y0 + int(rng.integers(4, 40)), x0:x0 + int(rng.integers(4, 70))] = 1
    n = int(mask.sum())
    if n == 0:
        continue
    res = {}
    for t in (0, thr):
        prev = engine.set_dense_threshold(t)
        try:
            st = engine.LossStep(B, 3, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=True)
            loss, grad = st(T(sr), T(gt), T(mask))
            res[t] = (loss.cpu().numpy().copy(), grad.cpu().numpy().copy(), st.ssg_sr[:n].cpu().numpy().copy(),
                      st.ssg_gt[:n].cpu().numpy().copy())
            if t:
                fu = engine.LossStep(B, 3, H, W, ks, kw, sigma, 1e-10, True, 1e3, 1e3, device=dev, deterministic=True,
                                     materialise=False)
                lf, gf = fu(T(sr), T(gt), T(mask))
                if ks == 49:   # (may run on tile-major rows: same numbers up to fp32 rounding and L1 sign ties)
                    gm = float(grad.abs().max())
                    same = (abs(float(lf[0] - loss[0])) <= 1e-6 * abs(float(loss[0])) and
                            abs(float(lf[1] - loss[1])) <= 1e-4 * abs(float(loss[1])) and
                            float((gf - grad).abs().max()) <= (1e-3 if sigma < 0.01 else 1e-4) * gm)
                else:
                    same = torch.equal(lf, loss) and torch.equal(gf, grad)
                if not same:
                    bad += 1
                    print("FUSED MISMATCH", ks, B, H, W, dens, sigma, thr)
        finally:
            engine.set_dense_threshold(prev)
    (l0, g0, a0, b0), (l1, g1, a1, b1) = res[0], res[thr]
    ea, eb = np.abs(a0 - a1).max(), np.abs(b0 - b1).max()
    el = np.abs(l0 - l1) / np.maximum(np.abs(l0), 1e-30)
    eg = np.abs(g0 - g1).max() / max(np.abs(g0).max(), 1e-30)
    # (sigma = 1 on natural images: the rows are nearly flat, KL and the gradient are second-order quantities of
    # differences at fp32 resolution -- two fp32 evaluations agree to ~1e-4 / ~2e-3 only; the parity tests bound each
    # path against the fp64 oracle instead)
    flat = sigma >= 1.0
    ok = ea <= 2e-6 and eb <= 2e-6 and el[0] <= 2e-5 and el[1] <= (1e-3 if flat else 2e-5) and eg <= (5e-3 if flat else 2e-3)
    if not ok or not np.isfinite(g1).all():
        bad += 1
    print(("ok  " if ok else "BAD ") + f"ks={ks} B={B} {H}x{W} kind={kind} dens={dens} sigma={sigma} thr={thr} n={n} "
          f"dSSG={max(ea, eb):.1e} dloss={el.max():.1e} dgrad={eg:.1e}")
print("bad cases:", bad)
