import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from ssl_amd import engine, synth
dev = torch.device("cuda:0"); T = lambda a: torch.as_tensor(a, device=dev)
B, H, W = 4, 512, 512
gt = np.stack([synth.natural_like(300 + i, H, W) for i in range(B)]); sr = np.stack([synth.degrade(gt[i], 7 + i) for i in range(B)])
mask = torch.ones((B, 1, H, W), device=dev)
for mat in (False, True):
    st = engine.LossStep(B, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=B * H * W, materialise=mat)
    l, g = st(T(sr), T(gt), mask); l = l.cpu().numpy(); g = g.clone()
    ls, worst = [], 0.0
    for i in range(B):
        one = engine.LossStep(1, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=H * W, materialise=mat)
        l1, g1 = one(T(sr[i:i + 1]), T(gt[i:i + 1]), mask[:1]); ls.append(l1.cpu().numpy())
        worst = max(worst, float((g[i] - g1[0] / B).abs().max() / (g1[0].abs().max() / B)))
        del one
    print("materialise", mat, "batch loss", l, "mean of singles", np.mean(ls, 0), "worst grad rel dev", worst)
    del st; torch.cuda.empty_cache()
