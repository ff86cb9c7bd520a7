import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr, gt, m = synth.make_batch(16, 256, 256)
mm = torch.as_tensor(m, device=dev)
for thr in (16, 18, 20):
    old = engine.set_dense_threshold(thr)
    el = engine.edge_list(mask=mm, capacity=int(m.sum()) + 1024, ks=25, order=False, plan=True)
    plan = el.plan.cpu().numpy()
    print("thr", thr, "rows", int(el.counts[0]), "direct rows", plan[0], "dense tiles", plan[1] + plan[3], "(heavy", plan[1], ")")
    engine.set_dense_threshold(old)
