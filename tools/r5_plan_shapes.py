"""Plan headers (rows for the direct kernels, dense tiles) of the benchmark configurations: what the stream-assignment /
two-chain estimate of ssg_api.hip sees.   python tools/r5_plan_shapes.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
def show(name, m, stride=0):
    el = engine.edge_list(mask=T(m.astype(np.float32)), mask_stride=stride, ks=25, order=False)
    p = el.plan[:4].cpu().numpy()
    n = int(el.counts[0])
    ns, nt = int(p[0]), int(p[1] + p[3])
    print("%-12s rows %6d  direct %6d  dense tiles %5d (heavy %d)  direct/dense estimate %.2f" % (name, n, ns, nt, p[1], ns * 38 / max(1, nt * 740)))
_, _, m = synth.make_batch(16, 256, 256, seed0=100); show("C2", m[:, :1])
_, _, m = synth.make_batch(2, 512, 512, seed0=2000); show("C4 stride 3", m[:, :1], 3)
rng = np.random.default_rng(0)
for d in (0.01, 0.04, 0.1, 0.2):
    show("Bernoulli %g" % d, rng.random((4, 1, 256, 256)) < d)
_, _, m = synth.make_batch(1, 256, 256, seed0=7); show("one image", m[:, :1])
