"""Step time against the dense-tile threshold (ssg_set_dense_threshold) under the round-5 schedules: C2, C4, one image.
   python tools/r5_thr_sweep.py [thr ...]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
thrs = [int(a) for a in sys.argv[1:]] or [10, 14, 18, 22, 28]
cases = {"c2": (synth.make_batch(16, 256, 256, seed0=100), 1.0, 0, 1e-10, 1e3),
         "c4": (synth.make_batch(2, 512, 512, seed0=2000), 0.004, 3, 1e-20, 5e2),
         "i1": (synth.make_batch(1, 256, 256, seed0=7), 0.004, 0, 1e-10, 1e3)}
def run(step, a, b, m, iters):
    for _ in range(30): step(a, b, m)
    torch.cuda.synchronize()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters): step(a, b, m)
    en.record(); en.synchronize()
    return st.elapsed_time(en) / iters
for name, ((sr, gt, m), sigma, stride, eps, w) in cases.items():
    B, C, H, W = sr.shape
    a, b, mm = T(sr), T(gt), T(m[:, :1])
    res = []
    for rep in range(2):
        for t in thrs:
            prev = engine.set_dense_threshold(t)
            try:
                n = int(engine.edge_list(mask=mm, mask_stride=stride).counts[0])
                step = engine.LossStep(B, C, H, W, 25, 9, sigma, eps, True, w, w, device=dev, mask_stride=stride, capacity=n + 1024)
                res.append((t, run(step, a, b, mm, 100 if name == "c2" else 300)))
            finally:
                engine.set_dense_threshold(prev)
    print(name, " ".join("thr %d: %.4f" % r for r in res))
