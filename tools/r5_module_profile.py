"""Host-side profile of the drop-in module at a launch-bound size (C1-like): where the ~0.19 ms per call go.
   python tools/r5_module_profile.py"""
import sys, os, time, cProfile, pstats, io, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import SSGLoss, synth
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
sr, gt, m = synth.uniform_case()
crit = SSGLoss(11, 5, 1.0, True, 1e3, 1e3)
x = T(sr).clone().requires_grad_(True); g = T(gt); mk = T(m)
def one():
    x.grad = None
    a, b = crit(x, g, mk)
    (a + b).backward()
for _ in range(20): one()
torch.cuda.synchronize()
import gc; gc.collect()
t0 = time.perf_counter()
for _ in range(300): one()
t1 = time.perf_counter(); torch.cuda.synchronize()
print("host %.1f us per call (forward + backward)" % ((t1 - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): one()
pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(18); print(st.getvalue()[:3500])
