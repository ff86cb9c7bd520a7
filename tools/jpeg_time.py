"""Time datapath.DiffJPEG on a 16 x 3 x 400 x 400 batch (per-sample quality)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import datapath
x = torch.rand(16, 3, 400, 400, device="cuda")
q = torch.empty(16, device="cuda").uniform_(30, 95)
m = datapath.DiffJPEG(differentiable=False)
m(x, q); torch.cuda.synchronize()
print("DiffJPEG 16x3x400x400: %.3f ms" % bench.event_time_ms(lambda: m(x, q), 20))
