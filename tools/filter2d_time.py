"""Time datapath.filter2D on a Real-ESRGAN-sized GT batch (16 x 3 x 400 x 400) with per-sample 9x9 and 21x21 kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ssl_amd import datapath
x = torch.rand(16, 3, 400, 400, device="cuda")
for k in (9, 21):
    kern = torch.rand(16, k, k, device="cuda"); kern /= kern.sum(dim=(1, 2), keepdim=True)
    datapath.filter2D(x, kern); torch.cuda.synchronize()
    print("filter2D 16x3x400x400 k=%d: %.3f ms" % (k, bench.event_time_ms(lambda: datapath.filter2D(x, kern), 20)))
