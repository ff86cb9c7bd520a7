"""Time datapath.USMSharp on a Real-ESRGAN-sized GT batch (16 x 3 x 400 x 400)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from ssl_amd import datapath
x = torch.rand(16, 3, 400, 400, device="cuda")
m = datapath.USMSharp()
m(x); torch.cuda.synchronize()
print("USM 16x3x400x400 k=%d: %.3f ms (module call, HIP events)" % (m.radius, bench.event_time_ms(lambda: m(x), 20)))
