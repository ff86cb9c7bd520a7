"""Profiling build: the fused C5 step with phases of ssg_fwd_strip ablated (results wrong, timing only).
   python tools/r5_strip_ablate.py        -> step ms (HIP events) per mask; the difference to `full` is the phase's cost"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import _lib, engine, synth
dev = torch.device("cuda:0")
H = W = 512
gt = synth.natural_like(300, H, W)[None]; sr = synth.degrade(gt[0], 7)[None]; m = np.ones((1, 1, H, W), np.float32)
with _lib.profile_build() as L:
    step = engine.LossStep(1, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=H * W + 64, materialise=False)
    a, b, mm = (torch.as_tensor(x, device=dev) for x in (sr, gt, m))
    def t(mask, n=6):
        L.ssg_set_profile_mask(mask)
        for _ in range(2): step(a, b, mm)
        torch.cuda.synchronize()
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record()
        for _ in range(n): step(a, b, mm)
        en.record(); en.synchronize()
        L.ssg_set_profile_mask(0)
        return st.elapsed_time(en) / n
    B = 16
    rest = t((1 << 25) | (1 << 26))          # no forward launches at all: the rest of the step
    for name, mask in (("full", 0), ("no stores", 1 << B), ("no edge stage", 2 << B), ("no E/H stage", 4 << B),
                       ("no edge, no E/H (barriers only)", 6 << B),
                       ("edge only", 4 << B), ("edge only, no stores", 5 << B), ("edge only, no exp/rowsum/store", (4 + 8) << B),
                       ("edge only, no gathers", (4 + 16) << B), ("edge only, gathers alone", (4 + 8 + 1) << B),
                       ("E/H only", 2 << B), ("E/H only, no H stores", (2 + 32) << B), ("E/H only, no window reads", (2 + 128) << B),
                       ("E/H only, no LDS at all", (2 + 32 + 128) << B), ("full again", 0)):
        ms = t(mask)
        print("%-34s step %.3f ms   forward ~ %.3f ms" % (name, ms, ms - rest))
