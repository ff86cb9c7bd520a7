#!/usr/bin/env python3
"""Where the two-launch C1 step (ssl_amd/csrc/ssg_tiny.hip) spends its time: LossStep at BASELINE's configs[0] on the
PROFILING build with phases of ssg_tiny_step switched off (SSG_TINY_DBG; the product library has no such switch and reads no
environment): 1 no forward sums, 2 no backward, 4 no gradient atomics, 8 no fold by the last workgroup, 16 no rows at all.

    python tools/r6_tiny_ablate.py            ->  one line per mask (wall clock per step over 5 x 2,000 steps, best block)
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MASKS = [("whole step", 0), ("no forward sums", 1), ("no backward", 2), ("no gradient atomics", 4), ("no fold (last workgroup)", 8),
         ("no backward, atomics, fold", 14), ("no rows", 16), ("two empty launches + edge list", 31)]


def child():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from ssl_amd import _lib, engine, synth
    dev = torch.device("cuda:0")
    gt = synth.natural_like(4064, 64, 64, 0.10, 0.04)[None].astype(np.float32)
    sr = synth.degrade(gt[0], 4164, 0.05)[None].astype(np.float32)
    m = (np.random.default_rng(4064).random((1, 1, 64, 64)) < 0.05).astype(np.float32)
    srt, gtt, mt = (torch.as_tensor(a, device=dev) for a in (sr, gt, m))
    with _lib.profile_build():
        step = engine.LossStep(1, 3, 64, 64, 11, 5, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=int(m.sum()) + 1024, deterministic=True)
        for _ in range(300):
            step(srt, gtt, mt)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(2000):
                step(srt, gtt, mt)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 2000)
    print("%.2f" % (best * 1e6))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        sys.exit(0)
    print("# C1 (1 x 3 x 64 x 64, ~200 edge px, (11,5)), LossStep on libssg_hip_prof.so, us per step")
    for name, mask in MASKS:
        env = dict(os.environ, SSG_TINY_DBG=str(mask))
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, stdout=subprocess.PIPE, text=True).stdout.strip().splitlines()
        print("%-34s SSG_TINY_DBG=%-3d %s us" % (name, mask, out[-1] if out else "failed"))
