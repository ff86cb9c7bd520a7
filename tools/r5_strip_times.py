"""Profiling build: start / end clock and CU of every ssg_fwd_strip workgroup of one fused C5 step."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import _lib, engine, synth
dev = torch.device("cuda:0")
with _lib.profile_build() as L:
    L.ssg_prof_strip_times.restype = ctypes.c_int
    L.ssg_prof_strip_times.argtypes = [ctypes.c_void_p, ctypes.c_int]
    H = W = 512
    gt = synth.natural_like(300, H, W)[None]; sr = synth.degrade(gt[0], 7)[None]; m = np.ones((1, 1, H, W), np.float32)
    step = engine.LossStep(1, 3, H, W, 49, 13, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=H * W + 64, materialise=False)
    a, b, mm = (torch.as_tensor(x, device=dev) for x in (sr, gt, m))
    for _ in range(4):
        step(a, b, mm)
    torch.cuda.synchronize()
    buf = np.zeros(3072, np.uint64)
    print("rc", L.ssg_prof_strip_times(buf.ctypes.data, 3072))
t = buf.reshape(-1, 3)[:480]
t0 = int(t[:, 0].min())
st, en = (t[:, 0].astype(np.int64) - t0), (t[:, 1].astype(np.int64) - t0)
hw = t[:, 2]
xcc, hwid = (hw >> np.uint64(32)).astype(np.int64), (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 16 + cu
print("clock ticks: kernel span", int(en.max()), " median wg duration", int(np.median(en - st)))
print("starts: <1%% of span: %d   later: %d" % (int((st < 0.01 * en.max()).sum()), int((st >= 0.01 * en.max()).sum())))
late = np.sort(st[st >= 0.01 * en.max()])
print("late starts (ticks):", late[:5], "...", late[-5:] if len(late) else "")
u, c = np.unique(key, return_counts=True)
print("distinct (xcc,se,sh,cu):", len(u), " workgroups per CU histogram:", dict(zip(*np.unique(c, return_counts=True))))
d = en - st
for k in (1, 2, 3):
    sel = np.isin(key, u[c == k])
    if sel.any(): print("CUs with %d wg: mean wg duration %d ticks" % (k, int(d[sel].mean())))

# per XCC (the cycle counters of different XCDs are not synchronised): when do the workgroups start / end?
for x in np.unique(xcc)[:3]:
    sel = xcc == x
    s0 = t[sel, 0].astype(np.int64); e0 = t[sel, 1].astype(np.int64); base = s0.min()
    order = np.argsort(s0)
    print("XCC %d: %d workgroups; starts (k ticks): %s" % (x, sel.sum(), np.round((s0[order] - base) / 1e3).astype(int).tolist()))
    print("        ends   (k ticks): %s" % np.round((e0[order] - base) / 1e3).astype(int).tolist())
