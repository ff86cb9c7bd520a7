"""Per-phase time breakdown of the fwd/bwd kernels via SSG_DEBUG_SKIP ablations (results are
wrong under ablation; this only measures time).  Run on the GPU box: python tools/ablate.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch, json
sys.path.insert(0, %r)
import bench
from ssl_amd import engine, synth
dev = torch.device("cuda:0")
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
n = int(mask_np.sum())
step = engine.LossStep(16, 3, 256, 256, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 1024)
step(sr, gt, mask); torch.cuda.synchronize()
t = bench.stage_times(step, sr, gt, mask, n, 10)
print("RESULT", json.dumps([t[k] for k in t if k.startswith("edge_list")] + [t["forward (all launches)"], t["backward (all launches)"]]))
''' % ROOT
def run(mask):
    env = dict(os.environ, SSG_DEBUG_SKIP=str(mask))
    out = subprocess.run([sys.executable, "-c", CODE], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    for l in out.splitlines():
        if l.startswith("RESULT"):
            return json.loads(l[6:])
    print(out[-2000:]); return [float('nan')] * 3
base = run(0)
print(f"baseline: edges {base[0]*1e3:.1f} us  fwd {base[1]:.3f} ms  bwd {base[2]:.3f} ms")
for name, m in (("fwd: no fill", 1), ("fwd: no compute", 2), ("fwd: no epilogue/store", 4), ("fwd: no fill+compute", 3), ("fwd: nothing but launch", 7)):
    r = run(m); print(f"  {name:28s} fwd {r[1]:.3f} ms  (saves {base[1]-r[1]:.3f})")
for name, m in (("bwd: no criteria math", 1), ("bwd: no pass A", 2), ("bwd: no pass B", 4), ("bwd: no atomics", 8), ("bwd: no A+B", 6), ("bwd: no A+B+atomics", 14), ("bwd: no A+B+atomics+criteria", 15)):
    r = run(m << 8); print(f"  {name:28s} bwd {r[2]:.3f} ms  (saves {base[2]-r[2]:.3f})")
