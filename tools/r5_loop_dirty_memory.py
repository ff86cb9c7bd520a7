"""The deferred C2 loop on dirty allocator memory: which SSG rows / scales / losses differ from the first (clean) run?
(How the unwritten-LDS merge flag of band_scatter was found: profiles/EXPERIMENTS.md, round 5.)"""
import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
from ssl_amd import engine, synth
from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy, lazy
from ssl_amd.reference_loop import gan_selfsim_block
dev = torch.device("cuda:0"); T = lambda a: torch.as_tensor(a, device=dev)
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256, seed0=100)
n = int(mask_np.sum())
setting = dict(ssl_mode='cuda', kernel_size_search=25, generalization=True, kernel_size_window=9, sigma=0.004)
snap = {}
orig = lazy._LazyStepFn.forward
def spy(ctx, *a):
    out = orig(ctx, *a)
    x, edges, counts, ssg_sr, ssg_gt = ctx.to_save
    order, rank, plan, rsc = ctx.keep
    snap.update(edges=edges[:ctx.n].clone(), counts=counts.clone(), ssg_sr=ssg_sr.clone(), ssg_gt=ssg_gt.clone(), rsc=rsc.clone(),
                rank=rank.clone(), plan=plan.clone(), x=x.clone(), loss=torch.stack([out[0], out[1]]).clone())
    return out
lazy._LazyStepFn.forward = staticmethod(spy)
base = None
def both(tag, with_step=True):
    global base
    if with_step:
        step = engine.LossStep(16, 3, 256, 256, 25, 9, 0.004, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 64)
        loss, grad = step(T(sr_np), T(gt_np), T(mask_np))
    sr = T(sr_np).requires_grad_(True)
    l1, kl = gan_selfsim_block(similarity_map, L1Loss(1e3), KLDistanceLoss(1e3), sr * 1.0, T(gt_np), T(mask_np), setting)
    (l1 + kl).backward()
    cur = {k: v.cpu() for k, v in snap.items()}
    cur["grad"] = sr.grad.cpu()
    if base is None:
        base = cur
    bad = [k for k in cur if k != "plan" and not torch.equal(cur[k].view(torch.uint8), base[k].view(torch.uint8))]
    print(tag, with_step, "loop %.9g %.9g gradsum %.9g" % (float(l1.detach()), float(kl.detach()), float(sr.grad.abs().sum())), bad)
    for k in bad:
        if k in ("ssg_sr", "ssg_gt"):
            d = (cur[k].view(torch.int32) != base[k].view(torch.int32))
            rows = d.any(1).nonzero().reshape(-1)
            print("   ", k, len(rows), "rows differ:", rows[:12].tolist(), "elements/row", d[rows[:6]].sum(1).tolist())
            r0 = int(rows[0])
            cols = d[r0].nonzero().reshape(-1)
            print("    row", r0, "edge", cur["edges"][r0].tolist(), "cols", cols[:10].tolist(), "...", cols[-3:].tolist(),
                  "now", cur[k][r0, cols[:4]].tolist(), "base", base[k][r0, cols[:4]].tolist(), "rsc now/base", float(cur["rsc"][r0]), float(base["rsc"][r0]))
            e = cur["edges"][rows[:200]]
            print("    images", sorted(set(e[:, 0].tolist())), "tile rows", sorted(set((e[:, 1] // 8).tolist()))[:10], "tile cols", sorted(set((e[:, 2] // 32).tolist()))[:10])
        if k == "rsc":
            d = (cur[k].view(torch.int64) != base[k].view(torch.int64)).nonzero().reshape(-1)
            print("    rsc", len(d), d[:10].tolist(), cur[k][d[:4]].tolist(), base[k][d[:4]].tolist())
both("clean")
for rep in range(3):
  for ws in (True, False):
    for fill in (3, 2, 4, 5, 3):
        junk = [torch.full((64 * 1024 * 1024,), fill, dtype=torch.int32, device=dev) for _ in range(12)]
        del junk
        both("dirty int %d" % fill, ws)
