"""Where the deferred caller loop's wall time goes at C2: host time of each phase (loop, cat, criteria = the fused
forward, backward call) with and without a device synchronisation after each phase."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import synth
from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
dev = torch.device("cuda:0")
sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256, seed0=100)
sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
cri1, cri2 = L1Loss(1e3), KLDistanceLoss(1e3)
x = sr.clone().requires_grad_(True)
kw = dict(ssl_mode='cuda', kernel_size_search=25, generalization=True, kernel_size_window=9, sigma=1.0)
set_lazy(True)


def once(sync):
    t = [time.perf_counter()]
    def mark():
        if sync:
            torch.cuda.synchronize()
        t.append(time.perf_counter())
    x.grad = None
    out = x * 1.0
    a_list, b_list = [], []
    for i in range(16):
        m = mask[i, :].unsqueeze(0)
        if m.sum() == 0:
            continue
        a_list.append(similarity_map(img=out[i, :].unsqueeze(0).clone(), mask=m.clone(), **kw).getitem())
        b_list.append(similarity_map(img=gt[i, :].unsqueeze(0).clone(), mask=m.clone(), **kw).getitem())
    mark()
    A = torch.cat(a_list, dim=1); Bm = torch.cat(b_list, dim=1)
    mark()
    l1 = cri1(A, Bm)
    mark()
    kl = cri2(A, Bm)
    mark()
    (l1 + kl).backward()
    mark()
    torch.cuda.synchronize()
    t.append(time.perf_counter())
    return [(b - a) * 1e3 for a, b in zip(t[:-1], t[1:])]


def gpu_phases():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    x.grad = None
    out = x * 1.0
    a_list, b_list = [], []
    for i in range(16):
        m = mask[i, :].unsqueeze(0)
        if m.sum() == 0:
            continue
        a_list.append(similarity_map(img=out[i, :].unsqueeze(0).clone(), mask=m.clone(), **kw).getitem())
        b_list.append(similarity_map(img=gt[i, :].unsqueeze(0).clone(), mask=m.clone(), **kw).getitem())
    A = torch.cat(a_list, dim=1); Bm = torch.cat(b_list, dim=1)
    ev[0].record()
    l1 = cri1(A, Bm); kl = cri2(A, Bm)
    ev[1].record()
    (l1 + kl).backward()
    ev[2].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])


for _ in range(3):
    gpu_phases()
g = [gpu_phases() for _ in range(10)]
print("GPU time between events (asynchronous run): criteria (fused forward [+ speculative backward]) %.3f ms, backward %.3f ms"
      % (sum(a for a, _ in g) / 10, sum(b for _, b in g) / 10))

for sync in (False, True):
    for _ in range(3):
        once(sync)
    acc = None
    for _ in range(10):
        r = once(sync)
        acc = r if acc is None else [p + q for p, q in zip(acc, r)]
    names = ["loop (16 syncs, clones, handles)", "2 x torch.cat", "L1Loss (fused forward)", "KLDistanceLoss (cached)", ".backward() call", "final sync"]
    print("host ms per phase, %s:" % ("device synchronised after each phase" if sync else "asynchronous"), "  ".join("%s %.3f" % (n, v / 10) for n, v in zip(names, acc)), " total %.3f" % (sum(acc) / 10))
