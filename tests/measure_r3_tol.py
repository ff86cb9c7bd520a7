"""Measure (GPU box) the errors the full-size tests bound: KL vs fp64 KL of the same fp32 SSGs, full C2 gradient vs
the fp64 oracle, C5 window losses/gradient vs the fp64 oracle."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import ssg_oracle as orc
from ssl_amd import engine, synth
sys.path.insert(0, "tests")
dev = torch.device("cuda:0")
T = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)

def kl64(s_sr, s_gt):
    return 1e3 * torch.nn.functional.kl_div(s_sr.clamp(min=1e-10).double().log(), s_gt.clamp(min=1e-10).double(), reduction="mean")

sr, gt, mask = synth.make_batch(16, 256, 256)
for sigma in (1.0, 0.004):
    step = engine.LossStep(16, 3, 256, 256, 25, 9, sigma, 1e-10, True, 1e3, 1e3, device=dev, capacity=int(mask.sum()) + 1000)
    loss, grad = step(T(sr), T(gt), T(mask))
    n = int(step.counts[0])
    s_sr, s_gt = step.ssg_sr[:n], step.ssg_gt[:n]
    l1 = 1e3 * (s_sr - s_gt).abs().double().mean(); kl = kl64(s_sr, s_gt)
    print(f"C2 sigma={sigma}: l1 rel {abs(float(loss[0]) - float(l1)) / float(l1):.2e}  kl {float(loss[1]):.6e} vs fp64-of-same-SSGs {float(kl):.6e} rel {abs(float(loss[1]) - float(kl)) / float(kl):.2e}")
    t0 = time.time()
    ref = orc.ssg_loss(sr.astype(np.float64), gt.astype(np.float64), mask[:, 0], 25, 9, sigma, 1e3, 1e3)
    t1 = time.time()
    r32 = orc.ssg_loss(sr, gt, mask[:, 0], 25, 9, sigma, 1e3, 1e3)
    mx = np.abs(ref["grad"]).max()
    g = grad.cpu().numpy().astype(np.float64)
    print(f"   fp64 oracle {t1 - t0:.1f}s, fp32 oracle {time.time() - t1:.1f}s; l1 rel {abs(float(loss[0]) - ref['l1']) / ref['l1']:.2e} kl rel vs fp64 oracle {abs(float(loss[1]) - ref['kl']) / ref['kl']:.2e} (fp32 oracle kl rel {abs(r32['kl'] - ref['kl']) / ref['kl']:.2e})")
    print(f"   grad: gpu-vs-fp64 {np.abs(g - ref['grad']).max() / mx:.2e} of max; fp32 oracle vs fp64 {np.abs(r32['grad'] - ref['grad']).max() / mx:.2e}; max|dSSG| {np.abs(s_sr.cpu().numpy() - ref['s_sr']).max():.2e}")
    from test_gpu_parity import ref_grad_with_gpu_signs
    gref, nflip = ref_grad_with_gpu_signs(sr, mask[:, 0], 25, 9, sigma, ref, s_sr.cpu().numpy(), s_gt.cpu().numpy())
    print(f"   grad with GPU signs at {nflip} undecided entries: {np.abs(g - gref).max() / mx:.2e}")

ks, kw, H, W = 49, 13, 512, 512
gt5 = synth.natural_like(300, H, W)[None]; sr5 = synth.degrade(gt5[0], 7)[None]
m = np.zeros((1, 1, H, W), np.float32); m[0, 0, :16, :64] = 1; m[0, 0, H - 16:, W - 64:] = 1; m[0, 0, 240:256, 200:264] = 1
step = engine.LossStep(1, 3, H, W, ks, kw, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=int(m.sum()) + 64)
loss, grad = step(T(sr5), T(gt5), T(m))
n = int(step.counts[0])
t0 = time.time()
ref = orc.ssg_loss(sr5.astype(np.float64), gt5.astype(np.float64), m[:, 0], ks, kw, 1.0, 1e3, 1e3)
t1 = time.time()
r32 = orc.ssg_loss(sr5, gt5, m[:, 0], ks, kw, 1.0, 1e3, 1e3)
mx = np.abs(ref["grad"]).max()
g = grad.cpu().numpy().astype(np.float64)
print(f"C5 windows N={n}: fp64 oracle {t1 - t0:.1f}s; l1 rel {abs(float(loss[0]) - ref['l1']) / ref['l1']:.2e} kl {float(loss[1]):.4e} rel {abs(float(loss[1]) - ref['kl']) / ref['kl']:.2e} (fp32 oracle: {abs(r32['kl'] - ref['kl']) / ref['kl']:.2e})")
print(f"   grad gpu-vs-fp64 {np.abs(g - ref['grad']).max() / mx:.2e}; fp32 oracle {np.abs(r32['grad'] - ref['grad']).max() / mx:.2e}; dSSG {np.abs(step.ssg_sr[:n].cpu().numpy() - ref['s_sr']).max():.2e}")
gref, nflip = ref_grad_with_gpu_signs(sr5, m[:, 0], ks, kw, 1.0, ref, step.ssg_sr[:n].cpu().numpy(), step.ssg_gt[:n].cpu().numpy())
print(f"   with GPU signs at {nflip}: {np.abs(g - gref).max() / mx:.2e}")
kl = kl64(step.ssg_sr[:n], step.ssg_gt[:n])
print(f"   kl vs fp64-of-same-SSGs rel {abs(float(loss[1]) - float(kl)) / float(kl):.2e}")
