"""Timings of the f3 degradation-chain kernels and of the composed datapath.Degradation.feed at the training
configuration's shapes (batch 12 x 3 x 400 x 400 GT crops, scale 4, gt_size 256), beside the same chain on stock torch
ops (F.interpolate, torch arithmetic) where one exists.  Run on the GPU box: python tools/degrade_time.py"""
import os, sys, random
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import datapath

dev = torch.device("cuda:0")


def t_ms(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / n


B, S = 12, 400
x = torch.rand(B, 3, S, S, device=dev)
for mode in ("area", "bilinear", "bicubic"):
    for sf in (0.5, 1.3):
        print(f"resize {mode:8s} x{sf}: ssg_resize {t_ms(lambda: datapath.interpolate(x, scale_factor=sf, mode=mode)):.3f} ms   "
              f"torch F.interpolate {t_ms(lambda: F.interpolate(x, scale_factor=sf, mode=mode)):.3f} ms")
sigma, gray = torch.rand(B, device=dev) * 29 + 1, (torch.rand(B, device=dev) < 0.4).float()
fc, fg = torch.randn(B, 3, S, S, device=dev), torch.randn(S, S, device=dev)
print(f"gaussian noise apply: {t_ms(lambda: datapath.add_gaussian_noise(x, sigma, gray, fc, fg)):.3f} ms  "
      f"(torch.randn of the fields: {t_ms(lambda: torch.randn(B, 3, S, S, device=dev)):.3f} ms)")
rate, rg, vals = datapath.poisson_rates(x, True)
dc, dg = torch.poisson(rate), torch.poisson(rg)
print(f"poisson: census+rates {t_ms(lambda: datapath.poisson_rates(x, True)):.3f} ms, apply "
      f"{t_ms(lambda: datapath.add_poisson_noise(x, sigma / 10, gray, vals, dc, dg)):.3f} ms  (torch.poisson: {t_ms(lambda: torch.poisson(rate)):.3f} ms; "
      f"the reference's per-sample torch.unique loop: {t_ms(lambda: [len(torch.unique(x[i])) for i in range(B)], 3):.1f} ms)")
print(f"clamp_round: {t_ms(lambda: datapath.clamp_round(x)):.3f} ms   torch: {t_ms(lambda: torch.clamp((x * 255.0).round(), 0, 255) / 255.):.3f} ms")
opt = dict(degradation_order="two", scale=4, Use_sharpen=None, Sharpen_before_degra=False, resize_prob=[0.1, 0.85, 0.05],
           resize_range=[0.9, 1.1], gaussian_noise_prob=0.5, noise_range=[1, 8], poisson_scale_range=[0.05, 0.5],
           gray_noise_prob=0.2, jpeg_range=[85, 95], second_blur_prob=0.8, resize_prob2=[0.1, 0.85, 0.05],
           resize_range2=[0.9, 1.1], gaussian_noise_prob2=0.5, noise_range2=[0, 4], poisson_scale_range2=[0, 0.3],
           gray_noise_prob2=0.2, jpeg_range2=[87, 95], queue_size=180, datasets=dict(train=dict(gt_size=256)))
k = torch.zeros(B, 9, 9, device=dev); k[:, 3:6, 3:6] = 1 / 9
pulse = torch.zeros(1, 9, 9, device=dev); pulse[0, 4, 4] = 1
data = dict(gt=x, gt_mask=(torch.rand(B, 1, S, S, device=dev) < 0.08).float(), kernel1=k, kernel2=k, sinc_kernel=pulse)
random.seed(0); np.random.seed(0); torch.manual_seed(0)
deg = datapath.Degradation(opt)
print(f"Degradation.feed (options of train_RealESRGANSSL_x4.yml, batch {B} x 3x{S}x{S}): {t_ms(lambda: deg.feed(data), 10):.2f} ms per batch")
