#!/usr/bin/env python
"""C3-shaped context measurement (reporting grid): one generator training step of a
Real-ESRGAN-sized x4 network on stock PyTorch-ROCm with and without the SSG loss from this engine, to show what
share of a step the loss costs.  The network is a stand-in with RRDBNet's published shape (23 residual-in-
residual dense blocks, 64 features, growth 32, two nearest x2 upsamplings; cf. basicsr/archs/rrdbnet_arch.py:88-
121) and random weights: it is context for the measurement, not part of the product.

    python tools/train_step_bench.py [--batch 4] [--steps 10]
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ssl_amd import SSGLoss, synth  # noqa: E402


class DenseBlock(nn.Module):
    def __init__(self, nf, gc):
        super().__init__()
        self.convs = nn.ModuleList([nn.Conv2d(nf + i * gc, gc if i < 4 else nf, 3, 1, 1) for i in range(5)])

    def forward(self, x):
        feats = [x]
        for i, conv in enumerate(self.convs):
            y = conv(torch.cat(feats, 1))
            if i < 4:
                feats.append(F.leaky_relu(y, 0.2))
        return x + 0.2 * y


class Generator(nn.Module):
    def __init__(self, nf=64, gc=32, blocks=23):
        super().__init__()
        self.first = nn.Conv2d(3, nf, 3, 1, 1)
        self.body = nn.ModuleList([nn.ModuleList([DenseBlock(nf, gc) for _ in range(3)]) for _ in range(blocks)])
        self.body_out = nn.Conv2d(nf, nf, 3, 1, 1)
        self.up1, self.up2 = nn.Conv2d(nf, nf, 3, 1, 1), nn.Conv2d(nf, nf, 3, 1, 1)
        self.hr, self.last = nn.Conv2d(nf, nf, 3, 1, 1), nn.Conv2d(nf, 3, 3, 1, 1)

    def forward(self, x):
        f = self.first(x)
        t = f
        for rrdb in self.body:
            u = t
            for blk in rrdb:
                u = blk(u)
            t = t + 0.2 * u
        f = f + self.body_out(t)
        f = F.leaky_relu(self.up1(F.interpolate(f, scale_factor=2, mode="nearest")), 0.2)
        f = F.leaky_relu(self.up2(F.interpolate(f, scale_factor=2, mode="nearest")), 0.2)
        return self.last(F.leaky_relu(self.hr(f), 0.2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)      # C3: bs 32 over 8 GPUs
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = Generator().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    _, gt_np, _ = synth.make_batch(args.batch, 256, 256, seed0=300)
    gt = torch.as_tensor(gt_np, device=dev)
    lq = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    crit = SSGLoss(kernel_size_search=25, kernel_size_window=9, sigma=0.004, generalization=True,
                   loss_weight_l1=1e3, loss_weight_kl=1e3)

    def step(with_ssl):
        opt.zero_grad(set_to_none=True)
        out = net(lq)
        loss = F.l1_loss(out, gt)
        if with_ssl:
            l1, kl = crit(out, gt, None)          # edge mask of GT generated on the device
            loss = loss + l1 + kl
        loss.backward()
        opt.step()

    res = {}
    for with_ssl in (False, True, False, True):
        for _ in range(3):
            step(with_ssl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(with_ssl)
        torch.cuda.synchronize()
        res.setdefault(with_ssl, []).append((time.perf_counter() - t0) / args.steps * 1e3)
    base, ssl = min(res[False]), min(res[True])
    n = int(crit.last_counts[0])
    print(f"generator step, batch {args.batch} x 3x256x256 GT (x4, 23 RRDB, fp32, Adam): {base:.1f} ms without SSL, "
          f"{ssl:.1f} ms with the SSG loss (N = {n} edge px, k_s=25, k_w=9, sigma=0.004, L1+KL): "
          f"+{ssl - base:.2f} ms = {100 * (ssl - base) / ssl:.1f} % of the step")


if __name__ == "__main__":
    main()
