#!/usr/bin/env python
"""C3-shaped context measurement (reporting grid): one generator training step of a
Real-ESRGAN-sized x4 network on stock PyTorch-ROCm with and without the SSG loss from this engine, to show what
share of a step the loss costs.  The network is a stand-in with RRDBNet's published shape (23 residual-in-
residual dense blocks, 64 features, growth 32, two nearest x2 upsamplings; cf. basicsr/archs/rrdbnet_arch.py:88-
121) and random weights: it is context for the measurement, not part of the product.

    python tools/train_step_bench.py [--batch 4] [--steps 4]

bench.py reports both figures as `extra.c3_step_share` / `extra.c4_step_share`; tests/test_gpu_training_steps.py asserts
on the same two functions.
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


def _paired_steps(step, pairs=20, timer=None):
    """Interleaved measurement (A B A B ...): `pairs` times one step WITHOUT and one step WITH the SSL terms, each
    bracketed by a synchronise.  Two back-to-back blocks of a 50 ms step differ by more than the 0.5 ms under test on a
    box that is still settling its clocks (round 5's driver run recorded a negative share that way); paired steps see the
    same clock.  Returns the medians, the median of the paired differences and its spread (quartiles)."""
    for _ in range(3):
        step(False)
        step(True)
    a, b = [], []
    if timer is not None:
        timer.on = True
    for _ in range(pairs):
        for flag, dst in ((False, a), (True, b)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(flag)
            torch.cuda.synchronize()
            dst.append((time.perf_counter() - t0) * 1e3)
    if timer is not None:
        timer.on = False
    d = sorted(y - x for x, y in zip(a, b))
    med = lambda v: sorted(v)[len(v) // 2]
    return dict(base=med(a), ssl=med(b), diff=med(d), diff_q1=d[len(d) // 4], diff_q3=d[(3 * len(d)) // 4], pairs=pairs)


class _LossTimer:
    """HIP events around the loss module's call inside the step (the module runs the engine's forward AND backward
    kernels in its forward and keeps the gradient: its autograd backward is one scaling): the stream time of the SSL
    kernels of every timed step, whatever the generator's 50 ms do to the wall clock."""

    def __init__(self):
        self.pairs, self.on = [], False

    def __call__(self, crit, *a):
        if not self.on:
            return crit(*a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = crit(*a)
        e1.record()
        self.pairs.append((e0, e1))
        return out

    def median_ms(self):
        v = sorted(a.elapsed_time(b) for a, b in self.pairs)
        return v[len(v) // 2] if v else None


def _share(t, timer=None, **kw):
    """One `extra.c*_step_share` record.  `ssl_ms` = the loss module's own stream time inside the step (HIP events around
    its call, median over the timed steps: never negative, independent of the clocks during the generator's part);
    `ssl_ms_paired_diff` = the median of the A B A B paired wall-clock differences with its quartiles (a 0.4 ms effect
    under a +-0.5 ms step-to-step spread: reported, not relied on); `ssl_share` = ssl_ms / median step with SSL."""
    inner = timer.median_ms() if timer is not None else None
    ssl_ms = inner if inner is not None else t["diff"]
    return dict(step_ms_without_ssl=t["base"], step_ms_with_ssl=t["ssl"], ssl_ms=ssl_ms, ssl_share=ssl_ms / t["ssl"],
                ssl_ms_paired_diff=t["diff"], ssl_ms_paired_diff_quartiles=[t["diff_q1"], t["diff_q3"]], pairs=t["pairs"],
                how="ssl_ms: HIP events around the loss module's call in every timed step (median); paired_diff: A B A B "
                    "interleaved single steps, median of the paired wall-clock differences", **kw)


def c3_step_share(dev, batch=4, pairs=20, check=None):
    """BASELINE configs[2] per GPU (bs 32 over 8 GPUs = 4 GT crops of 256 x 256): one generator step of the RRDBNet-shaped
    stand-in (fp32, Adam) with pixel L1 only and with the SSG loss where the reference's per-image loop sits
    (realesrganssl_model.py:379-430: sigma 0.004, weights 1e3).  Timed as interleaved single steps (_paired_steps)."""
    torch.manual_seed(0)
    net = Generator().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    _, gt_np, mask_np = synth.make_batch(batch, 256, 256, seed0=300)
    gt, mask = torch.as_tensor(gt_np, device=dev), torch.as_tensor(mask_np, device=dev)
    lq = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    crit = SSGLoss(25, 9, 0.004, True, 1e3, 1e3)
    last, timer = {}, _LossTimer()

    def step(with_ssl):
        opt.zero_grad(set_to_none=True)
        sr = net(lq)
        loss = F.l1_loss(sr, gt)
        if with_ssl:
            l1, kl = timer(crit, sr, gt, mask)
            loss = loss + l1 + kl
            last.update(sr=sr.detach(), l1=l1.detach(), kl=kl.detach())
        loss.backward()
        opt.step()

    t = _paired_steps(step, pairs, timer)
    if check is not None:
        check(net, last, gt, mask)
    return _share(t, timer, edge_px=int(mask_np.sum()),
                what=f"RRDBNet-shaped x4 generator (23 RRDB, fp32, Adam), {batch} x 3x256x256 GT, SSL (25,9) sigma 0.004 w 1e3")


def c4_step_share(dev, pairs=20, check=None):
    """BASELINE configs[3] per GPU (bs 8 over 4 GPUs = 2 crops of 512 x 512): the tail of the LDM-SR step -- a stand-in
    decoder producing 2 x 3x512x512 from a 4x64x64 latent, 0.1 pixel L1 (ddpmssl.py:424-425) -- with and without SSL on
    the decoded image (mask_stride 3, eps 1e-20, weights 5e2: configs/StableSRISSLStage1/*.yml:32-41,268-277)."""
    torch.manual_seed(0)
    dec = nn.Sequential(nn.Conv2d(4, 128, 3, padding=1), nn.SiLU(),
                        nn.Upsample(scale_factor=2), nn.Conv2d(128, 128, 3, padding=1), nn.SiLU(),
                        nn.Upsample(scale_factor=2), nn.Conv2d(128, 64, 3, padding=1), nn.SiLU(),
                        nn.Upsample(scale_factor=2), nn.Conv2d(64, 64, 3, padding=1), nn.SiLU(),
                        nn.Conv2d(64, 3, 3, padding=1)).to(dev)
    opt = torch.optim.Adam(dec.parameters(), lr=1e-4)
    _, gt_np, m_np = synth.make_batch(2, 512, 512, seed0=2000)
    gt, m = torch.as_tensor(gt_np, device=dev), torch.as_tensor(m_np, device=dev)
    z = torch.randn(2, 4, 64, 64, device=dev)
    crit = SSGLoss(25, 9, 0.004, True, 5e2, 5e2, mask_stride=3, eps=1e-20)
    timer = _LossTimer()

    def step(with_ssl):
        opt.zero_grad(set_to_none=True)
        img = dec(z)
        loss = 0.1 * F.l1_loss(img, gt)
        if with_ssl:
            l1, kl = timer(crit, img, gt, m)
            loss = loss + l1 + kl
        loss.backward()
        opt.step()

    t = _paired_steps(step, pairs, timer)
    if check is not None:
        check(dec)
    return _share(t, timer, edge_px=int(crit.last_counts[0]),
                what="stand-in decoder tail (4x64x64 latent -> 2 x 3x512x512), 0.1 pixel L1, SSL (25,9) stride 3 eps 1e-20 w 5e2")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)      # C3: bs 32 over 8 GPUs
    ap.add_argument("--pairs", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    import json
    print(json.dumps({"c3": c3_step_share(dev, args.batch, args.pairs), "c4": c4_step_share(dev, args.pairs)}, indent=1))


if __name__ == "__main__":
    main()
