"""C3 / C4-shaped end-to-end generator steps on the GPU (SURVEY section 8d reporting grid: "step time with the SSL
engine vs with SSL disabled"): the drop-in SSGLoss inside a stock-PyTorch training step, exactly where the
reference's per-image loop sits (realesrganssl_model.py:379-430, ddpmssl.py:424-428).  The networks are random-
weight stand-ins of the published shapes (context for the measurement, not product code).  The figures are printed
(`pytest -s`) and written to gpurun_out/r2_c3_c4_step_share.json; the assertions are on correctness (finite
gradients into the network, same loss value as the C-ABI step) and on the loss staying a minor share of the step."""
import importlib.util
import json
import os
import time

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _generator():
    spec = importlib.util.spec_from_file_location("train_step_bench", os.path.join(ROOT, "tools", "train_step_bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Generator


def _time_steps(step, flag, n=4, repeats=3):
    """Best of `repeats` timings of n steps (a fresh box ramps its clocks during the first seconds: one sample of 4
    steps of a 55 ms generator step has swung by 25 % between the two flags)."""
    for _ in range(2):
        step(flag)
    best = float("inf")
    for _ in range(repeats):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step(flag)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


def test_c3_and_c4_shaped_generator_steps_with_and_without_ssl():
    from ssl_amd import SSGLoss, engine, synth
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    out = {}
    # ---- C3: Real-ESRGAN x4 generator step, 4 GT crops of 256x256 per GPU (bs 32 over 8 GPUs), sigma 0.004, w 1e3
    net = _generator()().to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    _, gt_np, mask_np = synth.make_batch(4, 256, 256, seed0=300)
    gt, mask = torch.as_tensor(gt_np, device=dev), torch.as_tensor(mask_np, device=dev)
    lq = F.interpolate(gt, scale_factor=0.25, mode="bicubic", align_corners=False).clamp(0, 1)
    crit = SSGLoss(25, 9, 0.004, True, 1e3, 1e3)
    last = {}

    def step3(with_ssl):
        opt.zero_grad(set_to_none=True)
        sr = net(lq)
        loss = F.l1_loss(sr, gt)
        if with_ssl:
            l1, kl = crit(sr, gt, mask)
            loss = loss + l1 + kl
            last.update(sr=sr.detach(), l1=l1.detach(), kl=kl.detach())
        loss.backward()
        opt.step()

    base, ssl = _time_steps(step3, False), _time_steps(step3, True)
    base, ssl = min(base, _time_steps(step3, False)), min(ssl, _time_steps(step3, True))   # (interleaved twice)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
    ref = engine.LossStep(4, 3, 256, 256, 25, 9, 0.004, 1e-10, True, 1e3, 1e3, device=dev)
    loss_ref, _ = ref(last["sr"].contiguous(), gt, mask)
    assert abs(float(last["l1"]) - float(loss_ref[0])) <= 1e-6 * float(loss_ref[0])
    assert abs(float(last["kl"]) - float(loss_ref[1])) <= 1e-6 * float(loss_ref[1]) + 1e-12
    out["c3"] = dict(step_ms_without_ssl=base, step_ms_with_ssl=ssl, ssl_share=(ssl - base) / ssl,
                     edge_px=int(mask_np.sum()), what="RRDBNet-shaped x4 generator (23 RRDB, fp32, Adam), 4 x 3x256x256 GT")
    assert (ssl - base) / ssl < 0.15
    del net, opt
    # ---- C4: LDM-SR step tail: a decoder producing 2 x 3x512x512 from a 4x64x64 latent (bs 8 over 4 GPUs), pixel L1
    # + SSL with mask_stride 3, eps 1e-20, w 5e2 (configs/StableSRISSLStage1/*.yml:32-41,268-277)
    dec = torch.nn.Sequential(torch.nn.Conv2d(4, 128, 3, padding=1), torch.nn.SiLU(),
                              torch.nn.Upsample(scale_factor=2), torch.nn.Conv2d(128, 128, 3, padding=1), torch.nn.SiLU(),
                              torch.nn.Upsample(scale_factor=2), torch.nn.Conv2d(128, 64, 3, padding=1), torch.nn.SiLU(),
                              torch.nn.Upsample(scale_factor=2), torch.nn.Conv2d(64, 64, 3, padding=1), torch.nn.SiLU(),
                              torch.nn.Conv2d(64, 3, 3, padding=1)).to(dev)
    opt4 = torch.optim.Adam(dec.parameters(), lr=1e-4)
    gt4 = torch.as_tensor(np.stack([synth.natural_like(2000 + i, 512, 512) for i in range(2)]), device=dev)
    m4 = torch.as_tensor(np.stack([synth.laplacian_edge_mask(gt4[i].cpu().numpy()) for i in range(2)])[:, None]
                         .astype(np.float32), device=dev)
    z = torch.randn(2, 4, 64, 64, device=dev)
    crit4 = SSGLoss(25, 9, 0.004, True, 5e2, 5e2, mask_stride=3, eps=1e-20)

    def step4(with_ssl):
        opt4.zero_grad(set_to_none=True)
        img = dec(z)
        loss = 0.1 * F.l1_loss(img, gt4)
        if with_ssl:
            l1, kl = crit4(img, gt4, m4)
            loss = loss + l1 + kl
        loss.backward()
        opt4.step()

    base4, ssl4 = _time_steps(step4, False), _time_steps(step4, True)
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in dec.parameters())
    out["c4"] = dict(step_ms_without_ssl=base4, step_ms_with_ssl=ssl4, ssl_ms=ssl4 - base4,
                     edge_px=int(crit4.last_counts[0]),
                     what="stand-in decoder tail (4x64x64 latent -> 2 x 3x512x512), pixel L1 + SSL stride 3, eps 1e-20")
    assert ssl4 - base4 < 5.0   # ms: the loss itself is ~0.6 ms at this size
    print(json.dumps(out, indent=1))
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "r2_c3_c4_step_share.json"), "w") as f:
            json.dump(out, f, indent=1)
    except OSError:
        pass
