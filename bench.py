#!/usr/bin/env python
"""bench.py -- SSG-loss edge-pixels/sec (fwd+bwd) on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload = BASELINE.json configs[1]: a batch of 16 synthetic 3x256x256 crops PER GPU
(weak scaling; image i of rank r uses seed 100 + 16 r + i), Laplacian edge mask (~7.5 %),
k_s = 25, k_w = 9, sigma = 1.0, generalization, L1 + KL (weights 1e3).  One step = the whole
loss step through the C ABI (ssg_loss_fwd_bwd): edge list from the fp32 mask, SSG(sr), SSG(gt)
materialised once each, both criteria and d(l1+kl)/d sr.  Inputs are resident in HBM.
Unit of work: one edge pixel through that step.

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline      for the dominant kernel: algorithmic HBM bytes of the step (5,500 B per edge
                pixel at this config, SURVEY 8d: 8 k_s^2 + (12C+4) HW/N) x edge pixels per
                launch / that kernel's mean launch time measured with HIP events on the launch
                stream; `step` repeats it over the whole step's GPU time; `valu` prices the
                same time against the fp32 vector peak (the path is VALU-bound).
  cpu_baseline  the C/OpenMP oracle ("port") on the host cores over a bounded sample of the
                same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

KS, KW, SIGMA, EPS = 25, 9, 1.0, 1e-10
W_L1 = W_KL = 1e3
BATCH, C, H, W = 16, 3, 256, 256
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # fp32 vector peak


def alg_bytes_per_edge_px(n_edges, batch):
    """read SR+GT (2*C*H*W*4) + fp32 mask (H*W*4) + write both SSGs
    (2*N*k_s^2*4) + write dL/dsr (C*H*W*4), per image; per edge pixel = 8 k_s^2 + (12C+4) HW/N."""
    return 8.0 * KS * KS + (12.0 * C + 4.0) * H * W * batch / max(n_edges, 1)


def alg_flops_per_edge_px():
    """3*C*k_w^2*k_s^2 per SSG pass, ~4 pass-equivalents per loss step."""
    return 4.0 * 3.0 * C * KW * KW * KS * KS


def event_time_ms(fn, iters):
    """Mean duration of fn() on torch's current stream (HIP events)."""
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    en.synchronize()
    return st.elapsed_time(en) / iters


def stage_times(step, sr, gt, mask, n_edges, iters):
    """Per-kernel mean launch durations of one step, each stage launched separately through
    the same C entry points the fused call uses (HIP events on the launch stream)."""
    from ssl_amd import _lib, engine
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    B = sr.shape[0]
    edges = step.edges()
    scratch = torch.empty(L.ssg_edge_scratch_bytes(B, H, W), dtype=torch.uint8, device=sr.device)
    lscratch = torch.empty(L.ssg_loss_scratch_bytes(B, H, W, step.capacity, KS), dtype=torch.uint8, device=sr.device)
    rank = torch.empty((B, H, W), dtype=torch.int32, device=sr.device)
    order = torch.empty(step.capacity, dtype=torch.int32, device=sr.device)
    plan = torch.empty(L.ssg_forward_plan_bytes(B, H, W, step.capacity) // 4, dtype=torch.int32, device=sr.device)
    p = engine._ptr

    def f_edges():
        _lib.check(L.ssg_edge_list(p(mask), 0, 1, B, H, W, 0, 20.0, KS, p(edges), step.capacity, p(step.counts),
                                   p(rank), p(order), p(plan), p(scratch), st))

    def f_fwd():
        _lib.check(L.ssg_map_forward(p(sr), p(gt), B, C, H, W, p(edges), p(order), p(rank), p(plan), p(step.counts),
                                     n_edges, KS, KW, SIGMA, EPS, 1, p(step.ssg_sr), p(step.ssg_gt), st))

    def f_bwd():
        _lib.check(L.ssg_loss_backward(p(sr), B, C, H, W, p(edges), p(order), p(rank), p(plan), p(step.counts), n_edges, KS, KW, SIGMA, 1,
                                       p(step.ssg_sr), p(step.ssg_gt), W_L1, W_KL, None, p(step.loss), p(step.grad),
                                       p(lscratch), st))

    out = {}
    for name, f in (("edge_list+order+plan", f_edges), (L.ssg_kernel_name(KS, KW, 0).decode(), f_fwd),
                    (L.ssg_kernel_name(KS, KW, 1).decode() + "+finalize", f_bwd)):
        f()
        torch.cuda.synchronize()
        out[name] = event_time_ms(f, iters)
    return out


def cpu_baseline(sr, gt, mask, budget_s=20.0):
    """Oracle (C + OpenMP) timed on the host cores over a bounded sample of the batch
    (about budget_s seconds of wall time: whole images, as many as fit)."""
    from oracle import ssg_oracle as orc
    cores = os.cpu_count() or 1
    orc.ssg_loss(sr[:1, :, :64, :64], gt[:1, :, :64, :64], mask[:1, 0, :64, :64], KS, KW, SIGMA, W_L1, W_KL)  # warm up
    t0 = time.time()
    r = orc.ssg_loss(sr[:1], gt[:1], mask[:1, 0], KS, KW, SIGMA, W_L1, W_KL)     # calibrate on image 0
    per_img = max(time.time() - t0, 1e-3)
    nimg = int(min(sr.shape[0], max(1, round(budget_s / per_img))))
    if nimg > 1:
        t0 = time.time()
        r = orc.ssg_loss(sr[:nimg], gt[:nimg], mask[:nimg, 0], KS, KW, SIGMA, W_L1, W_KL)
        dt = time.time() - t0
    else:
        dt = per_img
    return {"value": r["n_edges"] / dt, "unit": "edge-px/s", "cores": cores, "kind": "port",
            "sample": f"first {nimg} of {sr.shape[0]} images ({r['n_edges']} edge px), fp32 C oracle, OpenMP over "
                      f"edge pixels on all {cores} host cores, {dt:.1f} s",
            "l1": r["l1"], "kl": r["kl"]}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed PMC passes (profiles/pmc_traffic.json,
    produced by tools/prof_pmc.sh + tools/pmc_to_json.py with the guide's unit / gfx950 corrections);
    None if there is no entry for this kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        for k, v in t.get("kernels", {}).items():
            if k.split("<")[0] in kernel_name and ("bwd" in k) == ("bwd" in kernel_name):
                return v.get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step as a recorded HIP graph instead of "
                    "launching its kernels one by one (measured: no difference, the step is not launch-bound)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ    # launched by torch.distributed.run (also with 1 rank)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", rank=rank, world_size=world)   # RCCL on ROCm

    from ssl_amd import engine, synth

    sr_np, gt_np, mask_np = synth.make_batch(BATCH, H, W, seed0=100 + BATCH * rank)
    sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
    n_edges = int(mask_np.sum())
    step = engine.LossStep(BATCH, C, H, W, KS, KW, SIGMA, EPS, True, W_L1, W_KL, device=dev,
                           capacity=n_edges + 1024, graph=args.graph)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(sr, gt, mask)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(sr, gt, mask)
    sync_all()
    elapsed = time.perf_counter() - t0

    tot_edges = torch.tensor([float(n_edges)], device=dev, dtype=torch.float64)
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(tot_edges, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax)
    total_edges = float(tot_edges)

    assert int(step.counts[0]) == n_edges
    loss = step.loss.cpu().numpy()

    if rank == 0:
        value = total_edges * args.steps / elapsed
        b_alg = alg_bytes_per_edge_px(n_edges, BATCH)
        # per-kernel durations on this rank (HIP events on the launch stream)
        stages = stage_times(step, sr, gt, mask, n_edges, max(3, min(args.steps, 10)))
        step_gpu_ms = event_time_ms(lambda: step(sr, gt, mask), max(3, min(args.steps, 10)))
        dom = max((k for k in stages if "ssg_" in k), key=lambda k: stages[k])
        ach = b_alg * n_edges / (stages[dom] * 1e-3) / 1e9
        ach_step = b_alg * n_edges / (step_gpu_ms * 1e-3) / 1e9
        tflops = alg_flops_per_edge_px() * n_edges / (step_gpu_ms * 1e-3) / 1e12
        res = {
            "metric": "SSG-loss edge-pixels/sec (fwd+bwd) 3x256x256 k_s=25 k_w=9",
            "value": value, "unit": "edge-px/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: batch 16 x 3x256x256 per GPU, Laplacian mask, k_s=25 k_w=9 sigma=1.0, "
                                   "L1+KL w=1e3, SSGs materialised",
                       "launch": "HIP graph replay" if args.graph else "per-kernel",
                       "edge_px_per_gpu": n_edges, "mask_density": n_edges / (BATCH * H * W),
                       "input_checksum": synth.checksum(sr_np, gt_np, mask_np), "parallelism": f"images sharded x{world}",
                       "l1": float(loss[0]), "kl": float(loss[1])},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom),
                         "alg_bytes_per_edge_px": b_alg, "kernel_ms": stages,
                         "step": {"gpu_ms": step_gpu_ms, "achieved": ach_step, "frac": ach_step / HBM_PEAK_GBS},
                         "valu": {"achieved": tflops, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": tflops / FP32_PEAK_TFLOPS,
                                  "alg_flops_per_edge_px": alg_flops_per_edge_px(),
                                  "note": "binding roofline: fp32 VALU (~330 flop/B), see DESIGN.md"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sr_np, gt_np, mask_np)
            res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
