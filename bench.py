#!/usr/bin/env python
"""bench.py -- SSG-loss edge-pixels/sec (fwd+bwd) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c5] [--scaling weak|strong]

N > 1: either launched by `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env), or plain `python bench.py --gpus N`, which spawns the
N ranks itself (one process per GPU, 127.0.0.1 rendezvous, RCCL through torch.distributed's "nccl" backend).

Workload (config c2, default) = BASELINE.json configs[1]: a batch of 16 synthetic 3x256x256 crops PER GPU (weak
scaling; image i of rank r uses seed 100 + 16 r + i) -- or the 16 images split N-ways (--scaling strong, SURVEY
8e) -- Laplacian edge mask (~7.5 %), k_s = 25, k_w = 9, sigma = 1.0, generalization, L1 + KL (weights 1e3).
One step = the whole loss step through the C ABI (ssg_loss_fwd_bwd): edge list from the fp32 mask, SSG(sr),
SSG(gt) materialised once each, both criteria and d(l1+kl)/d sr.  Inputs are resident in HBM.  Unit of work: one
edge pixel through that step.  Config c5 = BASELINE configs[4]: 1x3x512x512, dense mask,
k_s = 49, k_w = 13 (one image per GPU).

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline      for the dominant kernel: algorithmic HBM bytes of the step (8 k_s^2 + (12C+4) HW/N per
                edge pixel) x edge pixels per launch / that kernel's mean duration, measured here with HIP events
                on the launch stream while the step's other launches are masked out (ssg_set_profile_mask of the profiling build
                libssg_hip_prof.so -- the product library has no such switch);
                `kernel_ms` lists every kernel of the step measured that way (each ALONE on the chip: inside the
                step the direct kernel of a pass runs on a side stream beside the dense one for k_s <= 25, so the
                step is shorter than their sum; profiles/*_kernel_stats.csv is taken with --no-overlap for the
                same reason), `step` repeats the figure over the whole step's GPU time, `valu` prices the same
                time against the fp32 vector peak.
  extra         (N = 1, default config only) the other lines a reader wants next to the headline, measured in the same
                process: C5 materialised, C2 fused (no SSG output), C5 fused -- each with ITS OWN algorithmic bytes
                (B_alg = 8 k_s^2 + (12C+4)HW/N, fused B_alg' = (12C+4)HW/N; the two are never mixed).
  module        the same step through the drop-in nn.Module (ssl_amd.SSGLoss: autograd forward + backward).
  cpu_baseline  the C/OpenMP oracle ("port") on the host cores over a bounded sample of the same workload
                (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # BASELINE configs[0]: the reference's own CPU-runnable case (uniform-noise pair, fixed 5 % mask incl. corners)
    "c1": dict(ks=11, kw=5, sigma=1.0, batch=1, H=64, W=64, dense_mask=False, uniform=True,
               name="C1: 1 x 3x64x64, fixed 5 % mask (209 px), k_s=11 k_w=5 sigma=1.0, L1+KL w=1e3, SSGs materialised"),
    "c2": dict(ks=25, kw=9, sigma=1.0, batch=16, H=256, W=256, dense_mask=False,
               name="C2: batch 16 x 3x256x256 per GPU, Laplacian mask, k_s=25 k_w=9 sigma=1.0, L1+KL w=1e3, SSGs materialised"),
    "c5": dict(ks=49, kw=13, sigma=1.0, batch=1, H=512, W=512, dense_mask=True,
               name="C5: 1 x 3x512x512 per GPU, dense (100 %) mask, k_s=49 k_w=13 sigma=1.0, L1+KL w=1e3, SSGs materialised"),
    # BASELINE configs[3], the loss workload of one GPU of the LDM-SR step (Diffusion-Based-SR/ldm/models/diffusion/
    # ddpmssl.py:438-513, configs/StableSRISSLStage1/*.yml:32-41,268-277): bs 8 over 4 GPUs = 2 crops of 512 x 512,
    # Laplacian mask AND the stride-3 pattern (~0.8 % of the pixels), eps 1e-20, scaling_factor 0.004, weights 5e2
    "c4": dict(ks=25, kw=9, sigma=0.004, batch=2, H=512, W=512, dense_mask=False, eps=1e-20, stride=3, w=5e2, seed0=2000,
               name="C4: 2 x 3x512x512 per GPU, Laplacian mask x stride-3 pattern, k_s=25 k_w=9 sigma=0.004 eps=1e-20, "
                    "L1+KL w=5e2, SSGs materialised"),
    # the sparse regime (round 6, review item 5): 4 x 3x256x256, Bernoulli 1 % mask (2,627 edge pixels, no dense tile) --
    # direct kernels only, a step of ~10 dependent launches; the same generator stream as tools/sweep.py / sparse_step.py
    "b1": dict(ks=25, kw=9, sigma=1.0, batch=4, H=256, W=256, dense_mask=False, bernoulli=0.01,
               name="B1: 4 x 3x256x256 per GPU, Bernoulli 1 % mask, k_s=25 k_w=9 sigma=1.0, L1+KL w=1e3, SSGs materialised"),
}
EPS, C = 1e-10, 3
W_L1 = W_KL = 1e3


def cfg_eps(cfg):
    return cfg.get("eps", EPS)


def cfg_w(cfg):
    return cfg.get("w", W_L1)


def cfg_stride(cfg):
    return cfg.get("stride", 0)


def effective_mask(cfg, mask_np):
    """The mask the loss step really uses: the given mask AND the stride pattern (realesrganssl_model.py:64-72)."""
    from ssl_amd import synth
    if cfg_stride(cfg) <= 1:
        return mask_np
    return mask_np * synth.mask_stride_pattern(cfg["H"], cfg["W"], cfg_stride(cfg))[None, None].astype(mask_np.dtype)


def make_step(cfg, B, dev, capacity, **kw):
    from ssl_amd import engine
    return engine.LossStep(B, C, cfg["H"], cfg["W"], cfg["ks"], cfg["kw"], cfg["sigma"], cfg_eps(cfg), True, cfg_w(cfg),
                           cfg_w(cfg), mask_stride=cfg_stride(cfg), device=dev, capacity=capacity, **kw)
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3   # fp32 vector peak


def alg_bytes_per_edge_px(cfg, n_edges, batch):
    """read SR+GT (2*C*H*W*4) + fp32 mask (H*W*4) + write both SSGs
    (2*N*k_s^2*4) + write dL/dsr (C*H*W*4), per image; per edge pixel = 8 k_s^2 + (12C+4) HW/N."""
    return 8.0 * cfg["ks"] ** 2 + (12.0 * C + 4.0) * cfg["H"] * cfg["W"] * batch / max(n_edges, 1)


def alg_flops_per_edge_px(cfg):
    """3*C*k_w^2*k_s^2 per SSG pass, ~4 pass-equivalents per loss step."""
    return 4.0 * 3.0 * C * cfg["kw"] ** 2 * cfg["ks"] ** 2


def shard_images(total, rank, world):
    """Strong scaling: images [lo, hi) of the `total`-image batch for this rank (contiguous, sizes differ by <= 1)."""
    return total * rank // world, total * (rank + 1) // world


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) and relay rank 0's
    JSON line.  Same environment contract as torch.distributed.run."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), SSG_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit(f"bench ranks exited with {rcs}")


def gpu_clock():
    """Shader clock as rocm-smi reports it right after the timed region (string, or None without the tool)."""
    try:
        out = subprocess.run(["rocm-smi", "-c"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
        for line in out.splitlines():
            if "sclk" in line.lower():
                return line.split(":", 1)[-1].strip() if "GPU[0]" in line else line.strip()
    except (OSError, subprocess.SubprocessError):
        pass
    return None


def event_time_ms(fn, iters):
    """Mean duration of fn() on torch's current stream (HIP events).  A full collection first: right after a section that
    created many objects (the kernel table) the interpreter's cyclic GC ran a full pass in nearly every call of a
    host-bound loop -- the drop-in module's line read 1.6-2.3 ms per call at C4 instead of 0.47 (`profiles/EXPERIMENTS.md`)."""
    import gc
    import torch
    gc.collect()
    st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st.record()
    for _ in range(iters):
        fn()
    en.record()
    en.synchronize()
    return st.elapsed_time(en) / iters


def event_time_in_step_ms(fn, filler, iters):
    """Mean duration of fn() with one whole step (`filler`) in front of every timed call: HIP events around fn only.  Ten
    identical dense kernels back to back run at a lower clock than the same kernel inside the step's mix (DVFS: the
    dense forward read 0.48 ms that way against 0.39 inside the step by rocprofv3, round 6); with the step in between,
    the kernel under test sees the clocks and the cache state it has in the step."""
    import gc
    import torch
    gc.collect()
    pairs = []
    for _ in range(iters):
        filler()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in pairs) / iters


# launches of the step that ssg_set_profile_mask can skip: name -> bit
SKIP_BITS = {"fwd_dense": 25, "fwd_direct": 26, "bwd_dense": 27, "bwd_direct": 28, "grad_rows": 29}


def stage_times(step, sr, gt, mask, n_edges, iters, cfg=None):
    """Mean duration of every kernel of one step: the three C entry points the fused call uses are launched
    separately, and inside the forward / backward entry points all launches but one are masked out
    (HIP events on the launch stream; results are not used)."""
    import torch
    from ssl_amd import _lib, engine
    # geometry and kernel sizes come from the LossStep itself (`cfg` is kept for callers that pass it)
    _, _, H, W = step.shape
    KS, KW, SIGMA, EPS_, _, WL1, WKL, STRIDE, THR = step.cfg
    # the PROFILING build of the library (libssg_hip_prof.so, -DSSG_PROFILE): the only one that can mask launches out;
    # same kernels, same launch code -- the timed step of main() runs on the product library
    L = _lib.lib_prof()
    st = torch.cuda.current_stream().cuda_stream
    B = sr.shape[0]
    edges = step.edges()
    scratch = torch.empty(L.ssg_edge_scratch_bytes(B, H, W), dtype=torch.uint8, device=sr.device)
    lscratch = torch.empty(L.ssg_loss_scratch_bytes(B, H, W, step.capacity, KS), dtype=torch.uint8, device=sr.device)
    rank = torch.empty((B, H, W), dtype=torch.int32, device=sr.device)
    order = torch.empty(step.capacity, dtype=torch.int32, device=sr.device)
    plan = torch.empty(L.ssg_forward_plan_bytes(B, H, W, step.capacity) // 4, dtype=torch.int32, device=sr.device)
    rsc = torch.empty(2 * n_edges, dtype=torch.float64, device=sr.device)   # deferred normalisation, as in the fused call
    p = engine._ptr

    # (the edge list as the fused step builds it: sizes with shared-term kernels take every job order from the plan,
    #  the others have no plan -- three launches either way since the banded builder of round 4)
    dense_sizes = (KS, KW, C) in ((25, 9, 3), (49, 13, 3))

    def o_arg():
        return None if dense_sizes else p(order)

    def p_arg():
        return p(plan) if dense_sizes else None

    def f_edges():
        _lib.check(L.ssg_edge_list(p(mask), 0, 1, B, H, W, STRIDE, THR, KS, p(edges), step.capacity, p(step.counts),
                                   p(rank), o_arg(), p_arg(), p(scratch), st))

    def f_fwd():
        _lib.check(L.ssg_map_forward(p(sr), p(gt), B, C, H, W, p(edges), o_arg(), p(rank), p_arg(), p(step.counts),
                                     n_edges, KS, KW, SIGMA, EPS_, 1, p(step.ssg_sr), p(step.ssg_gt), p(rsc), st))

    def f_bwd():
        _lib.check(L.ssg_loss_backward(p(sr), B, C, H, W, p(edges), o_arg(), p(rank), p_arg(), p(step.counts), n_edges,
                                       KS, KW, SIGMA, 1, p(step.ssg_sr), p(step.ssg_gt), WL1, WKL, None, p(step.loss),
                                       p(step.grad), p(lscratch), None, p(rsc), 0, st))

    def only(fn, keep, group):
        mask_bits = sum(1 << SKIP_BITS[k] for k in group if k != keep)

        def masked():
            prev = L.ssg_set_profile_mask(mask_bits)
            try:
                fn()
            finally:
                L.ssg_set_profile_mask(prev)

        def whole_step():          # (un-masked: the three entry points one after the other = the step's kernels)
            f_edges()
            f_fwd()
            f_bwd()

        masked()
        torch.cuda.synchronize()
        return event_time_in_step_ms(masked, whole_step, iters)

    # every launch on the caller's stream while the kernels are timed one at a time (ssg_set_overlap(0)): with the side
    # stream the separate entry points fork and join around their one unmasked launch (2 x ~12 us of event round trips
    # inside the timed interval -- round 6: the dense forward read 0.47 ms here against 0.39 by rocprofv3)
    prev_overlap = L.ssg_set_overlap(0)
    try:
        return _stage_times_body(L, f_edges, f_fwd, f_bwd, only, iters, KS, KW, prev_overlap)
    finally:
        L.ssg_set_overlap(prev_overlap)


def _stage_times_body(L, f_edges, f_fwd, f_bwd, only, iters, KS, KW, prev_overlap):
    import torch
    f_edges()
    torch.cuda.synchronize()
    out = {"edge_list+plan (as in the step)": event_time_ms(f_edges, iters)}
    f_fwd()
    f_bwd()
    torch.cuda.synchronize()
    fwd_group, bwd_group = ("fwd_dense", "fwd_direct"), ("grad_rows", "bwd_dense", "bwd_direct")
    # ((25,9): both chunk classes of the 8 x 32 tiles in one launch since round 6 -- ssg_{fwd,bwd}_dense_classes<...>)
    dense = ("_classes" if KS == 25 else "") + f"<{KS},{KW},3>"
    out[f"ssg_fwd_dense{dense}"] = only(f_fwd, "fwd_dense", fwd_group)
    out[f"ssg_fwd_tiled<{KS},{KW}> merged+single (2 launches)"] = only(f_fwd, "fwd_direct", fwd_group)
    # (the masked forward runs above left the row scales of the dense-tile rows cleared: a complete forward again, so
    #  that the row passes below see the rows of both classes -- the dense-class pass skips rows with a zero scale)
    f_fwd()
    torch.cuda.synchronize()
    out[f"ssg_grad_rows<{KS},{KW}> (2 passes)+finalize"] = only(f_bwd, "grad_rows", bwd_group)
    out[f"ssg_bwd_dense{dense}"] = only(f_bwd, "bwd_dense", bwd_group)
    out[f"ssg_bwd_tiled<{KS},{KW}>"] = only(f_bwd, "bwd_direct", bwd_group)
    # the masked runs of the backward still launch the 6 us finalize kernel; it is part of the grad_rows line only
    fin = 0.006
    for k in (f"ssg_bwd_dense{dense}", f"ssg_bwd_tiled<{KS},{KW}>"):
        out[k] = max(out[k] - fin, 0.0)
    L.ssg_set_overlap(prev_overlap)   # (the two entry points as they run: dense and direct kernels on two streams)
    f_fwd()                           # (untimed: the first forked call creates the profiling library's side stream)
    f_bwd()
    torch.cuda.synchronize()
    out["forward (all launches)"] = event_time_ms(f_fwd, iters)
    out["backward (all launches)"] = event_time_ms(f_bwd, iters)
    return out


def module_time_ms(cfg, sr, gt, mask, n_edges, iters):
    """The step through the drop-in module: SSGLoss forward (edge list, SSGs, criteria, gradient) + autograd backward."""
    import torch
    from ssl_amd import SSGLoss
    crit = SSGLoss(cfg["ks"], cfg["kw"], cfg["sigma"], True, cfg_w(cfg), cfg_w(cfg), mask_stride=cfg_stride(cfg),
                   eps=cfg_eps(cfg), capacity=n_edges + 1024)
    x = sr.clone().requires_grad_(True)

    def one():
        x.grad = None
        a, b = crit(x, gt, mask)
        (a + b).backward()

    one()
    one()          # (the module reads the edge count back in its first two calls, SSGLoss sync_checks: not timed)
    one()          # (and the first call without that read-back sets up its pinned count buffer)
    torch.cuda.synchronize()
    return event_time_ms(one, iters)


def cpu_baseline(cfg, sr, gt, mask, budget_s=20.0):
    """Oracle (C + OpenMP) timed on the host cores over a bounded sample of the batch
    (about budget_s seconds of wall time: whole images, as many as fit; C5: a strip of edge pixels)."""
    from oracle import ssg_oracle as orc
    KS, KW, SIGMA = cfg["ks"], cfg["kw"], cfg["sigma"]
    W_L1 = W_KL = cfg_w(cfg)
    mask = effective_mask(cfg, mask)
    cores = os.cpu_count() or 1
    orc.ssg_loss(sr[:1, :, :64, :64], gt[:1, :, :64, :64], mask[:1, 0, :64, :64], KS, KW, SIGMA, W_L1, W_KL)  # warm up
    if cfg["dense_mask"]:
        m = mask[:1, 0].copy()
        m[:, 64:] = 0                                   # 64 rows of edge pixels of the dense image
        t0 = time.time()
        r = orc.ssg_loss(sr[:1], gt[:1], m, KS, KW, SIGMA, W_L1, W_KL)
        dt = time.time() - t0
        what = f"{r['n_edges']} of {mask[0, 0].size} edge px (rows 0-63 of the image)"
    else:
        t0 = time.time()
        r = orc.ssg_loss(sr[:1], gt[:1], mask[:1, 0], KS, KW, SIGMA, W_L1, W_KL, eps=cfg_eps(cfg))     # calibrate on image 0
        per_img = max(time.time() - t0, 1e-3)
        nimg = int(min(sr.shape[0], max(1, round(budget_s / per_img))))
        dt = per_img
        if nimg > 1:
            t0 = time.time()
            r = orc.ssg_loss(sr[:nimg], gt[:nimg], mask[:nimg, 0], KS, KW, SIGMA, W_L1, W_KL, eps=cfg_eps(cfg))
            dt = time.time() - t0
        what = f"first {nimg} of {sr.shape[0]} images ({r['n_edges']} edge px)"
    return {"value": r["n_edges"] / dt, "unit": "edge-px/s", "cores": cores, "kind": "port",
            "sample": f"{what}, fp32 C oracle, OpenMP over edge pixels on all {cores} host cores, {dt:.1f} s",
            "l1": r["l1"], "kl": r["kl"]}


def reference_cpu_record(config_key):
    """The reference's OWN ssl_pytorch (loss_util.py:182-229) + L1 + KL + backward timed by import in the build container
    (SURVEY 8(d) 'CPU baseline beside it' (1); tests/golden/make_golden.py timegrid -> tests/golden/
    reference_cpu_timing.json, a committed record: the reference's Python never travels to the GPU box).  Quoted beside
    the live `cpu_baseline` (the C/OpenMP port on this box's cores); None without a record for the configuration."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "reference_cpu_timing.json")) as f:
            t = json.load(f)
    except (OSError, ValueError):
        return None
    if config_key == "c5" and "c5_chunks" in t:
        c = t["c5_chunks"]
        return {"value": c["extrapolated_edge_px_per_s"], "unit": "edge-px/s", "cores": t.get("cores"), "extrapolated": True,
                "sample": "mask chunks of %s edge px of one 3x512x512 image, t = a + b N scaled to N = 262,144 (%.0f s per step)"
                          % ("/".join(str(k["n_edges"]) for k in c["chunks"]), c["extrapolated_full_step_s"]),
                "where": "build container (no GPU), torch %s CPU, %s threads" % (t.get("torch"), t.get("torch_threads"))}
    r = t.get(config_key)
    if not r or "edge_px_per_s" not in r:
        return None
    return {"value": r["edge_px_per_s"], "unit": "edge-px/s", "cores": t.get("cores"), "extrapolated": False,
            "sample": "%d edge px%s, median of 3 after warm-up: %.3f s" % (
                r["n_edges"], " (%d images one after the other)" % r["images"] if "images" in r else "",
                r.get("median_pass_s", r.get("median_after_warmup_s", 0.0))),
            "where": "build container (no GPU), torch %s CPU, %s threads" % (t.get("torch"), t.get("torch_threads"))}


def pmc_traffic(kernel_name):
    """HBM bytes per launch of `kernel_name` from the committed PMC passes (profiles/pmc_traffic.json, produced by
    tools/r*_final.sh with the guide's unit / gfx950 corrections); None if there is no entry for this kernel."""
    e = pmc_entry(kernel_name)
    return e.get("hbm_bytes_per_launch") if e else None


def pmc_step_bytes(config_key):
    """HBM bytes one step of a configuration moves: the per-launch figures of its kernels in profiles/pmc_traffic.json
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes with the guide's corrections) summed; None without a record."""
    if not config_key:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ks = [k for k in json.load(f).get("kernels", {}).values() if k.get("config") == config_key]
        # (a kernel that runs more than once per step -- ssg_grad_rows: one pass per chain -- counts that often:
        # `launches_per_step` = dispatches / steps of the kernel-trace next to the PMC pass, tools/pmc_to_json.py)
        tot = sum(k.get("hbm_bytes_per_launch", 0.0) * k.get("launches_per_step", 1.0) for k in ks)
        return tot or None
    except (OSError, ValueError):
        return None


VALU_CYCLES_PER_INST = 2.0   # MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues over 2 cycles (v_pk_*: 4)
N_SIMD, N_CU, CLOCK_HZ = 1024, 256, 2.4e9


def pmc_entry(kernel_name):
    """The committed PMC record of a kernel (profiles/pmc_traffic.json) or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        key = kernel_name.replace(" ", "")
        # (round 6: the (25,9) dense kernels run both chunk classes in one launch, ssg_{fwd,bwd}_dense_classes<...>)
        keys = [key] + ([key.replace("_dense<", "_dense_classes<")] if "_dense<" in key else []) + \
               ([key.replace("_dense_classes<", "_dense<")] if "_dense_classes<" in key else [])
        for key in keys:
            for k, v in t.get("kernels", {}).items():
                if k == key or k.startswith(key.rstrip(">") + ","):    # ssg_bwd_dense<25,9,3> ~ ssg_bwd_dense<25,9,3,8,2,8>
                    return v
    except (OSError, ValueError):
        pass
    return None


def binding_resource(kernel_name, kernel_ms):
    """Which resource the dominant kernel keeps busiest, from its committed counters priced against the duration
    measured here: HBM (PMC bytes / 8 TB/s), VALU issue (SQ_INSTS_VALU x 2 cycles on 1024 SIMDs: a LOWER bound, packed
    instructions hold the pipe 4 cycles), LDS (SQ_LDS_IDX_ACTIVE summed over the CUs / 256)."""
    e = pmc_entry(kernel_name)
    if not e or kernel_ms <= 0:
        return "hbm", None
    t = kernel_ms * 1e-3
    fr = {"hbm": e.get("hbm_bytes_per_launch", 0.0) / t / (HBM_PEAK_GBS * 1e9),
          "valu": e.get("valu_insts_per_launch", 0.0) * VALU_CYCLES_PER_INST / N_SIMD / CLOCK_HZ / t,
          "lds": e.get("lds_active_cycles_per_launch", 0.0) / N_CU / CLOCK_HZ / t}
    bound = max(fr, key=fr.get)
    # the same instruction count at the rate the MI355X measures for a plain fp32 instruction at 2 waves per SIMD
    # (profiles/r1_microbench2_valu.txt: 2.8 cycles; packed ones ~5): still a lower bound where instructions are packed
    fr["valu_measured_rate"] = fr["valu"] * 2.8 / VALU_CYCLES_PER_INST
    return bound, {"busy_fraction_of_kernel_time": fr,
                   "how": "profiles/pmc_traffic.json counters of this kernel / the duration measured here; hbm = PMC "
                          "bytes at 8 TB/s, valu = SQ_INSTS_VALU x 2 cycles / 1024 SIMDs at 2.4 GHz (lower bound: "
                          "v_pk_* take 4), lds = SQ_LDS_IDX_ACTIVE / 256 CUs at 2.4 GHz"}


def pmc_issue(config_key, step_gpu_ms):
    """Issued VALU wave-instructions and LDS-active cycles of one step from the committed SQ passes (profiles/
    pmc_traffic.json: SQ_INSTS_VALU / SQ_LDS_IDX_ACTIVE per launch, summed over the step's kernels), priced against
    the step's GPU time measured here: a wave64 VALU instruction holds its SIMD-32 for 2 cycles (packed fp32: 4, so
    the VALU figure is a lower bound), 1024 SIMDs at 2.4 GHz."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            ks = [k for k in json.load(f).get("kernels", {}).values() if k.get("config") == config_key]
        insts = sum(k.get("valu_insts_per_launch", 0.0) * k.get("launches_per_step", 1.0) for k in ks)
        lds = sum(k.get("lds_active_cycles_per_launch", 0.0) * k.get("launches_per_step", 1.0) for k in ks)
        hbm = sum(k.get("hbm_bytes_per_launch", 0.0) * k.get("launches_per_step", 1.0) for k in ks)
        if not insts:
            return None
        valu_ms = insts * VALU_CYCLES_PER_INST / N_SIMD / CLOCK_HZ * 1e3
        lds_ms = lds / N_CU / CLOCK_HZ * 1e3
        hbm_ms = hbm / (HBM_PEAK_GBS * 1e9) * 1e3
        return {"valu_wave_insts_per_step": insts, "valu_issue_ms_lower_bound": valu_ms,
                "valu_issue_frac_of_step": valu_ms / step_gpu_ms,
                "lds_active_ms_per_cu": lds_ms, "lds_active_frac_of_step": lds_ms / step_gpu_ms,
                "hbm_bytes_per_step": hbm, "hbm_ms_at_peak": hbm_ms, "hbm_frac_of_step": hbm_ms / step_gpu_ms,
                "source": "profiles/pmc_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU / SQ_LDS_IDX_ACTIVE / FETCH_SIZE / "
                          "WRITE_SIZE, one pass each)"}
    except (OSError, ValueError):
        return None


def extra_line(cfg_key, fused, dev, steps, warmup, maskgen=False):
    """One more (config, mode) measured in the same process for the driver's JSON line: wall-clock over `steps` steps
    between two synchronisations, its own algorithmic bytes (materialised B_alg = 8 k_s^2 + (12C+4)HW/N,
    fused B_alg' = (12C+4)HW/N -- never mixed)."""
    import torch
    from ssl_amd import engine
    cfg = CONFIGS[cfg_key]
    sr_np, gt_np, mask_np = make_inputs(cfg, 0, 1, "weak")
    n = int(effective_mask(cfg, mask_np).sum())
    B = sr_np.shape[0]
    sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
    step = make_step(cfg, B, dev, n + 1024, materialise=not fused)
    def timed(m):
        for _ in range(warmup):
            step(sr, gt, m)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(sr, gt, m)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    same_harness = first_round = None
    if maskgen:
        # mask_kind 2: the reference's offline Laplacian mask generated from GT inside the edge-list kernels.  The same
        # step object alternates between mask=None and the fp32 mask (this harness, this point of the process): the
        # difference of the two modes is what the on-device mask costs, whatever the box's clocks do meanwhile
        timed(mask)                      # (discarded: the clocks settle during the first block of a new workload)
        runs = [timed(m) for m in (None, mask, None, mask)]
        ms = 0.5 * (runs[0] + runs[2])
        same_harness = {"mask_none_ms": [runs[0], runs[2]], "fp32_mask_ms": [runs[1], runs[3]],
                        "delta_ms": ms - 0.5 * (runs[1] + runs[3])}
        mask = None
    else:
        first_round = timed(mask)        # (a rehearsal, like the headline's pre-warm: a workload new to the process runs its
        ms = timed(mask)                 #  first block 1-3 % slow; reported as `first_round_ms`)
        if cfg_key == "c1":
            # the same step object on the general six-launch path (ssg_set_tiny_step(0)) and back, alternating: what the
            # two-launch path of small (11,5) calls (ssg_tiny.hip) buys, whatever the box's clocks do meanwhile
            runs = []
            for on in (False, True, False, True):
                prev = engine.set_tiny_step(on)
                try:
                    runs.append(timed(mask))
                finally:
                    engine.set_tiny_step(prev)
            same_harness = {"general_path_ms": [runs[0], runs[2]], "two_launch_path_ms": [runs[1], runs[3]]}
    assert int(step.counts[0]) == n
    loss = step.loss.cpu().numpy()
    b_alg = alg_bytes_per_edge_px(cfg, n, B) - (8.0 * cfg["ks"] ** 2 if fused else 0.0)
    if maskgen:
        b_alg -= 4.0 * cfg["H"] * cfg["W"] * B / max(n, 1)     # (the fp32 mask read is 0 when generated on the GPU)
    ach = b_alg * n / (ms * 1e-3) / 1e9
    del step, sr, gt, mask
    torch.cuda.empty_cache()
    # what the step really moves (scratch rows included): the committed PMC passes of this configuration
    moved = pmc_step_bytes({"c2": "c2", "c5": "c5f" if fused else "c5"}.get(cfg_key) if not (fused and cfg_key == "c2") else None)
    return {"workload": (cfg["name"].replace("SSGs materialised", "fused step: no SSG output") if fused else cfg["name"]) +
                        (" [mask=None: Laplacian edge mask of GT generated on the device, generate_mask.py:22-31]" if maskgen else ""),
            "ms_per_step": ms, "value": n / (ms * 1e-3), "unit": "edge-px/s", "steps": steps, "warmup": warmup,
            "edge_px": n, "l1": float(loss[0]), "kl": float(loss[1]),
            **({"same_harness": same_harness} if same_harness else {}),
            **({"first_round_ms": first_round} if first_round is not None else {}),
            "roofline": {"step": {"alg_bytes_per_edge_px": b_alg, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": ach / HBM_PEAK_GBS, "traffic": moved,
                                  "traffic_GBps": None if not moved else moved / (ms * 1e-3) / 1e9,
                                  "traffic_frac": None if not moved else moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}}


def ref_api_lines(cfg, sr, gt, mask, n_edges, headline_ms):
    """Row R1: the reference's UNCHANGED caller loop (ssl_amd/reference_loop.py = realesrganssl_model.py:379-430) on the
    headline's inputs -- per image: mask slice, `mask.sum() == 0` (host sync), two clones + `similarity_map` for SR, the
    same for GT; then torch.cat, L1Loss, KLDistanceLoss, `.backward()` -- against ssl_amd's drop-in modules, for every
    `ssl_mode` ('cuda' is the reference YAML's default, train_RealESRGANSSL_x4.yml:115), with deferred handles
    (ssl_amd/losses/lazy.py, the default) and with eager tensors (SSG_LAZY=0).  Wall clock between two
    synchronisations, forward + backward.  `caller_floor` is the same loop around a `similarity_map` that computes
    nothing: what the loop's own clones, synchronisations, cat and autograd bookkeeping cost on this box."""
    import torch
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
    from ssl_amd.reference_loop import gan_selfsim_block
    cri1, cri2 = L1Loss(loss_weight=cfg_w(cfg)), KLDistanceLoss(loss_weight=cfg_w(cfg))
    x = sr.clone().requires_grad_(True)
    P = cfg["ks"] ** 2

    class nothing:            # a similarity_map that only keeps autograd's structure: (1, 1, k_s^2) view of the image
        def __init__(self, img, mask=None, **kw):
            self.s = img.reshape(1, 1, -1)[:, :, :P]

        def getitem(self):
            return self.s

    def loop(smap, mode):
        x.grad = None
        setting = dict(ssl_mode=mode, kernel_size_search=cfg["ks"], generalization=True,
                       kernel_size_window=cfg["kw"], sigma=cfg["sigma"])
        out = x * 1.0
        l1, kl = gan_selfsim_block(smap, cri1, cri2, out, gt, mask, setting)
        (l1 + kl).backward()
        return l1, kl

    def timed(smap, mode, iters, warm):
        """median of three blocks of `iters` loops (the loop is host-heavy: single blocks vary by 10-20 % between boxes)"""
        for _ in range(warm):
            loop(smap, mode)
        blocks = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                l1, kl = loop(smap, mode)
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / iters * 1e3)
        return sorted(blocks)[1], float(l1.detach()), float(kl.detach())

    out = {"what": "unchanged per-image caller loop (similarity_map x2 per image, torch.cat, L1Loss, KLDistanceLoss, "
                   "backward) on the headline's 16 images; ms per loop, forward + backward, wall clock",
           "headline_step_ms": headline_ms}
    floor, _, _ = timed(nothing, "cuda", 10, 3)
    out["caller_floor_ms"] = floor
    for lz in (True, False):
        prev = set_lazy(lz)
        try:
            for mode in ("cuda", "pytorch", "hip"):
                ms, l1, kl = timed(similarity_map, mode, 8 if lz else 2, 2 if lz else 1)
                out[f"{mode}{'' if lz else '_eager'}"] = {
                    "ms": ms, "value": n_edges / (ms * 1e-3), "unit": "edge-px/s", "x_headline": ms / headline_ms,
                    "ms_above_caller_floor": ms - floor, "l1": l1, "kl": kl}
        finally:
            set_lazy(prev)
        torch.cuda.empty_cache()
    return out


def ref_api_dm_line(dev):
    """The diffusion fork's unchanged `issl` loop (ddpmssl.py:438-513, ssl_amd/reference_loop.dm_issl) at C4: 2 crops of
    512 x 512, its constructor arguments (`simself_strategy` areaarea_mask_nonlocalavg_cuda_v1 = eps 1e-20, kernel_size
    25, kernel_size_center 9, scaling_factor 0.004, softmax), mask_stride 3, weights 5e2 -- against
    ssl_amd.losses.dm_loss_util.similarity_map and ssl_amd's criterion modules, deferred and eager."""
    import torch
    from ssl_amd.losses import KLDistanceLoss, L1Loss, set_lazy
    from ssl_amd.losses.dm_loss_util import similarity_map
    from ssl_amd.reference_loop import dm_issl, stride_pattern
    cfg = CONFIGS["c4"]
    sr_np, gt_np, mask_np = make_inputs(cfg, 0, 1, "weak")
    n = int(effective_mask(cfg, mask_np).sum())
    sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
    sslopt = dict(mask_stride=cfg_stride(cfg), simself_strategy="areaarea_mask_nonlocalavg_cuda_v1", kernel_size=cfg["ks"],
                  scaling_factor=cfg["sigma"], softmax_sr=True, softmax_gt=True, kernel_size_center=cfg["kw"])
    pat = stride_pattern(cfg["H"], cfg_stride(cfg), dev)
    cri1, cri2 = L1Loss(loss_weight=cfg_w(cfg)), KLDistanceLoss(loss_weight=cfg_w(cfg))
    x = sr.clone().requires_grad_(True)

    def loop():
        x.grad = None
        l1, kl = dm_issl(similarity_map, cri1, cri2, x * 1.0, gt, mask, sslopt, pat)
        (l1 + kl).backward()
        return l1, kl

    out = {"what": "unchanged DM-fork issl loop at C4 (2 x 3x512x512, stride 3, eps 1e-20), forward + backward, wall clock",
           "edge_px": n}
    for lz in (True, False):
        prev = set_lazy(lz)
        try:
            for _ in range(2):
                loop()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iters = 10 if lz else 5
            for _ in range(iters):
                l1, kl = loop()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            out["deferred" if lz else "eager"] = {"ms": ms, "value": n / (ms * 1e-3), "unit": "edge-px/s",
                                                  "l1": float(l1.detach()), "kl": float(kl.detach())}
        finally:
            set_lazy(prev)
    torch.cuda.empty_cache()
    return out


def operator_line(cfg, sr, mask):
    """INTEGRATION Level 0 / 0a: the reference's native operator on ONE image of the headline batch --
    compute_similarity(image (C,H,W), mask (H,W), psize, ksize) -> raw distances (N, psize, psize) and its backward
    (similaritywrapper.py:25-69 on the C ABI of similarity.h:2-23).  Wall clock incl. the wrapper's reflect pad and
    torch.nonzero (a host synchronisation, as in the reference)."""
    import torch
    from ssl_amd import compute_similarity
    img = sr[0].clone().requires_grad_(True)
    m = mask[0, 0]
    n = int(m.sum())
    cot = torch.rand(n, cfg["ks"], cfg["ks"], device=sr.device)

    def fwd():
        return compute_similarity(image=img, mask=m, psize=cfg["ks"], ksize=cfg["kw"])

    def both():
        img.grad = None
        fwd().backward(cot)

    res = {}
    for name, fn in (("fwd", fwd), ("fwd_bwd", both)):
        for _ in range(3):
            fn()
        blocks = []
        for _ in range(3):                      # median of three blocks (host-heavy: pad, nonzero, autograd)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / 10 * 1e3)
        res[name + "_ms"] = sorted(blocks)[1]
    res.update(what="compute_similarity forward / forward + backward on one 3x256x256 image of the batch (wall clock)",
               edge_px=n, value=n / (res["fwd_bwd_ms"] * 1e-3), unit="edge-px/s")
    return res


def step_share_lines(dev):
    """SURVEY 8d's reporting grid for BASELINE configs[2] / [3]: a training step with the SSL engine vs with SSL disabled
    (stand-in generator / decoder of the published shapes on stock PyTorch-ROCm, tools/train_step_bench.py -- context for
    the measurement, not product code; call sites realesrganssl_model.py:379-430, ddpmssl.py:438-513)."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("train_step_bench", os.path.join(ROOT, "tools", "train_step_bench.py"))
    tsb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tsb)
    out = {"c3_step_share": tsb.c3_step_share(dev), "c4_step_share": tsb.c4_step_share(dev)}
    torch.cuda.empty_cache()
    return out


def make_inputs(cfg, rank, world, scaling):
    """Synthetic batch of this rank (numpy): weak = `batch` images per rank, strong = the rank's share of ONE batch."""
    import numpy as np
    from ssl_amd import synth
    B, H, W = cfg["batch"], cfg["H"], cfg["W"]
    if cfg.get("uniform"):
        return synth.uniform_case(H, W)
    if cfg["dense_mask"]:
        gt = np.stack([synth.natural_like(300 + rank * B + i, H, W) for i in range(B)])
        sr = np.stack([synth.degrade(gt[i], 7 + rank * B + i) for i in range(B)])
        return sr, gt, np.ones((B, 1, H, W), np.float32)
    seed0 = cfg.get("seed0", 100)
    if cfg.get("bernoulli"):
        sr, gt, _ = synth.make_batch(B, H, W, seed0=seed0 + B * rank)
        rng = np.random.default_rng(rank)
        mask = None
        for d in (0.01, 0.04):          # (tools/sweep.py's stream: the 1 % mask is its first draw)
            mm = (rng.random((B, 1, H, W)) < d).astype(np.float32)
            if mask is None and d == cfg["bernoulli"]:
                mask = mm
        return sr, gt, mask
    if scaling == "strong":
        lo, hi = shard_images(B, rank, world)
        sr, gt, mask = synth.make_batch(B, H, W, seed0=seed0)
        return sr[lo:hi], gt[lo:hi], mask[lo:hi]
    return synth.make_batch(B, H, W, seed0=seed0 + B * rank)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 16 images per GPU; strong: the 16 images split over the GPUs (SURVEY 8e)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the timed pre-warm before the W warm-up steps")
    ap.add_argument("--no-module", action="store_true", help="skip the SSGLoss (nn.Module) timing")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` block (C5 and the fused steps)")
    ap.add_argument("--no-kernel-table", action="store_true",
                    help="skip the per-kernel table (it launches the kernels one at a time through the separate entry "
                         "points): a rocprofv3 run of the fused step then shows that step's own kernels only")
    ap.add_argument("--no-overlap", action="store_true",
                    help="ssg_set_overlap(0): every launch on the caller's stream -- for rocprofv3 --kernel-trace runs, whose "
                         "per-kernel durations otherwise include the time a kernel shares the chip with its side-stream twin")
    ap.add_argument("--no-ssg-output", action="store_true",
                    help="the fused step of the C ABI (ssg_sr = ssg_gt = NULL): a SEPARATE metric with SURVEY 8d's "
                         "B_alg' = (12C+4)HW/N -- not comparable with the default line")
    ap.add_argument("--graph", action="store_true", help="replay the step as a recorded HIP graph instead of "
                    "launching its kernels one by one (measured: no difference, the step is not launch-bound)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise the launcher / sharding / "
                    "reduction plumbing on CPU with the gloo backend (used by the CPU test-suite)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        return spawn_ranks(args)               # self-launch: one process per GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import numpy as np
    import torch
    cfg = CONFIGS[args.config]
    use_dist = world > 1 or "RANK" in os.environ    # launched by torch.distributed.run / self-spawned (also 1 rank)
    if args.dry_run:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if use_dist:
        import torch.distributed as dist
        if args.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:   # "nccl" IS RCCL on ROCm; one process per GPU, the rank's device stated explicitly
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from ssl_amd import synth
    if args.no_overlap and not args.dry_run:
        from ssl_amd import engine
        engine.set_overlap(False)
    sr_np, gt_np, mask_np = make_inputs(cfg, rank, world, args.scaling)
    B = sr_np.shape[0]
    n_edges = int(effective_mask(cfg, mask_np).sum())

    def sync_all():
        if not args.dry_run:
            torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            if not args.dry_run:
                torch.cuda.synchronize()

    if args.dry_run:
        step = None

        def run_step():
            return None
    else:
        sr, gt, mask = (torch.as_tensor(a, device=dev) for a in (sr_np, gt_np, mask_np))
        step = make_step(cfg, max(B, 1), dev, n_edges + 1024, graph=args.graph,
                         materialise=not args.no_ssg_output) if B else None

        def run_step():
            if step is not None:
                step(sr, gt, mask)

    # Pre-warm BY TIME before the counted warm-up (disclosure, not tuning: a fresh box ramps its shader clock over the
    # first tenths of a second of work, and `--warmup 10` is 13 ms of GPU time at C2).  The pre-warm REHEARSES the timed
    # region -- W warm-up steps, synchronise, K steps, synchronise -- until two consecutive rehearsals agree to 0.5 % (and
    # 0.3 s have passed) or 1 s has passed; the W warm-up steps and the K timed steps that follow are exactly the ones
    # asked for.  (Rehearsing the same shape matters: after a pre-warm made of differently shaped blocks the first timed
    # block came out 1.5-2 % slower than the two behind it, tools/r5_blocks*.py.)  Reported: config.prewarm_steps /
    # prewarm_ms / prewarm_block_ms_per_step.
    def timed_round():
        for _ in range(args.warmup):
            run_step()
        sync_all()
        t_0 = time.perf_counter()
        for _ in range(args.steps):
            run_step()
        sync_all()
        return time.perf_counter() - t_0

    prewarm_steps, prewarm_t0, last_blk, prewarm_blocks = 0, time.perf_counter(), None, []
    if not args.dry_run and not args.no_prewarm and step is not None:
        torch.cuda.synchronize()
        rounds = 0
        # (several ranks: a fixed number of rehearsals -- every round ends in a barrier, so the ranks must agree on the count)
        while (rounds < 3) if world > 1 else (time.perf_counter() - prewarm_t0 < 1.0):
            blk = timed_round()
            rounds += 1
            prewarm_blocks.append(blk / max(args.steps, 1) * 1e3)
            prewarm_steps += args.steps + args.warmup
            if world == 1 and last_blk is not None and abs(blk - last_blk) <= 0.005 * last_blk and \
                    time.perf_counter() - prewarm_t0 >= 0.3:
                break
            last_blk = blk
    prewarm_ms = (time.perf_counter() - prewarm_t0) * 1e3 if prewarm_steps else 0.0
    elapsed = timed_round()

    # two more blocks of the same K steps, for information only (`config.ms_per_step_blocks`): the first entry is the
    # timed region above; a clock still ramping after the warm-up shows up as a first block slower than the others
    blocks = [elapsed / max(args.steps, 1) * 1e3]
    if not args.dry_run and world == 1:
        for _ in range(2):
            sync_all()
            tb = time.perf_counter()
            for _ in range(args.steps):
                run_step()
            sync_all()
            blocks.append((time.perf_counter() - tb) / max(args.steps, 1) * 1e3)

    tot_edges = torch.tensor([float(n_edges)], device=dev, dtype=torch.float64)
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    tmin = tmax.clone()
    ranks_seen = 1
    if use_dist:
        dist.all_reduce(tot_edges, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        ranks_seen = dist.get_world_size()
    rank_ms_min, rank_ms_max = float(tmin) / max(args.steps, 1) * 1e3, float(tmax) / max(args.steps, 1) * 1e3
    elapsed = float(tmax)
    total_edges = float(tot_edges)

    if rank == 0:
        value = total_edges * args.steps / elapsed
        res = {
            "metric": {"c1": "SSG-loss edge-pixels/sec (fwd+bwd) 3x64x64 k_s=11 k_w=5",
                       "c2": "SSG-loss edge-pixels/sec (fwd+bwd) 3x256x256 k_s=25 k_w=9",
                       "c4": "SSG-loss edge-pixels/sec (fwd+bwd) 3x512x512 k_s=25 k_w=9 mask_stride=3 eps=1e-20",
                       "c5": "SSG-loss edge-pixels/sec (fwd+bwd) 3x512x512 k_s=49 k_w=13 dense mask"}[args.config] +
                      (" [fused step, no SSG output: B_alg' = (12C+4)HW/N]" if args.no_ssg_output else ""),
            "value": value, "unit": "edge-px/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (cfg["name"].replace("SSGs materialised", "fused step: no SSG output")
                                    if args.no_ssg_output else cfg["name"]) +
                                   (" (16 images split over the GPUs)" if args.scaling == "strong" else ""),
                       "launch": ("HIP graph replay" if args.graph else "per-kernel") + (", no side stream" if args.no_overlap else ""), "ranks_seen": ranks_seen,
                       "rank_ms_per_step_min": rank_ms_min, "rank_ms_per_step_max": rank_ms_max,
                       "edge_px_rank0": n_edges, "edge_px_total": total_edges, "images_rank0": B,
                       "mask_density": n_edges / max(B * cfg["H"] * cfg["W"], 1),
                       "input_checksum": synth.checksum(sr_np, gt_np, mask_np),
                       "gradient_accumulation": "fixed-point integer atomics (bit-reproducible, the shipped default)",
                       "prewarm_steps": prewarm_steps, "prewarm_ms": prewarm_ms, "prewarm_block_ms_per_step": prewarm_blocks,
                       "ms_per_step_blocks": blocks,
                       "ms_per_step_block2": blocks[1] if len(blocks) > 1 else None,
                       "ms_per_step_block3": blocks[2] if len(blocks) > 2 else None,
                       "sclk_after_timed_region": gpu_clock(),
                       "parallelism": f"images sharded x{world}, no data-path collective"},
        }
        if args.dry_run:
            res["dry_run"] = True
            res["value"] = None
        elif step is not None:
            assert int(step.counts[0]) == n_edges
            loss = step.loss.cpu().numpy()
            res["config"]["l1"], res["config"]["kl"] = float(loss[0]), float(loss[1])
            b_alg = alg_bytes_per_edge_px(cfg, n_edges, B) - (8.0 * cfg["ks"] ** 2 if args.no_ssg_output else 0.0)
            it = max(3, min(args.steps, 10))
            # The per-kernel table launches the kernels one at a time through the separate entry points.  It is skipped
            # on request (rocprofv3 runs of one step's own kernels) and for k_s = 49, where the timed step runs on
            # tile-major rows (ssg_fwd_strip, ssg_rows_tm[_mat], ssg_bwd_dense<..., TM>) while the separate entry points
            # run the row-major kernels: that step's per-kernel figures are the committed rocprofv3 stats.
            key = ("c5f" if args.no_ssg_output else "c5") if args.config == "c5" else (None if args.no_ssg_output else args.config)
            if args.no_kernel_table or cfg["ks"] == 49 or args.config == "c1":
                # (short steps: enough of them that the two event records do not show -- C1 read 0.036 ms over 10 steps)
                step_gpu_ms = event_time_ms(lambda: step(sr, gt, mask), it if cfg["ks"] == 49 else max(it, 200))
                ach_step = b_alg * n_edges / (step_gpu_ms * 1e-3) / 1e9
                moved = pmc_step_bytes(key)
                res["roofline"] = {"bound": "hbm", "alg_bytes_per_edge_px": b_alg, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "achieved": ach_step, "frac": ach_step / HBM_PEAK_GBS, "traffic": moved,
                                   "kernel": "whole step (per-kernel durations: profiles/r6_bench_%s_kernel_stats.csv)" % (key or "c2"),
                                   "step": {"gpu_ms": step_gpu_ms, "achieved": ach_step, "frac": ach_step / HBM_PEAK_GBS,
                                            "traffic": moved,
                                            "traffic_GBps": None if not moved else moved / (step_gpu_ms * 1e-3) / 1e9,
                                            "traffic_frac": None if not moved else moved / (step_gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
            else:
                # (per-kernel times: through the separate entry points, which need SSG tensors of their own)
                step_k = step if not args.no_ssg_output else make_step(cfg, B, dev, n_edges + 1024)
                if step_k is not step:
                    step_k(sr, gt, mask)
                stages = stage_times(step_k, sr, gt, mask, n_edges, it, cfg)
                del step_k
                # (the kernel table leaves the chip idle between its single-kernel timings: the clocks have to ramp again, or the
                #  first steps behind it read 15 % long -- rounds 5-6 reported 1.35-1.42 ms here beside a 1.16-1.25 ms headline)
                for _ in range(max(3 * it, 150)):
                    step(sr, gt, mask)
                torch.cuda.synchronize()
                step_gpu_ms = event_time_ms(lambda: step(sr, gt, mask), max(it, 50))
                dom = max((k for k in stages if k.startswith("ssg_") and "launches" not in k and "+" not in k),
                          key=lambda k: stages[k])
                ach = b_alg * n_edges / (stages[dom] * 1e-3) / 1e9
                ach_step = b_alg * n_edges / (step_gpu_ms * 1e-3) / 1e9
                tflops = alg_flops_per_edge_px(cfg) * n_edges / (step_gpu_ms * 1e-3) / 1e12
                bound, evidence = binding_resource(dom, stages[dom])
                res["roofline"] = {
                    # `achieved / peak / frac` are the HBM figures BASELINE.json's metric asks for (algorithmic bytes of the
                    # step over the dominant kernel's duration); `bound` names the resource that kernel actually keeps
                    # busiest by its counters -- "hbm", "valu" or "lds" (no MFMA on this path) -- see `bound_evidence`
                    "bound": bound, "bound_evidence": evidence,
                    "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic(dom), "alg_bytes_per_edge_px": b_alg,
                    "kernel_ms": stages,
                    "step": {"gpu_ms": step_gpu_ms, "achieved": ach_step, "frac": ach_step / HBM_PEAK_GBS,
                             "traffic": pmc_step_bytes(None if args.no_ssg_output and args.config == "c2" else
                                                       ("c5f" if args.no_ssg_output else args.config)),
                             "traffic_note": "HBM bytes the whole step moves by the committed PMC passes (scratch rows, "
                                             "G rows, atomics included); / gpu_ms = the step's real HBM rate"},
                    "valu": {"reference_equivalent_tflops": tflops, "fp32_vector_peak_tflops": FP32_PEAK_TFLOPS,
                             "alg_flops_per_edge_px": alg_flops_per_edge_px(cfg),
                             "note": "SURVEY's DIRECT flop count over the step's GPU time; the dense-tile kernels do "
                                     "~10x fewer real flops per (pixel, offset), so this is throughput in "
                                     "reference-equivalent flops and may exceed the peak -- NOT a utilisation "
                                     "(`issued` prices the instructions actually issued)",
                             "issued": pmc_issue(args.config, step_gpu_ms)}}
            # SURVEY 8(d)'s own definition as a scalar next to `frac` (which divides by the DOMINANT KERNEL's duration):
            # algorithmic bytes of the step x steps per second of the TIMED region / 8 TB/s
            res["roofline"]["step_frac"] = b_alg * value / 1e9 / HBM_PEAK_GBS
            if not args.no_module and not args.no_ssg_output:
                mm = module_time_ms(cfg, sr, gt, mask, n_edges, it if cfg["dense_mask"] else max(it, 30))
                res["module"] = {"what": "ssl_amd.SSGLoss forward + autograd backward (drop-in path)",
                                 "ms_per_step": mm, "value": n_edges / (mm * 1e-3), "unit": "edge-px/s"}
            if world == 1 and not args.no_extra and args.config == "c2" and not args.no_ssg_output:
                # the other configuration / mode lines, driver-visible (about 2 s of GPU time in all)
                res["extra"] = {"c1": extra_line("c1", False, dev, 200, 50),
                                "c5": extra_line("c5", False, dev, 10, 3),
                                "c2_fused": extra_line("c2", True, dev, 30, 5),
                                "c5_fused": extra_line("c5", True, dev, 10, 3),
                                "c4": extra_line("c4", False, dev, 50, 10),
                                "b1": extra_line("b1", False, dev, 100, 20),
                                "c2_maskgen": extra_line("c2", False, dev, 30, 5, maskgen=True),
                                "ref_api": ref_api_lines(cfg, sr, gt, mask, n_edges, elapsed / args.steps * 1e3),
                                "ref_api_dm": ref_api_dm_line(dev),
                                "operator": operator_line(cfg, sr, mask),
                                **step_share_lines(dev)}
            if world == 1 and not args.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(cfg, sr_np, gt_np, mask_np)
                res["cpu_baseline"]["gpu_over_cpu"] = value / res["cpu_baseline"]["value"]
                ref = reference_cpu_record(args.config)
                if ref:
                    res["cpu_baseline"]["reference_ssl_pytorch"] = dict(ref, gpu_over_reference=value / ref["value"])
        print(json.dumps(res))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
