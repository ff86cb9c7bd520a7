"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run only in the build container (needs /root/reference, which never travels
to the GPU box):

    python tests/golden/make_golden.py

It imports the reference's own GAN-Based-SR/basicsr/losses/loss_util.py by
path -- with the CUDA-only `basicsr.losses.similarity.similaritywrapper`
module stubbed in sys.modules, because the real one needs loguru + a CUDA
device and calls sys.exit() without them (similaritywrapper.py:11-13) -- runs
`similarity_map(..., ssl_mode='pytorch')` (the F.unfold path,
loss_util.py:182-229) on seeded inputs and stores inputs + outputs as small
.npz files.  L1Loss / KLDistanceLoss live in basic_loss.py, whose import pulls
torchvision (absent here); their forward bodies are the two one-liners
basic_loss.py:16 (F.l1_loss, 'mean') and basic_loss.py:281 (F.kl_div on
clamped logs), restated below with torch.nn.functional exactly as written
there.

Only DATA is stored (inputs, expected outputs); no reference source text.
"""
import importlib.util
import os
import sys
sys.dont_write_bytecode = True   # importing the reference by path must not leave a __pycache__ in /root/reference (read-only by contract)
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference/GAN-Based-SR/basicsr/losses/loss_util.py"

from ssl_amd import synth  # noqa: E402


def load_reference():
    for name in ("basicsr", "basicsr.losses", "basicsr.losses.similarity"):
        sys.modules.setdefault(name, types.ModuleType(name))
    stub = types.ModuleType("basicsr.losses.similarity.similaritywrapper")

    def compute_similarity(*a, **k):  # never reached in ssl_mode='pytorch'
        raise RuntimeError("CUDA op is not available in the build container")

    stub.compute_similarity = compute_similarity
    sys.modules["basicsr.losses.similarity.similaritywrapper"] = stub
    spec = importlib.util.spec_from_file_location("ref_loss_util", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref = load_reference()


def ref_ssg(img, mask, ks, kw, sigma, gen, dtype=torch.float32, requires_grad=False):
    """similarity_map(img (1,C,H,W), mask (1,c1,H,W)).getitem() from the reference."""
    t = torch.as_tensor(img, dtype=dtype).clone().requires_grad_(requires_grad)
    m = torch.as_tensor(mask, dtype=dtype)
    s = ref.similarity_map(img=t, mask=m, ssl_mode="pytorch", kernel_size_search=ks,
                           generalization=gen, kernel_size_window=kw, sigma=sigma).getitem()
    return t, s


def l1_loss(a, b, w):   # basic_loss.py:66 -> :16, reduction='mean'
    return w * F.l1_loss(a, b, reduction="mean")


def kl_loss(a, b, w):   # basic_loss.py:281
    return w * F.kl_div(torch.clamp(a, min=1e-10).log(), torch.clamp(b, min=1e-10), reduction="mean")


def caller_loop(sr, gt, masks, ks, kw, sigma, gen, w1, w2, dtype):
    """realesrganssl_model.py:379-430: per-image loop, skip empty, cat, two criteria."""
    sr_t = torch.as_tensor(sr, dtype=dtype).clone().requires_grad_(True)
    gt_t = torch.as_tensor(gt, dtype=dtype)
    m_t = torch.as_tensor(masks, dtype=dtype)
    ls, lg = [], []
    for i in range(sr_t.shape[0]):
        bm = m_t[i, :].unsqueeze(0)
        if bm.sum() == 0:
            continue
        ls.append(ref.similarity_map(img=sr_t[i, :].unsqueeze(0).clone(), mask=bm.clone(), ssl_mode="pytorch",
                                     kernel_size_search=ks, generalization=gen, kernel_size_window=kw,
                                     sigma=sigma).getitem())
        with torch.no_grad():
            lg.append(ref.similarity_map(img=gt_t[i, :].unsqueeze(0).clone(), mask=bm.clone(), ssl_mode="pytorch",
                                         kernel_size_search=ks, generalization=gen, kernel_size_window=kw,
                                         sigma=sigma).getitem())
    a = torch.cat(ls, dim=1)
    b = torch.cat(lg, dim=1)
    l1 = l1_loss(a, b, w1)
    kl = kl_loss(a, b, w2)
    (l1 + kl).backward()
    return a.detach(), b.detach(), l1.detach(), kl.detach(), sr_t.grad.detach()


def save(name, **kw):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **kw)
    print(f"  wrote {name}.npz  {os.path.getsize(path) / 1024:.0f} KB")


def smooth_cotangent(shape, seed):
    """Fixed smooth-ish random cotangent for dSSG/dimg checks."""
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float64)


def f1_c1():
    """F1 (BASELINE configs[0]): 1x3x64x64 uniform noise, fixed 5 % mask incl. the
    4 corners, k_s=11 k_w=5, sigma in {1.0, 0.05}, generalization in {F,T}."""
    sr, gt, mask = synth.uniform_case()
    ks, kw = 11, 5
    out = dict(sr=sr, gt=gt, mask=mask.astype(np.uint8), ks=ks, kw=kw)
    n = int(mask.sum())
    cot = smooth_cotangent((1, n, ks * ks), 7).astype(np.float32).astype(np.float64)
    out["cot"] = cot.astype(np.float32)
    for sigma in (1.0, 0.05):
        for gen in (False, True):
            tag = f"s{sigma}_g{int(gen)}"
            t, s = ref_ssg(sr, mask, ks, kw, sigma, gen, torch.float64, True)
            (s * torch.as_tensor(cot, dtype=torch.float64)).sum().backward()
            out[f"ssg_{tag}"] = s.detach().numpy()[0].astype(np.float32)
            out[f"dimg_{tag}"] = t.grad.numpy()[0].astype(np.float32)
            if gen and sigma == 1.0:   # one fp32 run of the reference, for the record
                _, s32 = ref_ssg(sr, mask, ks, kw, sigma, gen, torch.float32)
                out[f"ssg_{tag}_f32run"] = s32.detach().numpy()[0]
    # full caller-loop semantics on the same pair (weights 1e3 like the YAML)
    for dt, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
        a, b, l1, kl, g = caller_loop(sr, gt, mask, ks, kw, 1.0, True, 1e3, 1e3, dt)
        out[f"l1_{dn}"] = l1.numpy()
        out[f"kl_{dn}"] = kl.numpy()
        if dn == "f64":
            out["grad"] = g.numpy().astype(np.float32)
    save("f1_c1_64", **out)


def f2_paper(size, sigmas, name, rows=64):
    """F2 (BASELINE configs[1]-shaped, ONE image): natural-like pair, Laplacian
    mask ~8 %, k_s=25 k_w=9, generalization=True, L1+KL weights 1e3."""
    ks, kw = 25, 9
    gt = synth.natural_like(100, size, size)[None]
    sr = synth.degrade(gt[0], 10_100)[None]
    mask = synth.laplacian_edge_mask(gt[0])[None, None]
    n = int(mask.sum())
    # sampled rows: first/last (image corners side) + random
    rng = np.random.default_rng(3)
    sel = np.unique(np.concatenate([[0, 1, n - 2, n - 1], rng.choice(n, rows - 4, replace=False)]))
    out = dict(sr=sr, gt=gt, mask=mask.astype(np.uint8), ks=ks, kw=kw, rows=sel, n_edges=n,
               checksum=synth.checksum(sr, gt, mask.astype(np.float32)))
    print(f"  {name}: N={n} density={n / size / size:.4f}")
    for sigma in sigmas:
        for dt, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
            t0 = time.time()
            a, b, l1, kl, g = caller_loop(sr, gt, mask, ks, kw, sigma, True, 1e3, 1e3, dt)
            out[f"l1_s{sigma}_{dn}"] = l1.numpy()
            out[f"kl_s{sigma}_{dn}"] = kl.numpy()
            gg = g.numpy()
            out[f"grad_absmax_s{sigma}_{dn}"] = np.abs(gg).max()
            out[f"grad_sum_s{sigma}_{dn}"] = gg.astype(np.float64).sum()
            if dn == "f32":
                g32 = gg
            if dn == "f64":
                # how far the reference's OWN fp32 run is from its fp64 run (max-norm, relative to
                # max|grad|): the yardstick for fp32 gradient parity (L1's sign() and the softmax-like
                # normalisation make tiny-sigma=1 gradients ill-conditioned in fp32)
                out[f"grad_ref32_dev_s{sigma}"] = np.abs(g32.astype(np.float64) - gg).max() / np.abs(gg).max()
                out[f"ssg_sr_s{sigma}"] = a.numpy()[0][sel].astype(np.float32)
                out[f"ssg_gt_s{sigma}"] = b.numpy()[0][sel].astype(np.float32)
                out[f"grad_s{sigma}"] = gg.astype(np.float32)
            print(f"    sigma={sigma} {dn}: l1={float(l1):.6g} kl={float(kl):.6g} ({time.time() - t0:.0f}s)")
    save(name, **out)


def f3_masks():
    """F3: 3-channel mask (ssl_pytorch lists every edge pixel 3x, block-tiled:
    realesrganssl_model.py:339-341 feeds (b,3,H,W) masks) and a batch of 2 with
    one EMPTY mask (skipped, realesrganssl_model.py:387-388)."""
    ks, kw, sigma = 11, 5, 0.5
    rng = np.random.default_rng(11)
    img = rng.random((2, 3, 40, 48), dtype=np.float32)
    gt = rng.random((2, 3, 40, 48), dtype=np.float32)
    m1 = (rng.random((40, 48)) < 0.06).astype(np.float32)
    m1[0, 0] = m1[39, 47] = m1[0, 47] = 1
    out = dict(sr=img, gt=gt, ks=ks, kw=kw, sigma=sigma)
    mask3 = np.repeat(m1[None, None], 3, axis=1)
    _, s3 = ref_ssg(img[:1], mask3, ks, kw, sigma, True)
    _, s1 = ref_ssg(img[:1], m1[None, None], ks, kw, sigma, True)
    out["mask1"] = m1.astype(np.uint8)
    out["ssg_mask3_f32"] = s3.detach().numpy()[0]
    out["ssg_mask1_f32"] = s1.detach().numpy()[0]
    masks = np.stack([np.zeros_like(m1), m1])[:, None]  # image 0 empty
    for dt, dn in ((torch.float32, "f32"), (torch.float64, "f64")):
        a, b, l1, kl, g = caller_loop(img, gt, masks, ks, kw, sigma, True, 1e3, 1e3, dt)
        out[f"b2_l1_{dn}"] = l1.numpy()
        out[f"b2_kl_{dn}"] = kl.numpy()
        if dn == "f64":
            out["b2_grad"] = g.numpy().astype(np.float32)
    out["b2_masks"] = masks.astype(np.uint8)
    save("f3_masks", **out)


def f4_stress():
    """F4 (BASELINE configs[4]-shaped): 3x128x128, k_s=49 k_w=13, DENSE mask
    evaluated by the reference in chunks of 256 mask pixels (an SSG row depends
    only on (img, y, x)); 32 sampled rows incl. the image corners."""
    ks, kw, sigma = 49, 13, 1.0
    H = W = 128
    gt = synth.natural_like(300, H, W)[None]
    rng = np.random.default_rng(5)
    pix = np.unique(np.concatenate([[0, W - 1, (H - 1) * W, H * W - 1, 5 * W + 7],
                                    rng.choice(H * W, 27, replace=False)]))
    m = np.zeros(H * W, np.float32)
    m[pix] = 1
    m = m.reshape(1, 1, H, W)
    out = dict(img=gt, ks=ks, kw=kw, sigma=sigma, pix=pix)
    _, s = ref_ssg(gt, m, ks, kw, sigma, True, torch.float64)
    out["ssg"] = s.detach().numpy()[0].astype(np.float32)
    save("f4_stress_ks49", **out)


def f5_stride_f6_eps():
    """F5: mask_stride=3 eye pattern (ddpmssl.py:47-56,445-446) applied before the
    SSG; F6: the Diffusion fork's epsilon 1e-20 (DM loss_util.py:1250) -- obtained
    from the reference's un-normalised output (generalization=False) by the
    reference's own normalisation expression with that epsilon."""
    ks, kw, sigma = 25, 9, 0.004
    H = W = 96
    gt = synth.natural_like(400, H, W)[None]
    mask = synth.laplacian_edge_mask(gt[0])
    s = 3
    eye = torch.eye(s, s, dtype=torch.float32).repeat(int(np.ceil(H / s)), int(np.ceil(W / s)))[:H, :W]
    ms = (eye.numpy() * mask).astype(np.float32)
    out = dict(img=gt, mask=mask.astype(np.uint8), mask_strided=ms.astype(np.uint8), stride=s,
               ks=ks, kw=kw, sigma=sigma)
    n = int(ms.sum())
    sel = np.unique(np.concatenate([[0, n - 1], np.random.default_rng(9).choice(n, 46, replace=False)]))
    out["rows"] = sel
    out["n_edges"] = n
    _, a = ref_ssg(gt, ms[None, None], ks, kw, sigma, True, torch.float64)
    out["ssg_strided"] = a.detach().numpy()[0][sel].astype(np.float32)
    _, e = ref_ssg(gt, ms[None, None], ks, kw, sigma, False, torch.float64)
    q = e.detach()
    q20 = 1 / (torch.sum(q, dim=-1) + 1e-20).unsqueeze(-1) * q
    out["e_strided"] = q.numpy()[0][sel].astype(np.float32)
    out["ssg_eps1e-20"] = q20.numpy()[0][sel].astype(np.float32)
    save("f5_stride_f6_eps", **out)


def f7_mask_pil():
    """Edge-mask fixture: a 64x64 uint8 RGB image, PIL's convert('L') of it, and
    the mask by generate_mask.py:22-31 with cv2.Laplacian(CV_8U) replaced by its
    documented arithmetic evaluated with scipy.ndimage (cv2 is not installed):
    3x3 [[0,1,0],[1,-4,1],[0,1,0]], BORDER_REFLECT_101 == scipy 'mirror',
    saturate to [0,255], > 20."""
    from PIL import Image
    from scipy import ndimage
    g = synth.natural_like(500, 64, 64)
    u8 = np.rint(g * 255).astype(np.uint8).transpose(1, 2, 0).copy()
    L = np.array(Image.fromarray(u8).convert("L"))
    K = np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], np.int32)
    lap = np.clip(ndimage.correlate(L.astype(np.int32), K, mode="mirror"), 0, 255).astype(np.uint8)
    mask = np.zeros(L.shape, dtype="int")
    mask[lap > 20.0] = 1
    save("f7_edge_mask", rgb=u8, gray=L, lap=lap, mask=mask.astype(np.uint8))


def load_reference_dm():
    """The Diffusion-Based-SR fork's loss_util.py, imported by path.  Its only native dependency is
    `compute_similarity` (the CUDA operator, similaritywrapper.py:59-69); here it is backed by the fp64 CPU
    oracle's distance (pinned to the reference by F1-F6; the fork's own F.unfold strategy
    `areaarea_mask_nonlocal` cross-checks it below) wrapped as a torch autograd function, so every epilogue
    that runs is the reference's own code."""
    from oracle import ssg_oracle as orc

    class _Dist(torch.autograd.Function):
        @staticmethod
        def forward(ctx, image, mask, psize, ksize):
            pos = orc.mask_to_pos(mask.numpy())
            img = image.detach().numpy().astype(np.float64)
            ctx.meta = (img, pos, psize, ksize, image.dtype)
            return torch.as_tensor(orc.distance(img, pos, psize, ksize)).to(image.dtype)

        @staticmethod
        def backward(ctx, g):
            img, pos, psize, ksize, dt = ctx.meta
            gi = orc.distance_backward(img, pos, psize, ksize, g.numpy().astype(np.float64))
            return torch.as_tensor(gi).to(dt), None, None, None

    stub = types.ModuleType("basicsr.losses.similarity.similaritywrapper")
    stub.compute_similarity = lambda image, mask, psize=25, ksize=9: _Dist.apply(image, mask, psize, ksize)
    for name in ("basicsr", "basicsr.losses", "basicsr.losses.similarity"):
        sys.modules.setdefault(name, types.ModuleType(name))
    saved = sys.modules.get("basicsr.losses.similarity.similaritywrapper")
    sys.modules["basicsr.losses.similarity.similaritywrapper"] = stub
    spec = importlib.util.spec_from_file_location(
        "ref_dm_loss_util", "/root/reference/Diffusion-Based-SR/basicsr/losses/loss_util.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if saved is not None:
        sys.modules["basicsr.losses.similarity.similaritywrapper"] = saved
    return mod


def f8_dm_strategies():
    """F8: every operator-based `simself_strategy` of the Diffusion fork's similarity_map
    (DM loss_util.py:299-363) + trainable_similarity_map (:1448-1481), fp64, on one 3x48x64 GT/SR pair."""
    dm = load_reference_dm()
    ks, kc, sigma = 11, 5, 0.05
    H, W = 48, 64
    gt = synth.natural_like(800, H, W)[None]
    sr = synth.degrade(gt[0], 801)[None]
    m1 = synth.laplacian_edge_mask(gt[0]).astype(np.float32)
    rng = np.random.default_rng(8)
    m3 = np.stack([m1, m1 * (rng.random((H, W)) < 0.6), np.roll(m1, 3, axis=1)]).astype(np.float32)
    out = dict(gt=gt, sr=sr, mask=m1.astype(np.uint8), mask3=m3.astype(np.uint8), ks=ks, kc=kc, sigma=sigma,
               kc_list=np.array([3, 5, 7]), largest_k=16, dh=16, dw=32, sigma_raw=sigma * 3 * kc * kc)
    G, S = torch.as_tensor(gt, dtype=torch.float64), torch.as_tensor(sr, dtype=torch.float64)
    M1, M3 = torch.as_tensor(m1[None, None], dtype=torch.float64), torch.as_tensor(m3[None], dtype=torch.float64)
    common = dict(kernel_size=ks, softmax=True, dh=16, dw=32)

    def run(strategy, **kw):
        a = dict(common, simself_strategy=strategy, scaling_factor=sigma, kernel_size_center=kc, mask=M1)
        a.update(kw)
        return dm.similarity_map(G, **a)

    raw = sigma * 3 * kc * kc     # the un-averaged strategies divide D itself by sigma
    P = "areaarea_mask_"
    out["nonlocal"] = run(P + "nonlocal", scaling_factor=raw).getitem()[0].numpy()
    out["nonlocal_cuda_v1"] = run(P + "nonlocal_cuda_v1", scaling_factor=raw).getitem()[0].numpy()
    assert np.abs(out["nonlocal"] - out["nonlocal_cuda_v1"]).max() < 1e-12   # unfold twin == operator + epilogue
    out["nonlocal_cuda_v1_patch"] = run(P + "nonlocal_cuda_v1_patch", scaling_factor=raw).getitem()[0].numpy()
    out["nonlocal_cuda_v2"] = run(P + "nonlocal_cuda_v2", scaling_factor=0.5).getitem()[0].numpy()
    for name in ("nonlocalavg_cuda_v1", "nonlocalavg_cuda_v2", "eulardistanceavg_cuda_v1", "nonlocalavg_cuda_v3"):
        out[name] = run(P + name).getitem()[0].numpy()
    out["nonlocalavg_cuda_v1_nosoftmax"] = run(P + "nonlocalavg_cuda_v1", softmax=False).getitem()[0].numpy()
    out["nonlocalavg_cuda_v4"] = run(P + "nonlocalavg_cuda_v4", kernel_size_center=[3, 5, 7]).getitem()[0].numpy()
    out["nonlocalavg_cuda_v1RGB"] = run(P + "nonlocalavg_cuda_v1RGB", mask=M3).getitem()[0].numpy()
    out["nonlocalavg_cuda_v5_sum"] = run(P + "nonlocalavg_cuda_v5", gene_type="sum").getitem()[0].numpy()
    out["nonlocalavg_cuda_v5_softmax_top16"] = run(P + "nonlocalavg_cuda_v5", gene_type="softmax",
                                                   largest_k=16).getitem()[0].numpy()
    mh = dm.similarity_map(G, mask=M1, img_sr=S, simself_strategy=P + "nonlocalavg_cuda_maxh_v1", kernel_size=ks,
                           scaling_factor=sigma, softmax=True, kernel_size_center=kc)
    a, b = mh.getitem_simmutual()
    out["maxh_gt"], out["maxh_sr"] = a[0].numpy(), b[0].numpy()
    # learnable bandwidth: maps + d(loss)/d(sigma) and d(loss)/d(sr) of loss = sum((s_gt - s_sr)^2)
    tm = dm.trainable_similarity_map(scaling_factor=sigma).double()
    Sg = S.clone().requires_grad_(True)
    s0, s1 = tm(G, Sg, M1, kernel_size_search=ks, kernel_size_center=kc, softmax=True)
    loss = (s0 - s1).pow(2).sum()
    loss.backward()
    out["trainable_s_gt"], out["trainable_s_sr"] = s0[0].detach().numpy(), s1[0].detach().numpy()
    out["trainable_loss"] = float(loss.detach())
    out["trainable_dsigma"] = tm.sigma.grad.numpy()
    out["trainable_dsr"] = Sg.grad.numpy()
    # gradient of one epilogue variant through the operator: d sum(s * cot) / d img for v2 (centre removed)
    Gg = G.clone().requires_grad_(True)
    s = dm.similarity_map(Gg, mask=M1, simself_strategy=P + "nonlocalavg_cuda_v2", kernel_size=ks,
                          scaling_factor=sigma, softmax=True, kernel_size_center=kc).getitem()
    cot = torch.as_tensor(smooth_cotangent(tuple(s.shape), 88))
    (s * cot).sum().backward()
    out["v2_cotangent_seed"] = 88
    out["v2_dimg"] = Gg.grad.numpy()
    out["row_step"] = 4      # SSG maps are stored as rows [::4] to keep the fixture small
    for k, v in list(out.items()):
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.size > 64:
            out[k] = (v[::4] if v.ndim == 2 else v).astype(np.float32)
    save("f8_dm_strategies", **out)


def f9_c4_dm_loop():
    """F9 (BASELINE configs[3]-shaped, C4): the Diffusion fork's own caller loop (ddpmssl.py:438-513 `issl`) with
    its configured options (configs/StableSRISSLStage1/*.yml:32-41,268-277): mask_stride 3 (eye pattern built
    as in ddpmssl.py:47-56), simself_strategy areaarea_mask_nonlocalavg_cuda_v1 (eps 1e-20, DM loss_util.py:
    1239-1252), kernel_size 25, kernel_size_center 9, scaling_factor 0.004, softmax True, L1 + KL weights 5e2 --
    on a 2x3x128x128 crop pair (one image's mask emptied by the stride pattern is NOT forced; both are used).
    The loop below follows `issl` statement by statement; `similarity_map` is the fork's class, its
    `compute_similarity` backed by the fp64 oracle distance (see load_reference_dm)."""
    import math
    dm = load_reference_dm()
    ks, kc, sigma, stride, w = 25, 9, 0.004, 3, 5e2
    H = W = 128
    gt = np.stack([synth.natural_like(900 + i, H, W) for i in range(2)])
    sr = np.stack([synth.degrade(gt[i], 950 + i) for i in range(2)])
    mask = np.stack([synth.laplacian_edge_mask(gt[i]) for i in range(2)])[:, None].astype(np.float32)
    eye = torch.eye(stride, stride, dtype=torch.float32).repeat(math.ceil(H / stride), math.ceil(W / stride))
    eye = eye[:H, :W].unsqueeze(0).unsqueeze(0)
    out = dict(sr=sr, gt=gt, mask=mask.astype(np.uint8), ks=ks, kc=kc, sigma=sigma, stride=stride, w=w)
    for dt, dn in ((torch.float64, "f64"),):
        S = torch.as_tensor(sr, dtype=dt).clone().requires_grad_(True)
        G = torch.as_tensor(gt, dtype=dt)
        M = torch.as_tensor(mask, dtype=dt)
        b_sr_list, b_gt_list, n_per = [], [], []
        for i in range(2):
            b_mask_gt = M[i, :].unsqueeze(0)
            b_mask_gt = eye.to(dt) * b_mask_gt
            if b_mask_gt.sum() == 0:
                continue
            kw = dict(simself_strategy="areaarea_mask_nonlocalavg_cuda_v1", dh=64, dw=64, kernel_size=ks,
                      scaling_factor=sigma, temperature=0, crossentropy=False, rearrange_back=True, stride=1,
                      pix_num=1, index=None, kernel_size_center=kc, mean=False, var=False, gene_type="sum",
                      largest_k=0)
            a = dm.similarity_map(img=S[i, :].unsqueeze(0).clone(), mask=b_mask_gt.clone(), softmax=True, **kw).getitem()
            with torch.no_grad():
                b = dm.similarity_map(img=G[i, :].unsqueeze(0).clone(), mask=b_mask_gt.clone(), softmax=True, **kw).getitem()
            b_sr_list.append(a)
            b_gt_list.append(b)
            n_per.append(a.shape[1])
        A = torch.cat(b_sr_list, dim=1)
        Bm = torch.cat(b_gt_list, dim=1)
        l1 = l1_loss(A, Bm, w)
        kl = kl_loss(A, Bm, w)
        (l1 + kl).backward()
        n = A.shape[1]
        sel = np.unique(np.concatenate([[0, n - 1], np.random.default_rng(19).choice(n, 30, replace=False)]))
        out.update(rows=sel, n_edges=n, n_per_image=np.array(n_per), l1=l1.detach().numpy(), kl=kl.detach().numpy(),
                   ssg_sr=A.detach().numpy()[0][sel].astype(np.float32),
                   ssg_gt=Bm.detach().numpy()[0][sel].astype(np.float32),
                   grad=S.grad.numpy().astype(np.float32))
        print(f"  f9: N={n} per image {n_per} l1={float(l1):.6g} kl={float(kl):.6g}")
    save("f9_c4_dm_loop", **out)


def f10_paper_cotangent():
    """F10: d sum(SSG * cot)/d img at the paper's kernel sizes (k_s 25, k_w 9) under a fixed smooth cotangent,
    sigma in {1.0, 0.004}, from the reference's ssl_pytorch in fp64 -- one 3x96x96 natural-like image whose
    Laplacian mask touches the image border (reflect fold)."""
    ks, kw = 25, 9
    H = W = 96
    img = synth.natural_like(1000, H, W)[None]
    mask = synth.laplacian_edge_mask(img[0])
    mask[0, 0] = mask[0, W - 1] = mask[H - 1, 0] = mask[H - 1, W - 1] = mask[0, 40] = mask[50, 0] = 1
    n = int(mask.sum())
    cot = smooth_cotangent((1, n, ks * ks), 23).astype(np.float32)
    out = dict(img=img, mask=mask.astype(np.uint8), ks=ks, kw=kw, cot_seed=23, n_edges=n)   # cot = smooth_cotangent((1,n,ks*ks), 23) as float32
    for sigma in (1.0, 0.004):
        t, s = ref_ssg(img, mask[None, None], ks, kw, sigma, True, torch.float64, True)
        (s * torch.as_tensor(cot, dtype=torch.float64)).sum().backward()
        out[f"dimg_s{sigma}"] = t.grad.numpy()[0].astype(np.float32)
        print(f"  f10: N={n} sigma={sigma} max|dimg|={np.abs(out[f'dimg_s{sigma}']).max():.4g}")
    save("f10_paper_cotangent", **out)


def f11_datapath():
    """F11: the reference's own data-path code on seeded inputs.
    (a) basicsr/data/transforms.py `augment` + `paired_random_crop_img_mask` -- imported by path with `cv2`
        stubbed: its only use there is cv2.flip(src, flipCode, dst) in place, whose semantics (flipCode 1 =
        horizontal, 0 = vertical) are restated with numpy in the stub;
    (b) basicsr/models/realesrganssl_model.py `_dequeue_and_enqueue` -- the method's source is read from the
        reference at generation time, `.cuda()` dropped (no GPU here), and executed on a bare object.
    Stored: inputs, the python `random` / torch seeds, the draws and the outputs."""
    import inspect
    import random
    import re
    cv2 = types.ModuleType("cv2")

    def flip(src, flipCode, dst=None):
        out = src[:, ::-1].copy() if flipCode == 1 else src[::-1].copy()
        if dst is not None:
            dst[...] = out
            return dst
        return out

    cv2.flip = flip
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2
    spec = importlib.util.spec_from_file_location("ref_transforms", "/root/reference/GAN-Based-SR/basicsr/data/transforms.py")
    tr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr)
    if saved is not None:
        sys.modules["cv2"] = saved
    else:
        del sys.modules["cv2"]
    rng = np.random.default_rng(77)
    B, Hs, Ws, scale, gt_size = 4, 48, 48, 4, 32
    gts = rng.random((B, Hs, Ws, 3), dtype=np.float32)
    masks = (rng.random((B, Hs, Ws, 1)) < 0.1).astype(np.float32)
    out = dict(gt_src=gts, mask_src=masks, scale=scale, gt_size=gt_size, seed=1234)
    random.seed(1234)
    aug_gt, aug_mask, status = [], [], []
    for b in range(B):                                   # dataset side: my_realesrgan_image_mask_dataset.py:86
        (g, m), st = tr.augment([gts[b].copy(), masks[b].copy()], True, True, return_status=True)
        aug_gt.append(np.ascontiguousarray(g))
        aug_mask.append(np.ascontiguousarray(m))
        status.append([int(bool(v)) for v in st])
    out["flips"] = np.array(status, np.int32)
    G = torch.as_tensor(np.stack(aug_gt)).permute(0, 3, 1, 2).contiguous()
    M = torch.as_tensor(np.stack(aug_mask)).permute(0, 3, 1, 2).contiguous()
    LQ = torch.as_tensor(rng.random((B, 3, Hs // scale, Ws // scale), dtype=np.float32))
    out["lq_src"] = LQ.numpy()
    state = random.getstate()
    g2, l2, m2 = tr.paired_random_crop_img_mask(img_gts=G, img_lqs=LQ, masks=M, gt_patch_size=gt_size, scale=scale)
    random.setstate(state)
    out["top_left_lq"] = np.array([random.randint(0, Hs // scale - gt_size // scale),
                                   random.randint(0, Ws // scale - gt_size // scale)], np.int32)
    out["gt_out"], out["lq_out"], out["mask_out"] = g2.numpy(), l2.numpy(), m2.numpy()

    # (b) the pair pool
    src = open("/root/reference/GAN-Based-SR/basicsr/models/realesrganssl_model.py").read()
    m = re.search(r"    def _dequeue_and_enqueue\(self\):.*?(?=\n    def |\n    @)", src, re.S)
    body = inspect.cleandoc("\n" + m.group(0)).replace(".cuda()", "")
    ns = {"torch": torch}
    exec(body, ns)
    pool = types.SimpleNamespace(queue_size=8)
    torch.manual_seed(4321)
    b = 2
    steps = 9
    lq_in = rng.random((steps, b, 3, 4, 4), dtype=np.float32)
    gt_in = rng.random((steps, b, 3, 8, 8), dtype=np.float32)
    mk_in = (rng.random((steps, b, 1, 8, 8)) < 0.3).astype(np.float32)
    lq_o, gt_o, mk_o = [], [], []
    for t in range(steps):
        pool.lq, pool.gt, pool.gt_mask = (torch.as_tensor(a[t]) for a in (lq_in, gt_in, mk_in))
        ns["_dequeue_and_enqueue"](pool)
        lq_o.append(pool.lq.numpy().copy())
        gt_o.append(pool.gt.numpy().copy())
        mk_o.append(pool.gt_mask.numpy().copy())     # (b,3,h,w) once the pool is full: GT-channel copies (:339-341)
    out.update(pool_seed=4321, pool_size=8, pool_lq_in=lq_in, pool_gt_in=gt_in, pool_mask_in=mk_in.astype(np.uint8),
               pool_lq_out=np.stack(lq_o), pool_gt_out=np.stack(gt_o),
               pool_mask_out=np.stack([a[:, :1] for a in mk_o]).astype(np.uint8),
               pool_mask_channels_equal=all(bool((a == a[:, :1]).all()) for a in mk_o))
    save("f11_datapath", **out)


def cpu_reference_timing():
    """time the reference ssl_pytorch loss step on this
    container's cores (one 3x256x256 image of the C2 batch, like the reference's
    per-image loop) and record it next to the fixtures."""
    import json
    torch.set_num_threads(os.cpu_count())
    ks, kw, sigma = 25, 9, 1.0
    gt = synth.natural_like(100, 256, 256)[None]
    sr = synth.degrade(gt[0], 10_100)[None]
    mask = synth.laplacian_edge_mask(gt[0])[None, None]
    n = int(mask.sum())
    times = []
    for it in range(4):          # 1 warm-up + 3 timed ("1 warm-up + median of 3")
        t0 = time.time()
        caller_loop(sr, gt, mask, ks, kw, sigma, True, 1e3, 1e3, torch.float32)
        times.append(time.time() - t0)
        print(f"    reference step {it}: {times[-1]:.1f}s")
    med = float(np.median(times[1:]))
    res = dict(what="reference ssl_pytorch fwd(SR)+fwd(GT)+L1+KL+backward, 1x3x256x256, k_s=25 k_w=9 sigma=1.0, fp32",
               n_edges=n, seconds=times, median_after_warmup_s=med, edge_px_per_s=n / med,
               cores=os.cpu_count(), torch_threads=torch.get_num_threads(), torch=torch.__version__)
    with open(os.path.join(HERE, "reference_cpu_timing.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("  ", res)


def cpu_reference_timing_grid():
    """SURVEY section 8(d) 'CPU baseline beside it' (1): the reference's own ssl_pytorch
    (loss_util.py:182-229) + L1 + KL + backward, timed in the build container by import:
      c2  all 16 images of the C2 batch, one after the other (each image runs
          fwd(SR) + fwd(GT) + L1 + KL + backward before the next: the reference's loop keeps all
          16 graphs alive until one backward, >= 51 GB of saved unfold tensors, which the
          container's 62 GB without swap cannot hold; the arithmetic is the same);
      c1  the 64x64 / 209-pixel plumbing case, directly;
      c5  3x512x512, (49,13): mask chunks of 512 and 1024 pixels (a 4096-pixel chunk needs 4 x 20 GB of
          unfold intermediates), fitted t = a + b*N and scaled to N = 262,144: EXTRAPOLATED.
    1 warm-up + median of 3 (c2: 1 warm-up image + every image once per pass, 3 passes)."""
    import json
    torch.set_num_threads(os.cpu_count())
    out = dict(cores=os.cpu_count(), torch_threads=torch.get_num_threads(), torch=torch.__version__,
               what="reference ssl_pytorch fwd(SR)+fwd(GT)+L1+KL+backward by import of loss_util.py, fp32, w=1e3")
    path = os.path.join(HERE, "reference_cpu_timing.json")
    if os.path.exists(path):
        with open(path) as f:
            old = json.load(f)
        out["one_image_round1"] = old.get("one_image_round1", {k: old[k] for k in
                                          ("n_edges", "seconds", "median_after_warmup_s", "edge_px_per_s") if k in old})

    def timed(fn, reps=3, warm=1):
        ts = []
        for it in range(warm + reps):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return ts, float(np.median(ts[warm:]))

    # ---- C1
    sr, gt, mask = synth.uniform_case()
    n1 = int(mask.sum())
    ts, med = timed(lambda: caller_loop(sr, gt, mask, 11, 5, 1.0, True, 1e3, 1e3, torch.float32), reps=5)
    out["c1"] = dict(n_edges=n1, seconds=ts, median_after_warmup_s=med, edge_px_per_s=n1 / med)
    print("   c1", out["c1"], flush=True)

    # ---- C2: 16 images sequentially
    srs, gts, masks = synth.make_batch()
    ns = [int(m.sum()) for m in masks]
    caller_loop(srs[:1], gts[:1], masks[:1], 25, 9, 1.0, True, 1e3, 1e3, torch.float32)      # warm-up
    passes = []
    for p in range(3):
        per = []
        for i in range(len(ns)):
            t0 = time.time()
            caller_loop(srs[i:i + 1], gts[i:i + 1], masks[i:i + 1], 25, 9, 1.0, True, 1e3, 1e3, torch.float32)
            per.append(time.time() - t0)
        passes.append(per)
        print(f"   c2 pass {p}: {sum(per):.1f}s for {sum(ns)} edge px", flush=True)
    tot = [float(sum(p)) for p in passes]
    med = float(np.median(tot))
    out["c2"] = dict(images=len(ns), n_edges=int(sum(ns)), n_edges_per_image=ns, pass_seconds=tot,
                     per_image_seconds=passes, median_pass_s=med, edge_px_per_s=sum(ns) / med,
                     note="images one after the other, backward per image (memory: see docstring)")
    print("   c2", {k: out["c2"][k] for k in ("n_edges", "pass_seconds", "edge_px_per_s")}, flush=True)

    # ---- C5 chunks
    gt5 = synth.natural_like(500, 512, 512)[None]
    sr5 = synth.degrade(gt5[0], 10_500)[None]
    fits = []
    for nchunk in (512, 1024):
        m = np.zeros((1, 1, 512, 512), np.float32)
        m.reshape(-1)[256 * 512 - nchunk // 2: 256 * 512 + nchunk // 2] = 1.0   # rows in the middle of the image
        ts, med = timed(lambda: caller_loop(sr5, gt5, m, 49, 13, 1.0, True, 1e3, 1e3, torch.float32), reps=3)
        fits.append((nchunk, med, ts))
        print(f"   c5 chunk {nchunk}: {ts}", flush=True)
    (na, ta, _), (nb, tb, _) = fits
    b = (tb - ta) / (nb - na)
    a = ta - b * na
    full = a + b * 512 * 512
    out["c5_chunks"] = dict(chunks=[dict(n_edges=n, seconds=ts, median_after_warmup_s=t) for n, t, ts in fits],
                            fit_fixed_s=a, fit_per_edge_px_s=b, n_full=512 * 512,
                            extrapolated_full_step_s=full, extrapolated_edge_px_per_s=512 * 512 / full,
                            extrapolated=True,
                            note="the reference cannot run C5 whole (1.28 TB intermediate): t = a + b*N fitted on two "
                                 "mask chunks of one 3x512x512 image, scaled linearly in N")
    print("   c5", out["c5_chunks"], flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


def f12_usm():
    """F12: the reference's USMSharp (basicsr/utils/img_process_util.py:63-83), imported by path
    and run on the CPU (fp32).  `cv2` is not installed; the module's only cv2 call on this path is
    cv2.getGaussianKernel(radius, sigma) in USMSharp.__init__, stubbed with the documented formula (the same
    restatement the oracle uses, oracle/datapath_oracle.py gaussian_kernel_1d) -- so F12 pins filter2D's padding and
    correlation, the residual / mask / soft-mask / clip / blend chain and the 51 x 51 outer-product kernel, not
    OpenCV's kernel values themselves.  Inputs quantised to k/255 like real GT; the threshold test |residual| * 255
    > 10 has no pixel within 5e-4 of the threshold (the seed is searched for that), so the mask is the same at any precision."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import datapath_oracle as dorc
    from ssl_amd import synth
    cv2 = types.ModuleType("cv2")
    cv2.getGaussianKernel = lambda ksize, sigma: dorc.gaussian_kernel_1d(int(ksize), float(sigma)).reshape(-1, 1)
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2
    spec = importlib.util.spec_from_file_location("ref_img_process_util",
                                                  "/root/reference/GAN-Based-SR/basicsr/utils/img_process_util.py")
    ipu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ipu)
    usm = ipu.USMSharp()                  # radius 50 -> 51, sigma 0: the configuration of realesrganssl_model.py:35
    usm9 = ipu.USMSharp(radius=9, sigma=1.5)
    if saved is not None:
        sys.modules["cv2"] = saved
    else:
        del sys.modules["cv2"]
    def margin_of(im, mod, kw, sg):
        _, res, _ = dorc.usm_sharp(im, radius=mod.radius, sigma=sg, weight=kw.get("weight", 0.5),
                                   threshold=kw.get("threshold", 10), return_parts=True)
        return np.abs(np.abs(res) * 255 - kw.get("threshold", 10)).min()

    cfgs = (("r50", usm, dict(), 0.0), ("r9", usm9, dict(weight=0.8, threshold=4), 1.5))
    for seed in range(1200, 1300):   # first seed whose residuals keep clear of both thresholds (fp32 noise: ~3e-5)
        img = np.stack([synth.natural_like(seed + 100 * i, 72, 88, 0.15, 0.05) for i in range(2)]).astype(np.float32)
        if all(margin_of(img, m, kw, sg) > 5e-4 for _, m, kw, sg in cfgs):
            break
    else:
        raise RuntimeError("no seed with a clear threshold margin")
    x = torch.as_tensor(img)
    out = dict(img=img, kernel51=usm.kernel.numpy()[0], seed=seed)
    for tag, mod, kw, sg in cfgs:
        o32 = mod(x, **kw).numpy()   # (fp32 only: forward()'s mask.float() rules out a double run of the module)
        out["out32_" + tag] = o32
        orc_out = dorc.usm_sharp(img, radius=mod.radius, sigma=sg, weight=kw.get("weight", 0.5),
                                 threshold=kw.get("threshold", 10))
        margin = margin_of(img, mod, kw, sg)
        print("f12", tag, "fp64 oracle vs reference fp32:", np.abs(orc_out - o32).max(), "threshold margin", margin)
    save("f12_usm", **out)


def f13_filter2d():
    """F13: the reference's filter2D (basicsr/utils/img_process_util.py:7-31), imported by path (the
    module imports cv2 at the top; filter2D itself uses torch only, so an empty stub module suffices) and run on the
    CPU in fp32 and fp64: per-sample 9 x 9 blur kernels (this fork's padded size, my_realesrgan_image_mask_dataset.py:
    108-109), per-sample 21 x 21 kernels (Real-ESRGAN's) with a sinc-like sign-changing one among them, one shared
    7 x 7 kernel, on an image whose sides are not multiples of the tile."""
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = types.ModuleType("cv2")
    spec = importlib.util.spec_from_file_location("ref_img_process_util2",
                                                  "/root/reference/GAN-Based-SR/basicsr/utils/img_process_util.py")
    ipu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ipu)
    if saved is not None:
        sys.modules["cv2"] = saved
    else:
        del sys.modules["cv2"]
    rng = np.random.default_rng(1313)
    img = (np.round(rng.random((3, 3, 45, 83)) * 255) / 255).astype(np.float32)
    out = dict(img=img)

    def blur(k):
        a = rng.random((3, k, k)) ** 3
        return (a / a.sum(axis=(1, 2), keepdims=True)).astype(np.float32)

    k21 = blur(21)
    yy, xx = np.mgrid[-10:11, -10:11]
    rr = np.sqrt(yy * yy + xx * xx) + 1e-6
    sinc = np.sin(1.2 * rr) / rr
    k21[1] = (sinc / sinc.sum()).astype(np.float32)
    for tag, kern in (("k9", blur(9)), ("k21", k21), ("k7s", blur(7)[:1])):
        o32 = ipu.filter2D(torch.as_tensor(img), torch.as_tensor(kern)).numpy()
        o64 = ipu.filter2D(torch.as_tensor(img).double(), torch.as_tensor(kern).double()).numpy()
        out["kern_" + tag], out["out32_" + tag], out["out64_" + tag] = kern, o32, o64
        print("f13", tag, "fp32 vs fp64:", np.abs(o32 - o64).max())
    save("f13_filter2d", **out)


def f14_diffjpeg():
    """F14: the reference's DiffJPEG(differentiable=False) (basicsr/utils/diffjpeg.py -- pure torch,
    imported by path) on a batch whose sides are not multiples of 16, with a per-sample quality tensor (the model's
    call, realesrganssl_model.py:199-201) and with a scalar quality (fp32 runs)."""
    spec = importlib.util.spec_from_file_location("ref_diffjpeg", "/root/reference/GAN-Based-SR/basicsr/utils/diffjpeg.py")
    dj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(dj)
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from ssl_amd import synth
    img = np.stack([synth.natural_like(1400 + i, 40, 52, 0.15, 0.05) for i in range(3)]).astype(np.float32)
    qual = np.array([30.0, 72.5, 95.0], np.float32)
    out = dict(img=img, quality=qual)
    m32 = dj.DiffJPEG(differentiable=False)   # (fp32 only: the quantisers' image.float() rules out a .double() run)
    with torch.no_grad():
        out["out32_t"] = m32(torch.as_tensor(img), quality=torch.as_tensor(qual.copy())).numpy()
        out["out32_s"] = m32(torch.as_tensor(img), quality=50).numpy()
    save("f14_diffjpeg", **out)


def f15_resize():
    """F15: the resizes of the degradation chain exactly as the reference calls them
    (realesrganssl_model.py:185,203,224,255,280,293): torch.nn.functional.interpolate(out, scale_factor=s | size=(h, w),
    mode='area' | 'bilinear' | 'bicubic') -- torch itself is the implementation the reference runs, so the fixture is
    torch's own CPU result, in fp32 (what a CPU run of the reference produces) and fp64 (the yardstick)."""
    rng = np.random.default_rng(1515)
    img = (np.round(rng.random((1, 3, 37, 45)) * 255) / 255).astype(np.float32)
    img = np.concatenate([img, img[:, ::-1, ::-1, ::-1] * np.float32(0.5)], 0)[:, :2]     # (2 samples, 2 channels)
    out = dict(img=img)
    cases = [("sf0.37", dict(scale_factor=0.37)), ("sf1.43", dict(scale_factor=1.43)), ("sf0.15", dict(scale_factor=0.15)),
             ("sf0.9", dict(scale_factor=0.9)), ("sf1", dict(scale_factor=1)), ("sz9x11", dict(size=(9, 11))),
             ("sz50x61", dict(size=(50, 61))), ("sz37x45", dict(size=(37, 45)))]
    out["cases"] = np.array([c[0] for c in cases])
    for mode in ("area", "bilinear", "bicubic"):
        for tag, kw in cases:
            o32 = F.interpolate(torch.as_tensor(img), mode=mode, **kw).numpy()
            o64 = F.interpolate(torch.as_tensor(img).double(), mode=mode, **kw).numpy()
            out[f"{mode}_{tag}_32"] = o32
            if tag in ("sf0.37", "sf1.43", "sz9x11"):      # (fp64 results for three cases per mode: fixture size)
                out[f"{mode}_{tag}_64"] = o64
    save("f15_resize", **out)


class _Recorder:
    """Proxy for a module (torch / numpy.random / random): the listed callables are logged (name, result) in call
    order, everything else passes through."""

    def __init__(self, mod, names, log, prefix):
        self._mod, self._names, self._log, self._prefix = mod, set(names), log, prefix

    def __getattr__(self, name):
        attr = getattr(self._mod, name)
        if name in self._names:
            def wrapped(*a, **k):
                r = attr(*a, **k)
                self._log.append((self._prefix + name, r.clone() if torch.is_tensor(r) else r))
                return r
            return wrapped
        return attr


def _load_degradations(log):
    """basicsr/data/degradations.py imported by path.  cv2 (unused by the *_pt functions) is an empty stub;
    torchvision's rgb_to_grayscale (not installed) is restated with its documented formula (0.2989 R + 0.587 G +
    0.114 B); the module's `torch` is a recorder for randn / rand / poisson."""
    saved = {k: sys.modules.get(k) for k in ("cv2", "torchvision", "torchvision.transforms",
                                             "torchvision.transforms.functional_tensor")}
    sys.modules["cv2"] = types.ModuleType("cv2")
    ft = types.ModuleType("torchvision.transforms.functional_tensor")
    ft.rgb_to_grayscale = lambda img, num_output_channels=1: (0.2989 * img[..., 0:1, :, :] + 0.587 * img[..., 1:2, :, :] +
                                                               0.114 * img[..., 2:3, :, :]).to(img.dtype)
    for name in ("torchvision", "torchvision.transforms"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms.functional_tensor"] = ft
    spec = importlib.util.spec_from_file_location("ref_degradations", "/root/reference/GAN-Based-SR/basicsr/data/degradations.py")
    deg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(deg)
    for k, v in saved.items():
        if v is not None:
            sys.modules[k] = v
        else:
            sys.modules.pop(k, None)
    deg.torch = _Recorder(torch, ("randn", "rand", "poisson"), log, "torch.")
    return deg


def f16_noise():
    """F16: the reference's random_add_gaussian_noise_pt / random_add_poisson_noise_pt
    (basicsr/data/degradations.py:544-548, 714-720 and what they call) run on the CPU in fp32 with every torch.rand /
    torch.randn / torch.poisson draw recorded, for batches with and without gray-noise samples."""
    log = []
    deg = _load_degradations(log)
    rng = np.random.default_rng(1616)
    img = (rng.random((3, 3, 20, 28)) * 1.1 - 0.05).astype(np.float32)      # (un-clamped input, like the chain's)
    out = dict(img=img)
    for tag, fn, kw in (("gauss_gray", deg.random_add_gaussian_noise_pt, dict(sigma_range=[1, 30], gray_prob=0.6)),
                        ("gauss_color", deg.random_add_gaussian_noise_pt, dict(sigma_range=[1, 30], gray_prob=0.0)),
                        ("gauss_round", deg.random_add_gaussian_noise_pt, dict(sigma_range=[1, 30], gray_prob=0.6, rounds=True)),
                        ("poisson_gray", deg.random_add_poisson_noise_pt, dict(scale_range=[0.05, 3], gray_prob=0.6)),
                        ("poisson_color", deg.random_add_poisson_noise_pt, dict(scale_range=[0.05, 3], gray_prob=0.0))):
        for seed in range(100):
            torch.manual_seed(1600 + seed)
            del log[:]
            o = fn(torch.as_tensor(img), clip=True, **kw)
            names = [n for n, _ in log]
            want_gray = "gray" in tag or "round" in tag
            has_gray = len(names) == 4
            if has_gray == want_gray and (not want_gray or 0 < float((log[1][1] < kw["gray_prob"]).sum()) < 3):
                break
        out[tag + "_out"] = o.numpy()
        out[tag + "_draws"] = np.array(names)
        for i, (n, v) in enumerate(log):
            out[f"{tag}_d{i}"] = v.numpy()
        out[tag + "_range"] = np.array(kw.get("sigma_range", kw.get("scale_range")), np.float64)
        out[tag + "_gray_prob"] = kw["gray_prob"]
        print("f16", tag, names, o.shape)
    save("f16_noise", **out)


def f17_feed_data():
    """F17: the reference's `feed_data` (basicsr/models/realesrganssl_model.py:148-316) itself -- the
    method's source is read from the reference at generation time and executed on a bare object on the CPU (fp32),
    with the reference's own filter2D / USMSharp (img_process_util.py), DiffJPEG (diffjpeg.py), noise functions
    (degradations.py) and paired_random_crop_img_mask (transforms.py) bound into its namespace, and EVERY random draw
    (random.choices / choice / randint, np.random.uniform, torch.rand / randn / poisson / randperm, the JPEG quality
    tensors) recorded in call order.  Run A uses the options of options/train/RealESRGANSSL/train_RealESRGANSSL_x4.yml
    (Use_sharpen unset); run B switches USM sharpening on and widens the ranges so that the other noise kind, gray
    noise and the other JPEG / resize-back order are exercised.  Third-party stand-ins as in F12 / F16: cv2's Gaussian
    kernel and flip, torchvision's rgb_to_grayscale, by their documented formulas."""
    import inspect
    import random
    import re
    from oracle import datapath_oracle as dorc
    log = []
    deg = _load_degradations(log)
    cv2 = types.ModuleType("cv2")
    cv2.getGaussianKernel = lambda ksize, sigma: dorc.gaussian_kernel_1d(int(ksize), float(sigma)).reshape(-1, 1)
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    ipu = load("ref_ipu17", "/root/reference/GAN-Based-SR/basicsr/utils/img_process_util.py")
    dj = load("ref_dj17", "/root/reference/GAN-Based-SR/basicsr/utils/diffjpeg.py")
    tr = load("ref_tr17", "/root/reference/GAN-Based-SR/basicsr/data/transforms.py")
    if saved is not None:
        sys.modules["cv2"] = saved
    else:
        del sys.modules["cv2"]
    tr.random = _Recorder(random, ("randint",), log, "random.")
    src = open("/root/reference/GAN-Based-SR/basicsr/models/realesrganssl_model.py").read()

    def method(name):
        m = re.search(r"    def %s\(self.*?\):.*?(?=\n    def |\n    @)" % name, src, re.S)
        return inspect.cleandoc("\n" + m.group(0)).replace(".cuda()", "")

    ns = {"torch": _Recorder(torch, ("randperm",), log, "torch."), "F": F,
          "np": types.SimpleNamespace(random=_Recorder(np.random, ("uniform",), log, "np.random.")),
          "random": _Recorder(random, ("choices", "choice"), log, "random."),
          # (.contiguous(): DiffJPEG returns a permuted view and, on the CPU, reflect-padding it keeps that layout, which
          # filter2D's .view() refuses; values are unaffected)
          "filter2D": lambda img, k: ipu.filter2D(img.contiguous(), k), "random_add_gaussian_noise_pt": deg.random_add_gaussian_noise_pt,
          "random_add_poisson_noise_pt": deg.random_add_poisson_noise_pt,
          "paired_random_crop_img_mask": tr.paired_random_crop_img_mask}
    exec(method("feed_data"), ns)
    exec(method("_dequeue_and_enqueue"), ns)
    jpeger = dj.DiffJPEG(differentiable=False)

    def jpeg(x, quality):
        log.append(("jpeg_q", quality.clone()))      # (the module overwrites `quality` with its factors)
        return jpeger(x, quality=quality)

    base = dict(degradation_order="two", scale=4, Use_sharpen=None, Sharpen_before_degra=False,
                resize_prob=[0.1, 0.85, 0.05], resize_range=[0.9, 1.1], gaussian_noise_prob=0.5, noise_range=[1, 8],
                poisson_scale_range=[0.05, 0.5], gray_noise_prob=0.2, jpeg_range=[85, 95], second_blur_prob=0.8,
                resize_prob2=[0.1, 0.85, 0.05], resize_range2=[0.9, 1.1], gaussian_noise_prob2=0.5, noise_range2=[0, 4],
                poisson_scale_range2=[0, 0.3], gray_noise_prob2=0.2, jpeg_range2=[87, 95], queue_size=4,
                datasets=dict(train=dict(gt_size=64)))
    wide = dict(base, Use_sharpen=True, Sharpen_before_degra=True, resize_prob=[0.3, 0.5, 0.2], resize_range=[0.5, 1.4],
                noise_range=[1, 30], poisson_scale_range=[0.05, 3], gray_noise_prob=0.5, jpeg_range=[30, 95],
                resize_prob2=[0.3, 0.4, 0.3], resize_range2=[0.4, 1.2], noise_range2=[1, 25],
                poisson_scale_range2=[0.05, 2.5], gray_noise_prob2=0.5, jpeg_range2=[30, 95])
    rng = np.random.default_rng(1717)
    B, S = 2, 96
    gt = np.stack([synth.natural_like(1700 + i, S, S, 0.12, 0.04) for i in range(B)]).astype(np.float32)
    mask = np.stack([synth.laplacian_edge_mask(gt[i])[None] for i in range(B)]).astype(np.float32)

    def kern():
        a = rng.random((B, 9, 9)) ** 4
        return (a / a.sum(axis=(1, 2), keepdims=True)).astype(np.float32)

    k1, k2 = kern(), kern()
    yy, xx = np.mgrid[-4:5, -4:5]
    rr = np.sqrt(yy * yy + xx * xx) + 1e-6
    sinc = np.sin(1.6 * rr) / rr
    sk = np.stack([sinc / sinc.sum(), np.eye(81)[40].reshape(9, 9)]).astype(np.float32)   # a sinc filter and the pulse
    out = dict(gt=gt, mask=mask, kernel1=k1, kernel2=k2, sinc_kernel=sk)

    def run(opt, seed):
        del log[:]
        random.seed(seed)
        np.random.seed(seed)
        torch.manual_seed(seed)
        obj = types.SimpleNamespace(device="cpu", opt=opt, queue_size=opt["queue_size"], jpeger=jpeg,
                                    usm_sharpener=ipu.USMSharp())
        obj._dequeue_and_enqueue = lambda: ns["_dequeue_and_enqueue"](obj)
        data = {k: torch.as_tensor(v.copy()) for k, v in (("gt", gt), ("gt_mask", mask), ("kernel1", k1),
                                                          ("kernel2", k2), ("sinc_kernel", sk))}
        with torch.no_grad():            # (the method's @torch.no_grad() decorator is not part of the extracted source)
            ns["feed_data"](obj, data)
        return obj, list(log)

    def pattern(lg):
        names = [n for n, _ in lg]
        u = [v for n, v in lg if n == "np.random.uniform"]
        return names, u

    want = {"A": lambda names, lg: names.count("torch.randn") >= 1 and names.count("torch.poisson") >= 1,
            "B": lambda names, lg: names.count("torch.poisson") == 2 or names.count("torch.randn") == 2}
    for tag, opt in (("A", base), ("B", wide)):
        for seed in range(1700, 1900):
            obj, lg = run(opt, seed)
            names = [n for n, _ in lg]
            # A: one Gaussian and one Poisson stage; B: a stage with gray noise (4 tensor draws) and the JPEG-first order
            if tag == "A":
                i = names.index("jpeg_q", names.index("jpeg_q") + 1)
                resize_first = names[i - 1] == "random.choice"                # [resize back + sinc] before the JPEG
                if not (names.count("torch.randn") >= 1 and names.count("torch.poisson") >= 1 and resize_first):
                    continue
            if tag == "B":
                i = names.index("jpeg_q", names.index("jpeg_q") + 1)          # second JPEG: directly followed by the
                jpeg_first = names[i + 1:i + 2] == ["random.choice"]          # resize-back mode draw <=> JPEG first
                gray = any(names[j:j + 4] in (["torch.rand", "torch.rand", "torch.randn", "torch.randn"],
                                               ["torch.rand", "torch.rand", "torch.poisson", "torch.poisson"])
                           for j in range(len(names)))
                if not (jpeg_first and gray):
                    continue
            break
        else:
            raise RuntimeError("no seed with the wanted draw pattern for run " + tag)
        print(f"f17 run {tag}: seed {seed}, {len(lg)} draws:", [n.split('.')[-1] for n in names])
        out[f"{tag}_seed"] = seed
        out[f"{tag}_opt"] = np.array([repr(opt)])
        out[f"{tag}_draw_names"] = np.array(names)
        for i, (n, v) in enumerate(lg):
            if torch.is_tensor(v):
                v = v.numpy()
            elif isinstance(v, list):
                v = np.array(v)
            out[f"{tag}_d{i}"] = np.asarray(v)
        out[f"{tag}_lq"], out[f"{tag}_gt"], out[f"{tag}_mask"] = obj.lq.numpy(), obj.gt.numpy(), obj.gt_mask.numpy()
        if opt["Use_sharpen"] is not None:
            out[f"{tag}_gt_usm"] = obj.gt_usm.numpy()
    save("f17_feed_data", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["f1", "f2s", "f2", "f3", "f4", "f5", "f7", "f8", "f9", "f10", "f11", "f12", "f13", "f14", "f15", "f16", "f17"]
    torch.manual_seed(0)
    if "f1" in which:
        f1_c1()
    if "f2s" in which:
        f2_paper(128, (1.0, 0.004), "f2_paper_128")
    if "f2" in which:
        f2_paper(256, (0.004,), "f2_paper_256")
    if "f3" in which:
        f3_masks()
    if "f4" in which:
        f4_stress()
    if "f5" in which:
        f5_stride_f6_eps()
    if "f7" in which:
        f7_mask_pil()
    if "f8" in which:
        f8_dm_strategies()
    if "f9" in which:
        f9_c4_dm_loop()
    if "f10" in which:
        f10_paper_cotangent()
    if "f11" in which:
        f11_datapath()
    if "f12" in which:
        f12_usm()
    if "f13" in which:
        f13_filter2d()
    if "f14" in which:
        f14_diffjpeg()
    if "f15" in which:
        f15_resize()
    if "f16" in which:
        f16_noise()
    if "f17" in which:
        f17_feed_data()
    if "time" in which:
        cpu_reference_timing()
    if "timegrid" in which:
        cpu_reference_timing_grid()
