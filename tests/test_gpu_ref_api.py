"""Row R1: the reference's UNCHANGED caller loop on the GPU, end to end, behind the reference's own API.

The loop (ssl_amd/reference_loop.py: realesrganssl_model.py:379-430 and ddpmssl.py:438-513 restated) is run against
`ssl_amd.losses.{loss_util.similarity_map, basic_loss.L1Loss, basic_loss.KLDistanceLoss}` (and the DM fork's
`dm_loss_util.similarity_map`) for every `ssl_mode`, with deferred handles (the default) and with eager tensors, and
checked against the caller-loop values the fixtures hold: `l1`, `kl`, `grad` of F1, F2 (both sizes, both sigmas), F3
(empty image in the batch, 3-channel masks) and F9 (the DM fork's loop, stride 3, eps 1e-20).  Tolerances are those of
tests/test_gpu_parity.py (module docstring there).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MODES = ["cuda", "pytorch", "hip"]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from ssl_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


@pytest.fixture(params=[True, False], ids=["lazy", "eager"])
def lazy(request):
    from ssl_amd.losses import set_lazy
    prev = set_lazy(request.param)
    yield request.param
    set_lazy(prev)


def T(a, dev, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def graph_nodes(t, name):
    """Autograd nodes called `name` reachable from tensor t."""
    seen, found, todo = set(), set(), [t.grad_fn]
    while todo:
        f = todo.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        if type(f).__name__ == name:
            found.add(f)
        todo.extend(n for n, _ in f.next_functions)
    return found


def step_nodes(t):
    """The fused step's autograd nodes reachable from tensor t."""
    return graph_nodes(t, "_LazyStepFnBackward")


def run_gan_loop(dev, sr_np, gt_np, mask_np, mode, ks, kw, sigma, w=1e3, stride=0, gen=True):
    """The GAN model's SSL block followed by `l_g_total.backward()`; `sr` plays the generator's output."""
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map
    from ssl_amd.reference_loop import gan_selfsim_block, stride_pattern
    sr = T(sr_np, dev).requires_grad_(True)
    out = sr * 1.0                        # a non-leaf, as the generator's output is
    setting = dict(ssl_mode=mode, kernel_size_search=ks, generalization=gen, kernel_size_window=kw, sigma=sigma)
    pat = stride_pattern(sr_np.shape[-1], stride, dev) if stride > 1 else None
    l1, kl = gan_selfsim_block(similarity_map, L1Loss(loss_weight=w, reduction='mean'), KLDistanceLoss(loss_weight=w),
                               out, T(gt_np, dev), T(mask_np, dev), setting, pat)
    if l1 is None:
        return None, None, None
    (l1 + kl).backward()
    return float(l1), float(kl), sr.grad.cpu().numpy()


@pytest.mark.parametrize("mode", MODES)
def test_unchanged_loop_f1_c1(dev, golden, lazy, mode):
    g = golden("f1_c1_64")
    l1, kl, grad = run_gan_loop(dev, g["sr"], g["gt"], g["mask"], mode, int(g["ks"]), int(g["kw"]), 1.0)
    assert abs(l1 - float(g["l1_f64"])) <= 1e-5 * float(g["l1_f64"])
    assert abs(kl - float(g["kl_f64"])) <= 1e-5 * float(g["kl_f64"])
    assert maxerr(grad, g["grad"]) <= 1e-5 * np.abs(g["grad"]).max()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["f2_paper_128", "f2_paper_256"])
def test_unchanged_loop_f2_paper_config(dev, golden, lazy, mode, name):
    g = golden(name)
    ks, kw = int(g["ks"]), int(g["kw"])
    sigmas = [k[len("l1_s"):-len("_f64")] for k in g.files if k.startswith("l1_s") and k.endswith("_f64")]
    for sg in sigmas:
        l1, kl, grad = run_gan_loop(dev, g["sr"], g["gt"], g["mask"], mode, ks, kw, float(sg))
        rl1, rkl = float(g[f"l1_s{sg}_f64"]), float(g[f"kl_s{sg}_f64"])
        assert abs(l1 - rl1) <= 1e-5 * abs(rl1), (sg, l1, rl1)
        assert abs(kl - rkl) <= 1e-5 * abs(rkl) + 2e-8, (sg, kl, rkl)        # (see test_f2_paper_config_rows_losses_grad)
        ref = g[f"grad_s{sg}"]
        tol = max(1e-5, 3.0 * float(g[f"grad_ref32_dev_s{sg}"])) * np.abs(ref).max()
        assert maxerr(grad, ref) <= tol, (sg, maxerr(grad, ref) / np.abs(ref).max())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("channels", [1, 3])
def test_unchanged_loop_f3_empty_image_and_three_channel_masks(dev, golden, lazy, mode, channels):
    """Batch of two, image 0 with an empty mask (skipped by the loop); with the pair pool's 3-channel masks
    'cuda' / 'hip' read channel 0 and 'pytorch' lists every row three times -- the means do not move."""
    g = golden("f3_masks")
    masks = np.repeat(g["b2_masks"], channels, axis=1)
    l1, kl, grad = run_gan_loop(dev, g["sr"], g["gt"], masks, mode, int(g["ks"]), int(g["kw"]), float(g["sigma"]))
    assert abs(l1 - float(g["b2_l1_f64"])) <= 1e-5 * float(g["b2_l1_f64"])
    assert abs(kl - float(g["b2_kl_f64"])) <= 1e-5 * float(g["b2_kl_f64"])
    assert np.abs(grad[0]).max() == 0.0
    assert maxerr(grad, g["b2_grad"]) <= 1e-5 * np.abs(g["b2_grad"]).max()
    # every mask empty: the block adds nothing (GAN model) / returns 0.0, 0.0 (DM fork)
    assert run_gan_loop(dev, g["sr"], g["gt"], np.zeros_like(masks), mode, int(g["ks"]), int(g["kw"]), 1.0)[0] is None


def test_unchanged_dm_issl_loop_f9(dev, golden, lazy):
    """The diffusion fork's `issl` with its configured options (mask_stride 3, nonlocalavg_cuda_v1 = eps 1e-20, k 25 / 9,
    scaling_factor 0.004, softmax, weights 5e2) against fixture F9 = the fork's own loop."""
    from oracle import ssg_oracle as orc
    from ssl_amd.losses import KLDistanceLoss, L1Loss
    from ssl_amd.losses.dm_loss_util import similarity_map
    from ssl_amd.reference_loop import dm_issl, stride_pattern
    g = golden("f9_c4_dm_loop")
    ks, kc, sigma, stride, w = int(g["ks"]), int(g["kc"]), float(g["sigma"]), int(g["stride"]), float(g["w"])
    sslopt = dict(mask_stride=stride, simself_strategy="areaarea_mask_nonlocalavg_cuda_v1", kernel_size=ks,
                  scaling_factor=sigma, softmax_sr=True, softmax_gt=True, kernel_size_center=kc)
    sr = T(g["sr"], dev).requires_grad_(True)
    l1, kl = dm_issl(similarity_map, L1Loss(loss_weight=w), KLDistanceLoss(loss_weight=w), sr * 1.0, T(g["gt"], dev),
                     T(g["mask"], dev), sslopt, stride_pattern(128, stride, dev))
    (l1 + kl).backward()
    assert abs(float(l1) - float(g["l1"])) <= 1e-5 * float(g["l1"])
    assert abs(float(kl) - float(g["kl"])) <= 1e-5 * float(g["kl"])
    sr_np, gt_np = g["sr"], g["gt"]
    masks_s = np.stack([orc.mask_stride(g["mask"][i, 0], stride) for i in range(2)])
    ref = orc.ssg_loss(sr_np.astype(np.float64), gt_np.astype(np.float64), masks_s, ks, kc, sigma, w, w, eps=1e-20)
    r32 = orc.ssg_loss(sr_np, gt_np, masks_s, ks, kc, sigma, w, w, eps=1e-20)
    mx = np.abs(ref["grad"]).max()
    tol = max(1e-5, 4.0 * np.abs(r32["grad"].astype(np.float64) - ref["grad"]).max() / mx) * mx
    assert maxerr(sr.grad.cpu(), g["grad"]) <= tol
    # all masks empty
    z = dm_issl(similarity_map, L1Loss(), KLDistanceLoss(), sr * 1.0, T(g["gt"], dev), torch.zeros(2, 1, 128, 128, device=dev),
                sslopt, stride_pattern(128, stride, dev))
    assert z == (0.0, 0.0)


def test_lazy_and_eager_loops_agree_and_lazy_is_one_node(dev):
    """Same inputs through the deferred and the eager loop: losses to 2e-6, gradient to 3e-6 of its maximum; the
    deferred loop's two criteria hang off ONE autograd node and no SSG tensor was materialised."""
    from ssl_amd import synth
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
    from ssl_amd.losses.lazy import LazySSG
    from ssl_amd.reference_loop import gan_selfsim_block
    sr_np, gt_np, mask_np = synth.make_batch(3, 96, 112, seed0=400)
    mask_np[1] = 0
    setting = dict(ssl_mode='cuda', kernel_size_search=25, generalization=True, kernel_size_window=9, sigma=0.004)
    res = {}
    for lz in (True, False):
        prev = set_lazy(lz)
        try:
            sr = T(sr_np, dev).requires_grad_(True)
            handles = []

            class spy(similarity_map):
                def getitem(self):
                    handles.append(self.s)
                    return self.s
            l1, kl = gan_selfsim_block(spy, L1Loss(1e3), KLDistanceLoss(1e3), sr * 1.0, T(gt_np, dev), T(mask_np, dev),
                                       setting)
            if lz:
                assert all(isinstance(h, LazySSG) and h._t is None for h in handles)
                n1, n2 = step_nodes(l1), step_nodes(kl)
                assert len(n1) == 1 and n1 == n2
            else:
                assert all(isinstance(h, torch.Tensor) for h in handles)
            (l1 + kl).backward()
            res[lz] = (float(l1), float(kl), sr.grad.clone())
        finally:
            set_lazy(prev)
    (a1, a2, ga), (b1, b2, gb) = res[True], res[False]
    assert abs(a1 - b1) <= 2e-6 * abs(b1) and abs(a2 - b2) <= 2e-6 * abs(b2)
    assert float((ga - gb).abs().max()) <= 3e-6 * float(gb.abs().max())
    assert float(ga[1].abs().max()) == 0.0


def test_handle_behaves_like_the_tensor_when_used_any_other_way(dev, golden):
    """Whatever is not the cat + criteria pattern materialises the handle: values of the eager tensor."""
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
    from ssl_amd.losses.lazy import LazySSG
    g = golden("f1_c1_64")
    ks, kw = int(g["ks"]), int(g["kw"])
    prev = set_lazy(True)
    try:
        img = T(g["sr"], dev).requires_grad_(True)
        mask = T(g["mask"], dev)
        h = similarity_map(img, mask, 'cuda', ks, True, kw, 1.0).getitem()
        assert isinstance(h, LazySSG) and h._t is None
        assert tuple(h.shape) == (1,) + g["ssg_s1.0_g1"].shape          # attribute access computes the rows
        assert maxerr(h.detach().cpu()[0], g["ssg_s1.0_g1"]) <= 1e-5
        (h * T(g["cot"], dev)).sum().backward()
        assert maxerr(img.grad.cpu()[0], g["dimg_s1.0_g1"]) <= 1e-5 * np.abs(g["dimg_s1.0_g1"]).max()
        assert maxerr(torch.exp(h).detach().cpu(), np.exp(g["ssg_s1.0_g1"])[None]) <= 2e-5        # any other torch function
        assert maxerr(h[0, :5].detach().cpu(), g["ssg_s1.0_g1"][:5]) <= 1e-5
        # criteria with options the fused step does not cover fall back to the tensors
        gt = T(g["gt"], dev)
        a = similarity_map(img, mask, 'hip', ks, True, kw, 1.0).getitem()
        b = similarity_map(gt, mask, 'hip', ks, True, kw, 1.0).getitem()
        ref_l1 = float(g["l1_f64"]) / 1e3
        assert abs(float(L1Loss(1.0, 'sum')(a, b)) - ref_l1 * a.numel()) <= 2e-5 * ref_l1 * a.numel()
        a2 = similarity_map(img, mask, 'hip', ks, True, kw, 1.0).getitem()
        b2 = similarity_map(gt, mask, 'hip', ks, True, kw, 1.0).getitem()
        sm = float(KLDistanceLoss(1.0, softmax=True)(a2, b2))
        # (softmax of rows in [0,1]: nearly uniform rows, KL ~ 3e-10 -- a second-order quantity that fp32 evaluations
        # of the reference's expression only agree on to ~1 %; the yardstick is the fp64 evaluation of the same rows)
        ea, eb = a2.materialise().softmax(-1).double(), b2.materialise().softmax(-1).double()   # (fp32 softmax, like the module)
        want = float(torch.nn.functional.kl_div(ea.clamp(min=1e-10).log(), eb.clamp(min=1e-10), reduction='mean'))
        assert abs(sm - want) <= 2e-2 * abs(want) + 1e-12, (sm, want)
        # torch.nn.L1Loss (reduction inside F.l1_loss) takes the fused step too
        a3 = similarity_map(img, mask, 'hip', ks, True, kw, 1.0).getitem()
        b3 = similarity_map(gt, mask, 'hip', ks, True, kw, 1.0).getitem()
        assert abs(float(torch.nn.L1Loss()(a3, b3)) - ref_l1) <= 1e-5 * ref_l1 and a3._t is None
        # settings that differ between the two sides: eager rows, still the right numbers
        a4 = similarity_map(img, mask, 'hip', ks, True, kw, 1.0).getitem()
        b4 = similarity_map(gt, mask, 'hip', ks, False, kw, 1.0).getitem()
        v = float(L1Loss(1.0)(a4, b4))
        assert a4._t is not None and abs(v - float((a4.materialise() - b4.materialise()).abs().mean())) <= 1e-6 * v
        # two different masks on the two sides of a fused pair: NaN, not a silently wrong number
        m2 = mask.clone()
        m2[0, 0, 5, 5] = 1 - m2[0, 0, 5, 5]
        a5 = similarity_map(img, mask, 'hip', ks, True, kw, 1.0).getitem()
        b5 = similarity_map(gt, m2, 'hip', ks, True, kw, 1.0).getitem()
        assert torch.isnan(L1Loss(1.0)(a5, b5))
    finally:
        set_lazy(prev)


def test_reference_style_criterion_code_reaches_the_fused_step(dev, golden):
    """A training model that keeps the reference's OWN basic_loss.py (only `similarity_map` swapped): its criteria are
    `weight_reduce_loss(F.l1_loss(pred, target, reduction='none'))` behind the `weighted_loss` decorator
    (loss_util.py:33-101, basic_loss.py:14-16) and `F.kl_div(torch.clamp(input=x, min=1e-10).log(), torch.clamp(input=y,
    min=1e-10), reduction=...)` (basic_loss.py:281).  Written out here as plain functions with that call sequence."""
    import torch.nn.functional as F
    from ssl_amd.losses import similarity_map, set_lazy
    g = golden("f1_c1_64")
    ks, kw = int(g["ks"]), int(g["kw"])

    def their_l1(pred, target, loss_weight):
        loss = F.l1_loss(pred, target, reduction='none')
        reduction_enum = F._Reduction.get_enum('mean')
        loss = loss if reduction_enum == 0 else loss.mean() if reduction_enum == 1 else loss.sum()
        return loss_weight * loss

    def their_kl(x, y, loss_weight):
        return loss_weight * F.kl_div(torch.clamp(input=x, min=1e-10).log(), torch.clamp(input=y, min=1e-10),
                                      reduction='mean')

    prev = set_lazy(True)
    try:
        sr = T(g["sr"], dev).requires_grad_(True)
        mask = T(g["mask"], dev)
        a = torch.cat([similarity_map(sr.clone(), mask.clone(), 'cuda', ks, True, kw, 1.0).getitem()], dim=1)
        b = torch.cat([similarity_map(T(g["gt"], dev), mask.clone(), 'cuda', ks, True, kw, 1.0).getitem()], dim=1)
        l1, kl = their_l1(a, b, 1e3), their_kl(a, b, 1e3)
        assert a._t is None and b._t is None
        (l1 + kl).backward()
        assert abs(float(l1) - float(g["l1_f64"])) <= 1e-5 * float(g["l1_f64"])
        assert abs(float(kl) - float(g["kl_f64"])) <= 1e-5 * float(g["kl_f64"])
        assert maxerr(sr.grad.cpu(), g["grad"]) <= 1e-5 * np.abs(g["grad"]).max()
    finally:
        set_lazy(prev)


def test_loop_with_different_weights_and_a_single_criterion(dev, golden):
    """Autograd's two incoming gradients reach the node separately (device floats): different weights for L1 and KL,
    and a loop that uses only one of the two criteria."""
    from ssl_amd import SSGLoss
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
    from ssl_amd.reference_loop import gan_selfsim_block
    g = golden("f2_paper_128")
    ks, kw = int(g["ks"]), int(g["kw"])
    setting = dict(ssl_mode='cuda', kernel_size_search=ks, generalization=True, kernel_size_window=kw, sigma=0.004)
    gt, mask = T(g["gt"], dev), T(g["mask"], dev)
    prev = set_lazy(True)
    try:
        for w1, w2 in ((1e3, 2.5e2), (7.0, 0.0), (0.0, 3.0)):
            sr = T(g["sr"], dev).requires_grad_(True)
            l1, kl = gan_selfsim_block(similarity_map, L1Loss(w1) if w1 else None, KLDistanceLoss(w2) if w2 else None,
                                       sr * 1.0, gt, mask, setting)
            sum(t for t in (l1, kl) if t is not None).backward()
            x = T(g["sr"], dev).requires_grad_(True)
            r1, r2 = SSGLoss(ks, kw, 0.004, True, w1, w2)(x, gt, mask)
            (r1 + r2).backward()
            if w1:
                assert abs(float(l1) - float(r1)) <= 2e-6 * float(r1)
            if w2:
                assert abs(float(kl) - float(r2)) <= 2e-6 * float(r2)
            assert float((sr.grad - x.grad).abs().max()) <= 3e-6 * float(x.grad.abs().max())
    finally:
        set_lazy(prev)


def test_unchanged_loop_at_c2_full_size_equals_the_batched_step(dev):
    """BASELINE configs[1] at full size (16 x 3x256x256, Laplacian masks): the reference's unchanged loop on deferred
    handles gives the batched step's numbers -- losses to 2e-6, gradient to 3e-6 of its maximum (the batched step is
    checked against the fp64 oracle at this size in test_c2_full_size_properties)."""
    from ssl_amd import engine, synth
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
    from ssl_amd.reference_loop import gan_selfsim_block
    sr_np, gt_np, mask_np = synth.make_batch(16, 256, 256, seed0=100)
    n = int(mask_np.sum())
    step = engine.LossStep(16, 3, 256, 256, 25, 9, 0.004, 1e-10, True, 1e3, 1e3, device=dev, capacity=n + 64)
    loss, grad = step(T(sr_np, dev), T(gt_np, dev), T(mask_np, dev))
    setting = dict(ssl_mode='cuda', kernel_size_search=25, generalization=True, kernel_size_window=9, sigma=0.004)
    prev = set_lazy(True)
    try:
        sr = T(sr_np, dev).requires_grad_(True)
        l1, kl = gan_selfsim_block(similarity_map, L1Loss(1e3), KLDistanceLoss(1e3), sr * 1.0, T(gt_np, dev),
                                   T(mask_np, dev), setting)
        (l1 + kl).backward()
    finally:
        set_lazy(prev)
    assert abs(float(l1) - float(loss[0])) <= 2e-6 * float(loss[0]) and abs(float(kl) - float(loss[1])) <= 2e-6 * float(loss[1])
    assert float((sr.grad - grad).abs().max()) <= 3e-6 * float(grad.abs().max())


def test_native_criteria_on_foreign_tensors_nan_and_extreme_ratios(dev):
    """L1Loss / KLDistanceLoss accept ANY fp32 GPU tensors, not only SSG rows: a NaN in either operand
    reaches both losses and the gradient as it does through torch.clamp / F.kl_div / sign (v_max_f32 alone would drop it);
    an L1-only gradient never sees the KL ratio (|target| / pred beyond fp32: no NaN / inf); tensors on the same device
    only; `set_native_criteria(False)` gives the reference's torch expressions (double backward works there)."""
    from ssl_amd.losses import KLDistanceLoss, L1Loss, set_native_criteria
    a = torch.rand(1, 64, 121, device=dev) * 1e-3
    b = torch.rand(1, 64, 121, device=dev) * 1e-3
    for where in ("pred", "target"):
        x, y = a.clone(), b.clone()
        (x if where == "pred" else y)[0, 3, 7] = float("nan")
        x.requires_grad_(True)
        l1, kl = L1Loss(1.0)(x, y), KLDistanceLoss(1.0)(x, y)
        assert torch.isnan(l1) and torch.isnan(kl), where
        (l1 + kl).backward()
        assert torch.isnan(x.grad[0, 3, 7]) and bool(torch.isfinite(x.grad[0, 0]).all())
    x = a.clone()
    x[0, 0, :8] = 1e-38                       # pred tiny, target huge: the ratio overflows fp32
    y = b.clone()
    y[0, 0, :8] = 3e38
    x.requires_grad_(True)
    L1Loss(1.0, "sum")(x, y).backward()
    assert bool(torch.isfinite(x.grad).all()) and bool((x.grad[0, 0, :8] == -1.0).all())
    prev = set_native_criteria(False)
    try:
        z = a.clone().requires_grad_(True)
        l = KLDistanceLoss(1.0)(z, b)
        assert not graph_nodes(l, "_CriterionSumBackward")
        g, = torch.autograd.grad(l, z, create_graph=True)
        g.pow(2).sum().backward()             # second derivative through the torch expressions
        assert z.grad is not None and bool(torch.isfinite(z.grad).all())
    finally:
        set_native_criteria(prev)


def test_criterion_modules_on_materialised_tensors_match_torch(dev):
    """L1Loss / KLDistanceLoss on real fp32 GPU tensors run the engine's streaming criteria kernels (one pass forward,
    one backward); values and gradients against the reference's torch expressions evaluated in fp64 (basic_loss.py:16,
    281): sums to rel 2e-6 (fp64 accumulation inside), gradients element-wise to 2e-6 of their maximum.  Covers the
    clamp at 1e-10 on both sides, exact ties (sign 0), odd sizes / unaligned views (scalar path), every reduction the
    kernels take, and the cases that stay on torch (weight=, 'none', a target that wants a gradient, fp64, CPU)."""
    import torch.nn.functional as F
    from ssl_amd.losses import KLDistanceLoss, L1Loss
    from ssl_amd.losses.basic_loss import _CriterionSum
    g = torch.Generator(device="cpu").manual_seed(11)
    for shape, offset in (((1, 777, 625), 0), ((1, 33, 121), 0), ((2, 5, 49), 0), ((1, 1031), 1)):
        n = int(np.prod(shape))
        base_a = torch.rand(n + 4, generator=g) ** 6
        base_b = torch.rand(n + 4, generator=g) ** 6
        base_a[:50] = 0.0                      # below the KL clamp
        base_b[25:75] = 0.0
        base_a[100:120] = base_b[100:120]      # exact ties: sign(0) = 0
        base_a[200] = 3e-11
        A = base_a.to(dev)[offset:offset + n].view(shape)        # (offset 1: a 4-byte aligned view)
        Bt = base_b.to(dev)[offset:offset + n].view(shape)
        a64 = A.double().detach().requires_grad_(True)
        b64 = Bt.double()
        for red in ("mean", "sum"):
            x = A.detach().clone().requires_grad_(True) if offset == 0 else A.detach().requires_grad_(True)
            l1 = L1Loss(2.5, red)(x, Bt)
            kl = KLDistanceLoss(0.75, red)(x, Bt)
            assert graph_nodes(l1, "_CriterionSumBackward") and graph_nodes(kl, "_CriterionSumBackward")
            (l1 + kl).backward()
            r1 = 2.5 * F.l1_loss(a64, b64, reduction=red)
            r2 = 0.75 * F.kl_div(torch.clamp(input=a64, min=1e-10).log(), torch.clamp(input=b64, min=1e-10), reduction=red)
            a64.grad = None
            (r1 + r2).backward()
            assert abs(float(l1) - float(r1)) <= 2e-6 * abs(float(r1)), (shape, red, float(l1), float(r1))
            assert abs(float(kl) - float(r2)) <= 2e-6 * abs(float(r2)) + 1e-12, (shape, red, float(kl), float(r2))
            ref = a64.grad
            assert float((x.grad.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max()), (shape, red)
        x = A.detach().clone().requires_grad_(True)
        kb = KLDistanceLoss(1.0, "batchmean")(x, Bt)
        want = F.kl_div(torch.clamp(input=a64.detach(), min=1e-10).log(), torch.clamp(input=b64, min=1e-10), reduction="batchmean")
        assert abs(float(kb) - float(want)) <= 2e-6 * abs(float(want))
    # run-to-run bits
    A = torch.rand(3, 1000, 625, device=dev)
    Bt = torch.rand(3, 1000, 625, device=dev)
    s = [float(_CriterionSum.apply(A, Bt, k)) for k in (0, 1, 0, 1)]
    assert s[0] == s[2] and s[1] == s[3]
    # what stays on torch's own expressions
    w = torch.ones_like(A)
    x = A.clone().requires_grad_(True)
    assert not graph_nodes(L1Loss(1.0)(x, Bt, weight=w), "_CriterionSumBackward")
    assert L1Loss(1.0, "none")(x, Bt).shape == A.shape
    t = Bt.clone().requires_grad_(True)
    (L1Loss(1.0)(x, t) + KLDistanceLoss(1.0)(x, t)).backward()
    assert t.grad is not None and float(t.grad.abs().max()) > 0
    assert float(L1Loss(1.0)(A.double(), Bt.double())) > 0 and float(L1Loss(1.0)(A.cpu(), Bt.cpu())) > 0


def test_deferred_and_eager_loops_agree_on_random_batches(dev):
    """Property test of row R1: random batch sizes, image sizes that are not multiples of the tile sizes, mask
    densities (with empty images), mask dtypes and channel counts, kernel sizes with and without shared-term kernels,
    sigmas, all three `ssl_mode`s, the stride pattern -- the reference's unchanged loop on deferred handles and on eager
    tensors gives the same two losses (5e-6) and the same gradient up to the L1 term's sign ties: the eager rows of one
    small image come from the direct kernels, the batched step's from dense tiles + direct kernels (<= 2e-6 apart), and
    an entry whose sign(s_sr - s_gt) fp32 does not decide moves the gradient by ~1e-4 of its maximum per flip (see
    ref_grad_with_gpu_signs in test_gpu_parity.py): bounded at 5e-4."""
    from ssl_amd import synth
    from ssl_amd.losses import KLDistanceLoss, L1Loss, similarity_map, set_lazy
    from ssl_amd.reference_loop import gan_selfsim_block, stride_pattern
    rng = np.random.default_rng(77)
    cases = [(25, 9, 3, 40, 72, 0.3, 1, np.float32, 'cuda', 0.05, 0), (25, 9, 2, 37, 61, 0.08, 3, np.float32, 'pytorch', 1.0, 0),
             (11, 5, 4, 30, 34, 0.2, 1, np.uint8, 'hip', 0.5, 0), (7, 3, 2, 24, 20, 0.5, 1, np.float32, 'cuda', 0.1, 0),
             (25, 9, 2, 64, 64, 0.6, 1, np.float32, 'cuda', 0.004, 3), (49, 13, 1, 56, 60, 0.15, 1, np.float32, 'hip', 1.0, 0)]
    for ks, kw, B, H, W, dens, c1, mdt, mode, sigma, stride in cases:
        gt = np.stack([synth.natural_like(int(rng.integers(1 << 20)), H, W) for _ in range(B)])
        sr = np.stack([synth.degrade(gt[i], int(rng.integers(1 << 20))) for i in range(B)])
        mask = (rng.random((B, 1, H, W)) < dens).astype(mdt)
        if B > 2:
            mask[1] = 0
        mask = np.repeat(mask, c1, axis=1)
        setting = dict(ssl_mode=mode, kernel_size_search=ks, generalization=True, kernel_size_window=kw, sigma=sigma)
        pat = stride_pattern(W, stride, dev)[:, :, :H, :W] if stride > 1 else None
        if stride > 1 and H != W:
            pat = None
        res = {}
        for lz in (True, False):
            prev = set_lazy(lz)
            try:
                x = T(sr, dev).requires_grad_(True)
                m = torch.as_tensor(mask, device=dev)
                if pat is not None:
                    m = m.float()
                l1, kl = gan_selfsim_block(similarity_map, L1Loss(1e3), KLDistanceLoss(1e3), x * 1.0, T(gt, dev), m,
                                           setting, pat)
                (l1 + kl).backward()
                res[lz] = (float(l1.detach()), float(kl.detach()), x.grad.clone())
            finally:
                set_lazy(prev)
        (a1, a2, ga), (b1, b2, gb) = res[True], res[False]
        # (KL at sigma = 1 is a cancelling second-order sum, ~2e-6 after w = 1e3: the slack of test_f2_paper_config_...)
        assert abs(a1 - b1) <= 5e-6 * abs(b1) and abs(a2 - b2) <= 1e-5 * abs(b2) + 2e-8, (ks, mode, a1, b1, a2, b2)
        d, mx = (ga - gb).abs(), float(gb.abs().max())
        assert float(d.max()) <= 5e-4 * mx, (ks, mode, float(d.max()) / mx)     # (one flipped entry reaches 33 x 33 pixels)
