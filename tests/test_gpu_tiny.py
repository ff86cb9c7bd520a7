"""The two-launch step of small (11,5) calls (ssl_amd/csrc/ssg_tiny.hip: one workgroup builds the edge list, one workgroup
per edge pixel does forward, criteria and backward, the last one folds the step) -- BASELINE's configs[0] (C1), the
reference's own CPU-runnable case (loss_util.py:185-229, basic_loss.py:66,281, realesrganssl_model.py:411-430).

The path is the DEFAULT for such calls, so test_gpu_parity.py's (11,5) LossStep cases (F1 golden, the batch of three with
an empty mask) already run through it; this file adds the cases of its own structure -- every mask kind of the one-workgroup
builder, more rows than workgroups (the row loop), heavily reflected images, capacity cuts, the gradient as an output / an
accumulator / absent, no SSG output, a HIP-graph replay -- against the fp64 oracle, and re-runs the (11,5) parity cases with
the path switched off (ssg_set_tiny_step(0)) so that the general kernels of that size stay covered.
"""
import numpy as np
import pytest
import torch

from oracle import ssg_oracle as orc
import test_gpu_parity as tp
from test_gpu_parity import T, maxerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from ssl_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


class tiny_off:
    def __enter__(self):
        from ssl_amd import engine
        self.prev = engine.set_tiny_step(False)

    def __exit__(self, *exc):
        from ssl_amd import engine
        engine.set_tiny_step(self.prev)


def _raw_step(dev, fn_name, sr, gt, mask, kind, mc, ks, kw, sigma, cap, det, stride=0, thr=20.0, gen=1, eps=1e-10, fused=False,
              grad_fill=None, want_grad=True):
    """One call of ssg_loss_step / ssg_loss_fwd_bwd through ctypes; returns numpy results and the workspace's edge list."""
    from ssl_amd import _lib, engine
    L = _lib.lib()
    B, C, H, W = sr.shape
    P = ks * ks
    nb = L.ssg_loss_workspace_bytes(B, H, W, cap, ks) + (L.ssg_loss_rows_bytes(cap, ks) if fused else 0)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    counts = torch.full((B + 2,), -7, dtype=torch.int32, device=dev)
    loss = torch.full((2,), -1.0, device=dev)
    grad = torch.full((B, C, H, W), 0.0 if grad_fill is None else grad_fill, device=dev) if want_grad else None
    fix = torch.empty(L.ssg_grad_fix_bytes(B, C, H, W), dtype=torch.uint8, device=dev).fill_(0x5A) if det else None
    s_sr = None if fused else torch.full((cap, P), -3.0, device=dev)
    s_gt = None if fused else torch.full((cap, P), -3.0, device=dev)
    p = engine._ptr
    mt = None if mask is None else torch.as_tensor(np.ascontiguousarray(mask), device=dev)
    srt, gtt = T(sr, dev), T(gt, dev)          # (kept alive: two temporaries would be handed the same block)
    _lib.check(getattr(L, fn_name)(p(srt), p(gtt), p(mt), kind, mc, B, C, H, W, ks, kw, sigma, eps, gen, 1e3, 1e3,
                                   stride, thr, cap, p(s_sr), p(s_gt), p(counts), p(loss), p(grad), p(ws), nb, p(fix),
                                   torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    edges = ws[: cap * 12].view(torch.int32).view(cap, 3).cpu().numpy()
    return dict(loss=loss.cpu().numpy(), grad=None if grad is None else grad.cpu().numpy(), counts=counts.cpu().numpy(),
                s_sr=None if fused else s_sr.cpu().numpy(), s_gt=None if fused else s_gt.cpu().numpy(), edges=edges)


def _inputs(seed, B, H, W, dens):
    from ssl_amd import synth
    gt = np.stack([synth.natural_like(seed + i, H, W, 0.10, 0.04) for i in range(B)])
    sr = np.stack([synth.degrade(gt[i], seed + 100 + i, 0.05) for i in range(B)])
    rng = np.random.default_rng(seed)
    masks = (rng.random((B, H, W)) < dens).astype(np.float32)
    return sr.astype(np.float32), gt.astype(np.float32), masks


def _check_against_oracle(r, sr, gt, masks, ks, kw, sigma, cap=None):
    ref = orc.ssg_loss(sr.astype(np.float64), gt.astype(np.float64), masks, ks, kw, sigma, 1e3, 1e3)
    n = ref["n_edges"]
    assert r["counts"][0] == n == r["counts"][-1]
    want_pos = np.argwhere(masks > 0)
    assert np.array_equal(r["edges"][:n], want_pos)                      # torch.nonzero order (similaritywrapper.py:64-67)
    first = np.concatenate([[0], np.cumsum(masks.reshape(masks.shape[0], -1).sum(1))[:-1]]).astype(np.int64)
    assert np.array_equal(r["counts"][1:-1], first)
    if n == 0:
        assert not r["loss"].any() and not r["grad"].any()
        return
    assert abs(r["loss"][0] - ref["l1"]) <= 1e-5 * ref["l1"] and abs(r["loss"][1] - ref["kl"]) <= 1e-5 * ref["kl"] + 2e-8
    if r["s_sr"] is not None:
        assert maxerr(r["s_sr"][:n], ref["s_sr"]) <= 1e-5 and maxerr(r["s_gt"][:n], ref["s_gt"]) <= 1e-5
        assert (r["s_sr"][n:] == -3.0).all()                                # rows behind the last one untouched
        want, nflip = tp.ref_grad_with_gpu_signs(sr.astype(np.float64), masks, ks, kw, sigma, ref, r["s_sr"][:n], r["s_gt"][:n])
        assert nflip <= max(2, int(1e-4 * ref["s_sr"].size))
    else:
        want = ref["grad"]
    assert maxerr(r["grad"], want) <= tp.grad_tol_from_oracle(sr, gt, masks, ks, kw, sigma, ref)


@pytest.mark.parametrize("B,H,W,dens,sigma,det", [(1, 64, 64, 0.05, 1.0, True), (3, 48, 56, 0.08, 0.1, True),
                                                  (2, 33, 47, 0.3, 0.05, False), (4, 64, 64, 0.02, 0.3, True),
                                                  (1, 7, 9, 0.5, 0.2, True), (1, 128, 128, 0.1, 0.5, False)])
def test_tiny_step_vs_oracle(dev, B, H, W, dens, sigma, det):
    """Sizes from one heavily reflected 7 x 9 image (every search tile folds back twice) to the 16,384-pixel limit; a batch
    with an empty image; fp32 atomics and deterministic sums."""
    sr, gt, masks = _inputs(4000 + H, B, H, W, dens)
    if B == 3:
        masks[1] = 0
    cap = min(4096, B * H * W)
    r = _raw_step(dev, "ssg_loss_step", sr, gt, masks[:, None], 0, 1, 11, 5, sigma, cap, det, grad_fill=77.0)
    _check_against_oracle(r, sr, gt, masks, 11, 5, sigma)


def test_tiny_step_more_rows_than_workgroups_and_capacity_cut(dev):
    """1,640 rows on 1,024 workgroups (the row loop), then the same call under a capacity of 1,000: NaN losses, counts[0]
    still the number found, the first 1,000 rows exact."""
    sr, gt, masks = _inputs(5100, 1, 64, 64, 0.4)
    n = int(masks.sum())
    assert n > 1024
    r = _raw_step(dev, "ssg_loss_step", sr, gt, masks[:, None], 0, 1, 11, 5, 0.1, 4096, True)
    _check_against_oracle(r, sr, gt, masks, 11, 5, 0.1)
    c = _raw_step(dev, "ssg_loss_step", sr, gt, masks[:, None], 0, 1, 11, 5, 0.1, 1000, True)
    assert c["counts"][0] == n and np.isnan(c["loss"]).all()
    assert np.array_equal(c["s_sr"][:1000], r["s_sr"][:1000]) and np.array_equal(c["edges"], r["edges"][:1000])


@pytest.mark.parametrize("kind", ["u8", "f32x3", "laplacian", "stride"])
def test_tiny_builder_mask_kinds_equal_the_general_path(dev, kind):
    """uint8 masks, 3-channel float masks (channel 0 counts, loss_util.py:196), the Laplacian mask generated on the device
    (generate_mask.py:22-31) and the stride pattern (realesrganssl_model.py:64-72): rows, counts, SSG rows, losses and
    gradient of the two-launch path against the general path's on the same call."""
    from ssl_amd import synth
    B, H, W, ks, kw, sigma = 2, 64, 80, 11, 5, 0.2
    sr, gt, m = _inputs(6000, B, H, W, 0.06)
    stride, mask, mk, mc = 0, m[:, None], 0, 1
    if kind == "u8":
        mask, mk = m[:, None].astype(np.uint8), 1
    elif kind == "f32x3":
        mask, mc = np.concatenate([m[:, None], np.ones_like(m[:, None]), np.zeros_like(m[:, None])], 1), 3
    elif kind == "laplacian":
        mask, mk, mc = None, 2, 3
    elif kind == "stride":
        mask, stride = np.ones_like(m[:, None]), 4
    cap = 4096
    a = _raw_step(dev, "ssg_loss_step", sr, gt, mask, mk, mc, ks, kw, sigma, cap, True, stride=stride)
    with tiny_off():
        b = _raw_step(dev, "ssg_loss_step", sr, gt, mask, mk, mc, ks, kw, sigma, cap, True, stride=stride)
    n = int(a["counts"][0])
    assert n > 50 and np.array_equal(a["counts"], b["counts"]) and np.array_equal(a["edges"][:n], b["edges"][:n])
    if kind == "laplacian":
        want = np.stack([synth.laplacian_edge_mask(gt[i]) for i in range(B)])
        assert np.array_equal(a["edges"][:n], np.argwhere(want > 0))
    assert maxerr(a["s_sr"][:n], b["s_sr"][:n]) <= 2e-7 and maxerr(a["s_gt"][:n], b["s_gt"][:n]) <= 2e-7
    assert np.abs(a["loss"] - b["loss"]).max() <= 2e-6 * np.abs(b["loss"]).max()
    # (sign(s_sr - s_gt) of the L1 term may differ where the two paths round a near-tie differently: a handful of entries)
    d = np.abs(a["grad"] - b["grad"])
    assert d.max() <= 2e-2 * np.abs(b["grad"]).max() and (d > 1e-5 * np.abs(b["grad"]).max()).mean() <= 1e-3


def test_tiny_step_gradient_modes_fused_rows_and_reproducibility(dev):
    """ssg_loss_step (gradient assigned over garbage) == ssg_loss_fwd_bwd (accumulated into zeros) bit for bit in
    deterministic mode and over three runs; into a non-zero gradient it adds; without SSG output (fused) the same losses and
    gradient; without generalization (loss_util.py:226) against the oracle's un-normalised rows."""
    sr, gt, masks = _inputs(7000, 2, 40, 52, 0.1)
    ks, kw, sigma, cap = 11, 5, 0.15, 2 * 40 * 52
    args = (sr, gt, masks[:, None], 0, 1, ks, kw, sigma, cap)
    out = [_raw_step(dev, "ssg_loss_step", *args, True, grad_fill=123.0) for _ in range(3)]
    acc = _raw_step(dev, "ssg_loss_fwd_bwd", *args, True, grad_fill=0.0)
    for o in out[1:] + [acc]:
        for k in ("loss", "grad", "s_sr", "s_gt", "counts"):
            assert np.array_equal(out[0][k], o[k]), k
    add = _raw_step(dev, "ssg_loss_fwd_bwd", *args, True, grad_fill=2.0)
    assert np.abs((add["grad"] - 2.0) - out[0]["grad"]).max() <= 1e-6 * np.abs(out[0]["grad"]).max() + 5e-7
    fused = _raw_step(dev, "ssg_loss_step", *args, True, fused=True, grad_fill=9.0)
    assert np.array_equal(fused["loss"], out[0]["loss"]) and np.array_equal(fused["grad"], out[0]["grad"])
    atom = _raw_step(dev, "ssg_loss_step", *args, False, grad_fill=9.0)
    assert np.array_equal(atom["loss"], out[0]["loss"])
    assert np.abs(atom["grad"] - out[0]["grad"]).max() <= 2e-6 * np.abs(out[0]["grad"]).max()
    lossonly = _raw_step(dev, "ssg_loss_step", *args, False, want_grad=False)
    assert np.array_equal(lossonly["loss"], out[0]["loss"]) and np.array_equal(lossonly["s_sr"], out[0]["s_sr"])
    raw = _raw_step(dev, "ssg_loss_step", *args, True, gen=0)
    n = int(raw["counts"][0])
    pos = [orc.mask_to_pos(masks[i]) for i in range(2)]
    want = np.concatenate([orc.ssg_epilogue(orc.distance(sr[i].astype(np.float64), pos[i], ks, kw), kw, 3, sigma, False) for i in range(2)])
    assert maxerr(raw["s_sr"][:n], want) <= 1e-5
    with tiny_off():
        rawb = _raw_step(dev, "ssg_loss_step", *args, True, gen=0)
    assert np.abs(raw["loss"] - rawb["loss"]).max() <= 2e-6 * np.abs(rawb["loss"]).max()
    assert np.abs(raw["grad"] - rawb["grad"]).max() <= 2e-2 * np.abs(rawb["grad"]).max()


def test_tiny_step_replays_as_hip_graph(dev):
    """Two launches, nothing host-side in them: a recorded step reproduces the eager one bit for bit (deterministic sums),
    also after the content at the recorded addresses changed."""
    from ssl_amd import engine
    sr_np, gt_np, m_np = _inputs(8000, 1, 64, 64, 0.05)
    sr, gt, mask = T(sr_np, dev), T(gt_np, dev), T(m_np[:, None], dev)
    eager = engine.LossStep(1, 3, 64, 64, 11, 5, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=1500, deterministic=True)
    graph = engine.LossStep(1, 3, 64, 64, 11, 5, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=1500, deterministic=True, graph=True)
    for seed in (8000, 8000, 8100):
        s2, g2, m2 = _inputs(seed, 1, 64, 64, 0.05 if seed == 8000 else 0.2)
        sr.copy_(T(s2, dev)); gt.copy_(T(g2, dev)); mask.copy_(T(m2[:, None], dev))
        l0, g0 = eager(sr, gt, mask)
        l1, g1 = graph(sr, gt, mask)
        n = int(eager.counts[0])
        assert n == int(m2.sum()) == int(graph.counts[0])
        assert torch.equal(l0, l1) and torch.equal(g0, g1) and torch.equal(eager.ssg_sr[:n], graph.ssg_sr[:n])


def test_general_11_5_path_stays_covered(dev, golden):
    """The (11,5) parity cases of test_gpu_parity.py with the two-launch path switched off: the direct kernels of that size
    are what larger (11,5) calls and the separate entry points run."""
    with tiny_off():
        tp.test_loss_step_matches_oracle_small_batch(dev)
        tp.test_f1_full_loss_step_golden(dev, golden)
    sr, gt, masks = _inputs(9000, 1, 160, 160, 0.02)                     # 25,600 pixels: beyond the builder's limit anyway
    r = _raw_step(dev, "ssg_loss_step", sr, gt, masks[:, None], 0, 1, 11, 5, 0.3, 2000, True)
    _check_against_oracle(r, sr, gt, masks, 11, 5, 0.3)


def test_tiny_step_fold_sees_every_workgroups_sums(dev):
    """Regression (round 6, found by tools/r6_fuzz_tiny.py): with a relaxed ticket behind a bare s_waitcnt the last workgroup
    folded fixed-point sums that lacked other workgroups' last additions -- once in a few hundred steps of a tiny, fully
    masked call, where sixty workgroups queue on a few hundred addresses.  Deterministic mode promises the same bits every
    run: 400 steps of a 2 x 10 x 6 call (and of 1 x 6 x 6, every search tile folded twice) equal the first, and the general path."""
    from ssl_amd import engine
    for (B, H, W) in ((2, 10, 6), (1, 6, 6)):
        sr, gt, m = _inputs(9100 + H, B, H, W, 0.6)
        srt, gtt, mt = T(sr, dev), T(gt, dev), T(m[:, None], dev)
        step = engine.LossStep(B, 3, H, W, 11, 5, 0.004, 1e-10, True, 1e3, 1e3, device=dev, deterministic=True)
        loss, grad = step(srt, gtt, mt)
        l0, g0 = loss.clone(), grad.clone()
        for _ in range(400):
            loss, grad = step(srt, gtt, mt)
            assert torch.equal(grad, g0) and torch.equal(loss, l0)
        with tiny_off():
            ref = engine.LossStep(B, 3, H, W, 11, 5, 0.004, 1e-10, True, 1e3, 1e3, device=dev, deterministic=True)
            lr, gr = ref(srt, gtt, mt)
            assert float((gr - g0).abs().max()) <= 2e-6 * float(gr.abs().max())
