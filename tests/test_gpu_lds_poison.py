"""LDS-poison audit inside the DEFAULT `-m gpu` run (round 6, review item 7).

LDS keeps what the previous workgroup on a CU left in it, so a kernel that reads a word it has not written usually sees
plausible stale values and fails once in a blue moon: `band_scatter`'s merge flags did exactly that for a whole round
under a green suite (round 5; similarity.cu has no counterpart -- the edge list replaces similaritywrapper.py:64's
torch.nonzero).  The profiling build of the library (libssg_hip_prof.so, -DSSG_PROFILE) can run a kernel that fills the
LDS of every CU with a word in front of every launch (include/ssg_hip.h: ssg_prof_set_lds_poison).  tools/r5_poison_suite.sh
runs the whole suite that way on request; THIS file makes the cases of the bug class -- the banded edge-list builder,
its plan and merge flags, and the merged / dense forward + backward that consume them -- part of every run, under two
words: a NaN pattern (any arithmetic use shows) and a small-integer pattern (any index / flag use shows).

The checks are the parity tests' own (numpy restatement of the builder; the product library's un-poisoned step, which
the other tests pin against the oracle and the reference's fixtures) -- nothing here goes through oracle/.
"""
import numpy as np
import pytest
import torch

import test_gpu_parity as tp

PATTERNS = [0x7FC00000, 0x00640064]   # quiet NaN; 100 | 100 << 16 (a plausible count / flag / lane index)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class poisoned:
    """Inside the block the engine's host code runs on the profiling build with the LDS poison on."""

    def __init__(self, word):
        self.word = word

    def __enter__(self):
        from ssl_amd import _lib
        self.ctx = _lib.profile_build()
        self.L = self.ctx.__enter__()
        self.L.ssg_prof_set_lds_poison(1, self.word)
        return self.L

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        self.L.ssg_prof_set_lds_poison(0, 0)
        return self.ctx.__exit__(*exc)


@pytest.mark.gpu
@pytest.mark.parametrize("word", PATTERNS)
def test_banded_builder_under_lds_poison(dev, word):
    """The builder cases that hold every tile class, ragged bands, several segments, the stride pattern, a capacity cut,
    and the round-5 regression (groups reaching into the next band), with every CU's LDS poisoned before every launch."""
    with poisoned(word):
        tp.test_banded_edge_list_builder_vs_numpy(dev, 3, 61, 83, 0.30, 0, None, "u8")
        tp.test_banded_edge_list_builder_vs_numpy(dev, 2, 72, 300, 0.50, 3, None, "f32")
        tp.test_banded_edge_list_builder_vs_numpy(dev, 2, 64, 288, 0.25, 0, 0.55, "f32")
        tp.test_banded_builder_groups_reaching_into_the_next_band_stay_unflagged(dev)
        tp.test_banded_builder_laplacian_mask_on_the_fly(dev, 1, 40, 600, 0)


def _clustered_mask(B, H, W, seed):
    """Edge pixels in thin columns (groups of five that merge, some across two bands), a dense stripe (dense tiles next
    to them) and a sprinkle of isolated pixels (single-job groups): every forward / backward variant gets rows."""
    rng = np.random.default_rng(seed)
    m = rng.random((B, H, W)) < 0.01
    for b in range(B):
        for band in range(H // 8):
            m[b, band * 8:band * 8 + 8, 40:52] |= rng.random((8, 12)) < rng.uniform(0.03, 0.15)
    m[:, :, W - 32:] |= rng.random((B, H, 32)) < 0.3
    m[:, 16:24, 64:96] |= rng.random((B, 8, 32)) < 0.7      # a heavy / huge tile
    return m


@pytest.mark.gpu
@pytest.mark.parametrize("word", PATTERNS)
def test_merged_forward_and_backward_under_lds_poison(dev, word):
    """One fused loss step (25,9) whose plan holds mergeable groups, groups across bands, light / heavy / huge dense tiles:
    SSG rows, both losses and the gradient under LDS poison (profiling build) equal the product library's to 1e-6 of
    their maximum (a poisoned read shows as NaN or as values of order 100)."""
    from ssl_amd import engine, synth
    B, H, W = 2, 96, 160
    gt = np.stack([synth.natural_like(7000 + i, H, W, 0.10, 0.05) for i in range(B)])
    sr = np.stack([synth.degrade(gt[i], 17_000 + i) for i in range(B)])
    m = _clustered_mask(B, H, W, 31)
    mk = torch.as_tensor(m[:, None].astype(np.float32), device=dev)
    srt, gtt = tp.T(sr, dev), tp.T(gt, dev)
    thr = engine.set_dense_threshold(20)
    try:
        def run():
            step = engine.LossStep(B, 3, H, W, 25, 9, 1.0, 1e-10, True, 1e3, 1e3, device=dev, capacity=int(m.sum()) + 64)
            loss, grad = step(srt, gtt, mk)
            torch.cuda.synchronize()
            n = int(step.counts[0])
            return (loss.clone(), grad.clone(), step.ssg_sr[:n].clone(), step.ssg_gt[:n].clone(), n)
        want = run()
        with poisoned(word):
            got = run()
        assert got[4] == want[4] == int(m.sum())
        # (the profiling build is the same source compiled with -DSSG_PROFILE: its kernels may contract / order a few
        # operations differently, last-bit differences -- not what a poisoned read looks like: NaN, or values of order 100)
        for name, a, b, tol in zip(("loss", "grad", "ssg_sr", "ssg_gt"), got, want, (1e-6, 1e-6, 1e-6, 1e-6)):
            assert bool(torch.isfinite(a).all()), name
            err, ref = float((a - b).abs().max()), float(b.abs().max())
            assert err <= tol * ref, (name, err, ref)
    finally:
        engine.set_dense_threshold(thr)


@pytest.mark.gpu
@pytest.mark.parametrize("word", PATTERNS)
def test_two_launch_small_step_under_lds_poison(dev, word):
    """ssg_tiny.hip keeps tiles, rows, G, the staged gradient and six kinds of partial sums in LDS and runs rows in a loop
    (1,640 rows on 1,024 workgroups): its oracle cases under LDS poison."""
    import test_gpu_tiny as tt
    with poisoned(word):
        tt.test_tiny_step_vs_oracle(dev, 3, 48, 56, 0.08, 0.1, True)
        tt.test_tiny_step_vs_oracle(dev, 1, 7, 9, 0.5, 0.2, True)
        tt.test_tiny_step_more_rows_than_workgroups_and_capacity_cut(dev)
