"""`similarity_map` -- the reference's loss-module API, on MI355X.

Mirror of GAN-Based-SR/basicsr/losses/loss_util.py:165-248: same constructor
signature, `.getitem()` returning the SSG tensor (1, N, k_s^2) that is
differentiable w.r.t. `img` (or the deferred handle of it, below), `ValueError`
for an unknown `ssl_mode`.  All modes run the hand-written HIP engine:

  'hip'      one fused launch: search tile -> distances -> exp -> row
             normalisation (ssg_map_forward); edge list built on device.
  'cuda'     ssl_cuda's convention (loss_util.py:231-244: mask[0,0] only) on the
             fused launch; 'cuda_operator' (not a reference mode) is the literal
             decomposition: compute_similarity() raw distances from the C-ABI
             operator, then the five element-wise steps in PyTorch.
  'pytorch'  the engine with ssl_pytorch's conventions (loss_util.py:182-229):
             a (1,3,H,W) mask lists every edge pixel three times
             (block-tiled), because torch.where runs over all mask channels.

`eps` (default 1e-10, loss_util.py:227,242) is exposed because the
Diffusion-Based-SR fork uses 1e-20 / 1e-6 for the same expression.

Deferred evaluation (default; SSG_LAZY=0 or `set_lazy(False)` turn it off): `.getitem()` returns a
`LazySSG` handle -- a tensor-like that computes nothing until it is used.  The reference's unchanged
caller loop (per-image constructions, `torch.cat`, `L1Loss`, `KLDistanceLoss`) then runs as ONE batched
step over all images; any other use of the handle computes the rows eagerly, per image, with the values
and row conventions below.  See ssl_amd/losses/lazy.py.
"""
import math

import torch

from . import similarity as _similarity  # noqa: F401  (package marker)
from .similarity.similaritywrapper import compute_similarity
from .. import engine
from .lazy import LazySSG, lazy_enabled, set_lazy  # noqa: F401


_PLAN_FROM_ROWS = 10000   # edge pixels of ONE image from which its eager rows go through the dense/direct plan


def eager_rows(img, mask, conv, kernel_size_search, kernel_size_window, sigma, eps, generalization):
    """(1, N, k_s^2) SSG rows of ONE image (1,C,H,W), computed now: device-side edge list, one host read of the row
    count (the reference synchronises there too: torch.where / nonzero), one fused forward launch.
    conv 'ch0': channel 0 of the mask decides (ssl_cuda's mask[0,0], loss_util.py:233);
    conv 'all': torch.where over a c1-channel mask (ssl_pytorch, loss_util.py:195-198) lists the edge pixels of
    channel 0, then those of channel 1, ...: one block of rows per channel.  The pair pool produces channels that are
    copies of each other (realesrganssl_model.py:339-341), i.e. the same block c1 times; channels that differ get
    their own edge lists."""
    if img.shape[0] != 1:
        raise ValueError("similarity_map takes one image at a time (1,C,H,W); use ssl_amd.SSGLoss for batches")
    if conv == 'all' and mask.shape[1] > 1:
        blocks = []
        for c in range(mask.shape[1]):
            if c > 0 and torch.equal(mask[:, c], mask[:, 0]):
                blocks.append(blocks[0])
            else:
                blocks.append(eager_rows(img, mask[:, c:c + 1], 'ch0', kernel_size_search, kernel_size_window, sigma,
                                         eps, generalization))
        return torch.cat(blocks, dim=1)
    cap = img.shape[-1] * img.shape[-2]
    el = engine.edge_list(mask=mask, capacity=cap, ks=kernel_size_search, plan=cap >= 4 * _PLAN_FROM_ROWS)
    num = int(el.counts[0].item())
    if el.fwd is None and num >= _PLAN_FROM_ROWS:     # (a small image with an unusually dense mask: now with the plan)
        el = engine.edge_list(mask=mask, capacity=cap, ks=kernel_size_search)
    # One image rarely has enough dense tiles to fill the chip: its shared-term launch is one resident round (~85 us)
    # whatever the tile count, longer than the direct kernels need for the whole image below ~10 k edge pixels
    # (profiles/r4_operator_vs_plan.txt: 4,820 px 0.086 vs 0.117 ms, 18,417 px 0.187 vs 0.144) -- the plan is used from there.
    s = engine.ssg_map(img, el.edges, el.counts, num, kernel_size_search, kernel_size_window, sigma, eps,
                       generalization, order=el.order, fwd=el.fwd if num >= _PLAN_FROM_ROWS else None)
    return s.unsqueeze(0)           # 1, num, k_s*k_s


class similarity_map():
    def __init__(self, img, mask=None, ssl_mode='cuda', kernel_size_search=5, generalization=True,
                 kernel_size_window=9, sigma=0.004, eps=1e-10):
        super(similarity_map, self).__init__()
        self.eps = eps
        if ssl_mode == 'pytorch':
            self.ssl_pytorch(img=img, mask=mask, kernel_size_search=kernel_size_search,
                             kernel_size_window=kernel_size_window, sigma=sigma, generalization=generalization)
        elif ssl_mode == 'cuda':
            self.ssl_cuda(img=img, mask=mask, kernel_size_search=kernel_size_search,
                          kernel_size_window=kernel_size_window, sigma=sigma, generalization=generalization)
        elif ssl_mode == 'cuda_operator':
            self.ssl_cuda_operator(img=img, mask=mask, kernel_size_search=kernel_size_search,
                                   kernel_size_window=kernel_size_window, sigma=sigma, generalization=generalization)
        elif ssl_mode == 'hip':
            self.ssl_hip(img=img, mask=mask, kernel_size_search=kernel_size_search,
                         kernel_size_window=kernel_size_window, sigma=sigma, generalization=generalization)
        else:
            raise ValueError(f"The ssl_mode should either be cuda or pytorch.")

    def _rows(self, img, mask, conv, kernel_size_search, kernel_size_window, sigma, generalization):
        engine._need_gpu(img, mask)
        cfg = (kernel_size_search, kernel_size_window, sigma, getattr(self, "eps", 1e-10), generalization)
        if lazy_enabled():
            if img.shape[0] != 1:
                raise ValueError("similarity_map takes one image at a time (1,C,H,W); use ssl_amd.SSGLoss for batches")
            if img.shape[-2] <= kernel_size_search // 2 or img.shape[-1] <= kernel_size_search // 2:
                # (what the eager launch reports as SSG_E_IMAGESMALL, and F.pad(mode='reflect') in the reference)
                raise RuntimeError(f"ssl_amd: image {tuple(img.shape[-2:])} is not larger than k_s/2 = "
                                   f"{kernel_size_search // 2}: reflect padding is undefined")
            self.s = LazySSG([(img, mask, conv)], cfg)
        else:
            self.s = eager_rows(img, mask, conv, *cfg)

    def ssl_hip(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
        # img 1*C*h*w, mask 1*c1*h*w (channel 0 decides)
        self._rows(img, mask, 'ch0', kernel_size_search, kernel_size_window, sigma, generalization)

    def ssl_pytorch(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
        # torch.where over every mask channel: one block of rows per channel (see eager_rows)
        self._rows(img, mask, 'all', kernel_size_search, kernel_size_window, sigma, generalization)

    def ssl_cuda(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
        # The reference decomposes this mode into compute_similarity(image=img[0], mask=mask[0,0], ...) and five
        # element-wise steps (loss_util.py:231-244).  Here the mode keeps its convention -- channel 0 of the mask
        # decides -- and runs the fused forward: distances, /(C k_w^2), exp(-q/sigma) and the row normalisation in one
        # launch.  The operator itself stays available as ssl_amd.compute_similarity; `ssl_cuda_operator` below is the
        # literal decomposition on it (tests pin both to the same fixtures).
        self._rows(img, mask, 'ch0', kernel_size_search, kernel_size_window, sigma, generalization)

    def ssl_cuda_operator(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
        b, c, h, w = img.shape
        eps = getattr(self, "eps", 1e-10)
        q = compute_similarity(image=img[0], mask=mask[0, 0], psize=kernel_size_search, ksize=kernel_size_window)
        q = q / (c * math.pow(kernel_size_window, 2))
        q = q.unsqueeze(0)
        b, num, _, _ = q.shape
        q = q.reshape(b, num, kernel_size_search * kernel_size_search)
        q = torch.exp(-1 * q / sigma)
        if generalization:
            q = 1 / (torch.sum(q, dim=-1) + eps).unsqueeze(-1) * q
        self.s = q

    def getitem(self):
        return self.s
