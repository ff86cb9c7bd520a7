"""`similarity_map` -- the reference's loss-module API, on MI355X.

Mirror of GAN-Based-SR/basicsr/losses/loss_util.py:165-248: same constructor
signature, eager computation in the constructor, `.getitem()` returning the
SSG tensor (1, N, k_s^2) that is differentiable w.r.t. `img`, `ValueError` for
an unknown `ssl_mode`.  All modes run the hand-written HIP engine:

  'hip'      one fused launch: search tile -> distances -> exp -> row
             normalisation (ssg_map_forward); edge list built on device.
  'cuda'     the reference's operator decomposition (loss_util.py:231-244):
             compute_similarity() raw distances from the C-ABI operator, then
             the five element-wise steps in PyTorch.  Uses mask[0,0] only.
  'pytorch'  the engine with ssl_pytorch's conventions (loss_util.py:182-229):
             a (1,3,H,W) mask lists every edge pixel three times
             (block-tiled), because torch.where runs over all mask channels.

`eps` (default 1e-10, loss_util.py:227,242) is exposed because the
Diffusion-Based-SR fork uses 1e-20 / 1e-6 for the same expression.
"""
import math

import torch

from . import similarity as _similarity  # noqa: F401  (package marker)
from .similarity.similaritywrapper import compute_similarity
from .. import engine


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
        elif ssl_mode == 'hip':
            self.ssl_hip(img=img, mask=mask, kernel_size_search=kernel_size_search,
                         kernel_size_window=kernel_size_window, sigma=sigma, generalization=generalization)
        else:
            raise ValueError(f"The ssl_mode should either be cuda or pytorch.")

    def ssl_hip(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
        # img 1*C*h*w, mask 1*c1*h*w (channel 0 decides)
        if img.shape[0] != 1:
            raise ValueError("similarity_map takes one image at a time (1,C,H,W); use ssl_amd.SSGLoss for batches")
        eps = getattr(self, "eps", 1e-10)
        el = engine.edge_list(mask=mask, capacity=img.shape[-1] * img.shape[-2], ks=kernel_size_search)
        num = int(el.counts[0].item())    # the reference synchronises here too (torch.where / nonzero)
        s = engine.ssg_map(img, el.edges, el.counts, num, kernel_size_search, kernel_size_window, sigma, eps,
                           generalization, order=el.order, fwd=el.fwd)
        self.s = s.unsqueeze(0)           # 1, num, k_s*k_s

    def ssl_pytorch(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
        # torch.where over a c1-channel mask (loss_util.py:195-198) lists the edge pixels of channel 0, then those
        # of channel 1, ...: one block of rows per channel.  The pair pool produces channels that are copies of
        # each other (realesrganssl_model.py:339-341), i.e. the same block c1 times; channels that differ get
        # their own edge lists.
        _, c1, _, _ = mask.shape
        blocks = []
        for c in range(c1):
            if c > 0 and torch.equal(mask[:, c], mask[:, 0]):
                blocks.append(blocks[0])
                continue
            self.ssl_hip(img, mask[:, c:c + 1], kernel_size_search, kernel_size_window, sigma, generalization)
            blocks.append(self.s)
        self.s = blocks[0] if c1 == 1 else torch.cat(blocks, dim=1)

    def ssl_cuda(self, img, mask, kernel_size_search=25, kernel_size_window=9, sigma=1.0, generalization=False):
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
