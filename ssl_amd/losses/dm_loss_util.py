"""`similarity_map` with the Diffusion-Based-SR fork's constructor (StableSR / LDM-SR + SSL).

Mirror of Diffusion-Based-SR/basicsr/losses/loss_util.py:242-363 for the strategies that are
built on the native operator (the fork's ~30 other `simself_*` experiments are not configured
anywhere and stay out of scope).  The configured one,
`areaarea_mask_nonlocalavg_cuda_v1` (configs/StableSRISSLStage1/*.yml:32-41, code
loss_util.py:1239-1252), runs the fused HIP launch with the fork's epsilon 1e-20; the others
apply their PyTorch epilogue to `compute_similarity` (the C-ABI operator), line by line as the
fork writes them:

  areaarea_mask_nonlocal_cuda_v1      exp(-D/s)                    / (sum + 1e-6)   :1180-1191
  areaarea_mask_nonlocal              (unfold twin of the above)                     :725-777
  areaarea_mask_nonlocal_cuda_v2      exp(-sqrt(D + 1e-8)/s)       / (sum + 1e-6)   :1227-1237
  areaarea_mask_nonlocalavg_cuda_v1   exp(-(D/(C kc^2))/s)         / (sum + 1e-20)  :1239-1252
  areaarea_mask_nonlocalavg_cuda_v2   same, centre offset removed  / (sum + 1e-6)   :1253-1271

Call sites: ldm/models/diffusion/ddpmssl.py:438-513 (`issl`), which passes
simself_strategy, kernel_size, scaling_factor, softmax, kernel_size_center and a dozen
arguments the operator-based strategies ignore.
"""
import math

import torch

from .loss_util import similarity_map as _gan_similarity_map
from .similarity.similaritywrapper import compute_similarity

_OPERATOR_STRATEGIES = ('areaarea_mask_nonlocal', 'areaarea_mask_nonlocal_cuda_v1', 'areaarea_mask_nonlocal_cuda_v2',
                        'areaarea_mask_nonlocalavg_cuda_v1', 'areaarea_mask_nonlocalavg_cuda_v2')


class similarity_map():
    def __init__(self, img, mask=None, img_sr=None, simself_strategy='imgimg', is_shift=False, shift_h=16, shift_w=16,
                 dh=32, dw=32, gray=False, threshold=2e-3, kernel_size=5, scaling_factor=4, softmax=True,
                 rearrange_back=True, crossentropy=False, temperature=0, stride=1, pix_num=1, index=None,
                 kernel_size_center=9, mean=False, var=False, largest_k=0, gene_type="sum"):
        super(similarity_map, self).__init__()
        if simself_strategy not in _OPERATOR_STRATEGIES:
            raise NotImplementedError(
                f"simself_strategy '{simself_strategy}' is one of the fork's unconfigured experiments; "
                f"ssl_amd provides {_OPERATOR_STRATEGIES}")
        ks, kc, sigma = kernel_size, kernel_size_center, scaling_factor
        if simself_strategy == 'areaarea_mask_nonlocalavg_cuda_v1':
            # one fused launch; `softmax` is this fork's name for the row normalisation
            self.s = _gan_similarity_map(img, mask, ssl_mode='hip', kernel_size_search=ks, generalization=softmax,
                                         kernel_size_window=kc, sigma=sigma, eps=1e-20).getitem()
            return
        b, c, h, w = img.shape
        q = compute_similarity(image=img[0], mask=mask[0, 0], psize=ks, ksize=kc)
        if simself_strategy == 'areaarea_mask_nonlocalavg_cuda_v2':
            q = q / (3 * math.pow(kc, 2))
        q = q.unsqueeze(0)
        b, num, _, _ = q.shape
        q = q.reshape(b, num, ks * ks)
        if simself_strategy == 'areaarea_mask_nonlocal_cuda_v2':
            q = torch.sqrt(q + 1e-8)
        q = torch.exp(-1 * q / sigma)
        if simself_strategy == 'areaarea_mask_nonlocalavg_cuda_v2':
            q_1 = q[:, :, : ks * ks // 2]
            q_2 = q[:, :, ks * ks // 2 + 1:]
            q = torch.cat([q_1, q_2], dim=-1)
        if softmax:
            q = 1 / (torch.sum(q, dim=-1) + 1e-6).unsqueeze(-1) * q
        self.s = q

    def getitem(self):
        return self.s
