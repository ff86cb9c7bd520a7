"""`similarity_map` / `trainable_similarity_map` with the Diffusion-Based-SR fork's constructor
(StableSR / LDM-SR + SSL).

Mirror of Diffusion-Based-SR/basicsr/losses/loss_util.py:242-363 for every strategy that is
built on the native operator `compute_similarity`.  The configured
one, `areaarea_mask_nonlocalavg_cuda_v1` (configs/StableSRISSLStage1/*.yml:32-41, code
loss_util.py:1239-1252), runs the fused HIP launch with the fork's epsilon 1e-20; the others
are cheap epilogues on the same distance kernel (the C-ABI operator), each a row of this table:

  strategy (areaarea_mask_...)  distance scaling     row weight                 normaliser      fork lines
  nonlocal_cuda_v1              D                    exp(-q/s)                  /(sum+1e-6)     :1180-1191
  nonlocal                      (unfold twin of the above)                                       :725-777
  nonlocal_cuda_v1_patch        D per dh x dw patch  exp(-q/s)                  /(sum+1e-6)     :1193-1225
  nonlocal_cuda_v2              sqrt(D+1e-8)         exp(-q/s)                  /(sum+1e-6)     :1227-1237
  nonlocalavg_cuda_v1           D/(C kc^2)           exp(-q/s)                  /(sum+1e-20)    :1239-1252
  nonlocalavg_cuda_v2           D/(3 kc^2)           exp(-q/s), centre removed  /(sum+1e-6)     :1254-1270
  eulardistanceavg_cuda_v1      D/(3 kc^2)/s         centre removed             softmax(-q)     :1272-1290
  nonlocalavg_cuda_v3           D/(3 kc^2)           exp(-q/s) * mean_row(exp)  /(sum+1e-6)     :1292-1309
  nonlocalavg_cuda_v4           D/(3 k^2), k in kc[] exp(-q/s), max over k      /(sum+1e-10)    :1311-1331
  nonlocalavg_cuda_v1RGB        D_ch/kc^2 per channel, rows concatenated        /(sum+1e-10)    :1333-1350
  nonlocalavg_cuda_v5           D/(C kc^2)           exp(-q/s), sum|softmax, top-k              :1352-1373
  nonlocalavg_cuda_maxh_v1      D/(C kc^2) for GT and SR, shared adaptive bandwidth /(sum+1e-20) :1375-1399

`trainable_similarity_map` (:1448-1481) is the learnable-bandwidth module.  The fork's ~20 other
`simself_*` experiments do not touch the operator and stay out of scope (NotImplementedError).

Call sites: ldm/models/diffusion/ddpmssl.py:438-513 (`issl`), which passes simself_strategy,
kernel_size, scaling_factor, softmax, kernel_size_center and a dozen arguments the
operator-based strategies ignore.
"""
import math

import torch
import torch.nn as nn

from .loss_util import similarity_map as _gan_similarity_map
from .similarity.similaritywrapper import compute_similarity

_P = 'areaarea_mask_'
_OPERATOR_STRATEGIES = tuple(_P + s for s in (
    'nonlocal', 'nonlocal_cuda_v1', 'nonlocal_cuda_v1_patch', 'nonlocal_cuda_v2', 'nonlocalavg_cuda_v1',
    'nonlocalavg_cuda_v2', 'eulardistanceavg_cuda_v1', 'nonlocalavg_cuda_v3', 'nonlocalavg_cuda_v4',
    'nonlocalavg_cuda_v1RGB', 'nonlocalavg_cuda_v5', 'nonlocalavg_cuda_maxh_v1'))


def _distances(image, mask2d, ks, kc, scale=1.0):
    """(1, N, ks*ks) squared window distances of one image, divided by `scale`."""
    q = compute_similarity(image=image, mask=mask2d, psize=ks, ksize=kc)
    if scale != 1.0:
        q = q / scale
    return q.reshape(1, q.shape[0], ks * ks)


def _row_normalise(q, eps):
    return 1 / (torch.sum(q, dim=-1) + eps).unsqueeze(-1) * q


def _drop_centre(q, ks):
    c = ks * ks // 2
    return torch.cat([q[:, :, :c], q[:, :, c + 1:]], dim=-1)


class similarity_map():
    def __init__(self, img, mask=None, img_sr=None, simself_strategy='imgimg', is_shift=False, shift_h=16, shift_w=16,
                 dh=32, dw=32, gray=False, threshold=2e-3, kernel_size=5, scaling_factor=4, softmax=True,
                 rearrange_back=True, crossentropy=False, temperature=0, stride=1, pix_num=1, index=None,
                 kernel_size_center=9, mean=False, var=False, largest_k=0, gene_type="sum"):
        super(similarity_map, self).__init__()
        if simself_strategy not in _OPERATOR_STRATEGIES:
            raise NotImplementedError(
                f"simself_strategy '{simself_strategy}' is one of the fork's experiments that do not use the "
                f"native operator; ssl_amd provides {_OPERATOR_STRATEGIES}")
        name = simself_strategy[len(_P):]
        ks, kc, sigma = kernel_size, kernel_size_center, scaling_factor
        c = img.shape[1]
        if name == 'nonlocalavg_cuda_v1':
            # one fused launch; `softmax` is this fork's name for the row normalisation
            self.s = _gan_similarity_map(img, mask, ssl_mode='hip', kernel_size_search=ks, generalization=softmax,
                                         kernel_size_window=kc, sigma=sigma, eps=1e-20).getitem()
        elif name in ('nonlocal', 'nonlocal_cuda_v1', 'nonlocal_cuda_v2'):
            q = _distances(img[0], mask[0, 0], ks, kc)
            if name == 'nonlocal_cuda_v2':
                q = torch.sqrt(q + 1e-8)
            q = torch.exp(-1 * q / sigma)
            self.s = _row_normalise(q, 1e-6) if softmax else q
        elif name == 'nonlocal_cuda_v1_patch':
            self.s = self._per_patch(img, mask, ks, kc, sigma, softmax, dh, dw)
        elif name == 'nonlocalavg_cuda_v2':
            q = _drop_centre(torch.exp(-1 * _distances(img[0], mask[0, 0], ks, kc, 3 * math.pow(kc, 2)) / sigma), ks)
            self.s = _row_normalise(q, 1e-6) if softmax else q
        elif name == 'eulardistanceavg_cuda_v1':
            q = _drop_centre(_distances(img[0], mask[0, 0], ks, kc, 3 * math.pow(kc, 2)) / sigma, ks)
            self.s = (-1 * q).softmax(dim=-1) if softmax else q
        elif name == 'nonlocalavg_cuda_v3':
            q = torch.exp(-1 * _distances(img[0], mask[0, 0], ks, kc, 3 * math.pow(kc, 2)) / sigma)
            q = q.sum(dim=-1, keepdim=True) / math.pow(ks, 2) * q
            self.s = _row_normalise(q, 1e-6) if softmax else q
        elif name == 'nonlocalavg_cuda_v4':
            maps = []
            for k in kc:
                q = torch.exp(-1 * _distances(img[0], mask[0, 0], ks, k, 3 * math.pow(k, 2)) / sigma)
                maps.append(_row_normalise(q, 1e-10) if softmax else q)
            self.s = torch.max(torch.stack(maps, dim=3), dim=3)[0]
        elif name == 'nonlocalavg_cuda_v1RGB':
            rows = []
            for i in range(c):
                q = torch.exp(-1 * _distances(img[0, i].unsqueeze(0), mask[0, i], ks, kc, math.pow(kc, 2)) / sigma)
                rows.append(_row_normalise(q, 1e-10) if softmax else q)
            self.s = torch.cat(rows, dim=1)
        elif name == 'nonlocalavg_cuda_v5':
            q = torch.exp(-1 * _distances(img[0], mask[0, 0], ks, kc, c * math.pow(kc, 2)) / sigma)
            if softmax:
                if gene_type == 'sum':
                    q = _row_normalise(q, 1e-10)
                elif gene_type == 'softmax':
                    q = q.softmax(dim=-1)
            if largest_k > 0:
                q = torch.sort(q, dim=-1, descending=True)[0][:, :, :largest_k]
            self.s = q
        elif name == 'nonlocalavg_cuda_maxh_v1':
            # img is the GT; both maps share a per-entry bandwidth derived from the two distances
            q_gt = _distances(img[0], mask[0, 0], ks, kc, c * math.pow(kc, 2))
            q_sr = _distances(img_sr[0], mask[0, 0], ks, kc, c * math.pow(kc, 2))
            max_h = (q_gt - q_sr + 1e-20) / (((q_gt.pow(2) + 1e-20) / (q_sr.pow(2) + 1e-20)).log() + 1e-20)
            q_gt, q_sr = torch.exp(-1 * q_gt / max_h), torch.exp(-1 * q_sr / max_h)
            if softmax:
                q_gt, q_sr = _row_normalise(q_gt, 1e-20), _row_normalise(q_sr, 1e-20)
            self.s, self.s1 = q_gt, q_sr

    @staticmethod
    def _per_patch(img, mask, ks, kc, sigma, softmax, dh, dw):
        """Non-overlapping dh x dw patches, each its own search domain (reflect-padded on its own border);
        patches are visited row-major, empty ones skipped (loss_util.py:1193-1225)."""
        b, c, h, w = img.shape
        rows = []
        for py in range(h // dh):
            for px in range(w // dw):
                sub = img[0, :, py * dh:(py + 1) * dh, px * dw:(px + 1) * dw]
                m = mask[0, 0, py * dh:(py + 1) * dh, px * dw:(px + 1) * dw]
                if m.sum() != 0:
                    q = torch.exp(-1 * _distances(sub, m, ks, kc) / sigma)
                    rows.append(_row_normalise(q, 1e-6) if softmax else q)
        return torch.cat(rows, dim=1)

    def getitem(self):
        return self.s

    def getitem_simmutual(self):
        return self.s, self.s1


class trainable_similarity_map(nn.Module):
    """Learnable bandwidth (loss_util.py:1448-1481): s = exp(-q/relu(sigma) + 1e-20), rows / (sum + 1e-20).
    forward(img, img_sr, mask) -> (s_img, s_sr); the gradient reaches `sigma` through the epilogue and both
    images through the operator."""

    def __init__(self, scaling_factor=4):
        super(trainable_similarity_map, self).__init__()
        self.sigma = nn.Parameter(torch.tensor([float(scaling_factor)]), requires_grad=True)

    def _one(self, image, mask, ks, kc, softmax):
        c = image.shape[1]
        q = _distances(image[0], mask[0, 0], ks, kc, c * math.pow(kc, 2))
        q = torch.exp(-1 * q / torch.relu(self.sigma) + 1e-20)
        return _row_normalise(q, 1e-20) if softmax else q

    def forward(self, img, img_sr, mask, kernel_size_search=25, kernel_size_center=9, softmax=False):
        return (self._one(img, mask, kernel_size_search, kernel_size_center, softmax),
                self._one(img_sr, mask, kernel_size_search, kernel_size_center, softmax))

    def getitem_h(self):
        return self.sigma.item()
