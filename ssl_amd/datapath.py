"""The step before the SSG loss, on the GPU (minimal slice).

`augment_crop`  = the dataset's joint flip / rot90 of GT and mask (GAN-Based-SR/basicsr/data/transforms.py:152-219
                  `augment`) + the joint random crop of GT, LQ and mask (transforms.py:93-149
                  `paired_random_crop_img_mask`) as one gather kernel per tensor (ssg_augment_crop).
`PairPool`      = the training pair pool (basicsr/models/realesrganssl_model.py:327-367 `_dequeue_and_enqueue`)
                  with the mask kept as ONE uint8 channel (the reference keeps a float copy per GT channel,
                  :339-341): the queue is never permuted in memory -- the reference's `queue[idx]` shuffle is a slot
                  table here -- and a step moves only the b samples that change places (ssg_pool_swap).

Random draws follow the reference call by call (python `random` for flips / crop origin, `torch.randperm` for the
pool), so a run seeded like the reference's produces the same batches.  Byte moves only: results are bit exact.
"""
import random

import torch

from . import _lib
from .engine import _need_gpu, _ptr, _stream


def draw_augment(hflip=True, rotation=True):
    """(hflip, vflip, rot90) of one sample: the three draws of transforms.py:174-176, in that order."""
    h = hflip and random.random() < 0.5
    v = rotation and random.random() < 0.5
    r = rotation and random.random() < 0.5
    return int(bool(h)), int(bool(v)), int(bool(r))


def draw_crop(h_lq, w_lq, gt_patch_size, scale):
    """(top, left) of the LQ patch: transforms.py:122-123."""
    lq_patch = gt_patch_size // scale
    if h_lq < lq_patch or w_lq < lq_patch:
        raise ValueError(f'LQ ({h_lq}, {w_lq}) is smaller than patch size ({lq_patch}, {lq_patch}).')
    return random.randint(0, h_lq - lq_patch), random.randint(0, w_lq - lq_patch)


def augment_crop(x, out_hw, top_left, flips=None):
    """x (B,C,Hs,Ws) float32 / uint8 on the GPU -> (B,C,Ho,Wo): flips[b] = (hflip, vflip, rot90) applied to sample b
    in the reference's order, then the crop at top_left[b] = (top, left) of the AUGMENTED sample.  One kernel."""
    _need_gpu(x)
    if x.dtype not in (torch.float32, torch.uint8):
        raise TypeError("augment_crop moves float32 or uint8 elements")
    x = x.contiguous()
    B, C, Hs, Ws = x.shape
    Ho, Wo = out_hw
    flips = flips if flips is not None else [(0, 0, 0)] * B
    rows = []
    for b in range(B):
        (top, left), (h, v, r) = top_left[b], flips[b]
        Ha, Wa = (Ws, Hs) if r else (Hs, Ws)
        if top < 0 or left < 0 or top + Ho > Ha or left + Wo > Wa:
            raise ValueError(f"crop ({top},{left})+({Ho},{Wo}) leaves the augmented sample ({Ha},{Wa})")
        rows.append([top, left, h, v, r])
    params = torch.tensor(rows, dtype=torch.int32).to(x.device, non_blocking=True)
    out = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssg_augment_crop(_ptr(x), _ptr(out), x.element_size(), B, C, Hs, Ws, Ho, Wo, _ptr(params),
                                               _stream()))
    return out


def paired_random_crop_img_mask(img_gts, img_lqs, masks, gt_patch_size, scale, flips=None):
    """Batch form of transforms.py:93-149 on GPU tensors (B,C,H,W): one (top, left) for the whole batch, GT and mask
    cropped at `scale` times the LQ origin.  `flips` (optional, per sample) folds the dataset-side `augment` of GT
    and mask into the same pass (LQ is produced from the augmented GT by the degradation pipeline upstream, so it is
    only cropped)."""
    h_lq, w_lq = img_lqs.shape[-2:]
    h_gt, w_gt = (img_gts.shape[-1], img_gts.shape[-2]) if (flips and flips[0][2]) else img_gts.shape[-2:]
    if h_gt != h_lq * scale or w_gt != w_lq * scale:
        raise ValueError(f'Scale mismatches. GT ({h_gt}, {w_gt}) is not {scale}x multiplication of LQ ({h_lq}, {w_lq}).')
    top, left = draw_crop(h_lq, w_lq, gt_patch_size, scale)
    B = img_gts.shape[0]
    lq_patch = gt_patch_size // scale
    lq = augment_crop(img_lqs, (lq_patch, lq_patch), [(top, left)] * B)
    tl = [(int(top * scale), int(left * scale))] * B
    gt = augment_crop(img_gts, (gt_patch_size, gt_patch_size), tl, flips)
    mk = augment_crop(masks, (gt_patch_size, gt_patch_size), tl, flips)
    return gt, lq, mk


class PairPool:
    """realesrganssl_model.py:327-367 with in-place swaps.  exchange(lq, gt, gt_mask) returns the batch to train on:
    while the pool fills, the incoming batch itself (and a copy is queued); once full, b samples drawn by the
    reference's shuffle leave the pool and the incoming batch takes their slots."""

    def __init__(self, queue_size):
        self.queue_size = int(queue_size)
        self.queue_ptr = 0
        self.q = None
        self.slot = None      # logical position -> physical slot (the reference's successive queue[idx] shuffles)

    def exchange(self, lq, gt, gt_mask):
        _need_gpu(lq, gt, gt_mask)
        b = lq.shape[0]
        if gt_mask.dtype != torch.uint8:
            gt_mask = (gt_mask[:, :1] == 1).to(torch.uint8)      # one uint8 channel (the reference: float, C copies)
        batch = [lq.contiguous().clone(), gt.contiguous().clone(), gt_mask[:, :1].contiguous().clone()]
        if self.q is None:
            assert self.queue_size % b == 0, f'queue size {self.queue_size} should be divisible by batch size {b}'
            self.q = [torch.zeros((self.queue_size,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in batch]
            self.slot = torch.arange(self.queue_size)
        if self.queue_ptr == self.queue_size:
            idx = torch.randperm(self.queue_size)                 # the reference's shuffle draw
            self.slot = self.slot[idx]
            slots = self.slot[:b].to(torch.int32).to(lq.device, non_blocking=True)
            L = _lib.lib()
            with torch.cuda.device(lq.device):
                for q, t in zip(self.q, batch):
                    _lib.check(L.ssg_pool_swap(_ptr(q), _ptr(t), t[0].numel() * t.element_size(), _ptr(slots), b,
                                               _stream()))
            return tuple(batch)
        for q, t in zip(self.q, batch):
            q[self.queue_ptr:self.queue_ptr + b] = t
        self.queue_ptr += b
        return tuple(batch)


class USMSharp(torch.nn.Module):
    """basicsr/utils/img_process_util.py:63-83 on the HIP engine (ssg_usm_sharp): same constructor and forward
    arguments; `img` (B,C,H,W) float32 CUDA in [0,1].  Not differentiable (the reference applies it to GT only,
    realesrganssl_model.py:165,315)."""

    def __init__(self, radius=50, sigma=0):
        super().__init__()
        self.radius = radius + 1 if radius % 2 == 0 else radius
        self.sigma = float(sigma)

    @torch.no_grad()
    def forward(self, img, weight=0.5, threshold=10):
        if not img.is_cuda:
            raise RuntimeError("ssl_amd.datapath.USMSharp: tensor must be on the GPU (there is no CPU path)")
        x = img.detach().to(torch.float32).contiguous()
        B, C, H, W = x.shape
        L = _lib.lib()
        out = torch.empty_like(x)
        nb = L.ssg_usm_scratch_bytes(B, C, H, W)
        scratch = torch.empty(max(nb, 1), dtype=torch.uint8, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.ssg_usm_sharp(x.data_ptr(), out.data_ptr(), B, C, H, W, self.radius, self.sigma, float(weight),
                                       float(threshold), scratch.data_ptr(), nb, torch.cuda.current_stream().cuda_stream))
        return out.to(img.dtype)


@torch.no_grad()
def filter2D(img, kernel):
    """basicsr/utils/img_process_util.py:7-31 on the HIP engine (ssg_filter2d): img (b,c,h,w) float32 CUDA, kernel (b,k,k)
    or (1,k,k), k odd (ValueError otherwise, like the reference) and <= 21."""
    if not (img.is_cuda and kernel.is_cuda):
        raise RuntimeError("ssl_amd.datapath.filter2D: tensors must be on the GPU (there is no CPU path)")
    k = kernel.size(-1)
    if k % 2 != 1:
        raise ValueError('Wrong kernel size')
    x = img.detach().to(torch.float32).contiguous()
    kk = kernel.detach().to(torch.float32).contiguous()
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssg_filter2d(x.data_ptr(), kk.data_ptr(), out.data_ptr(), B, C, H, W, k, kk.size(0),
                                           torch.cuda.current_stream().cuda_stream))
    return out.to(img.dtype)


class DiffJPEG(torch.nn.Module):
    """basicsr/utils/diffjpeg.py:449-487 on the HIP engine (ssg_diffjpeg), non-differentiable rounding only -- the
    configuration the model uses (`DiffJPEG(differentiable=False)`, realesrganssl_model.py:34).  forward(x, quality):
    x (B,3,H,W) float32 CUDA in [0,1]; quality a number or a (B,) tensor.  Unlike the reference, a quality tensor is not
    overwritten with its compression factors."""

    def __init__(self, differentiable=False):
        super().__init__()
        if differentiable:
            raise NotImplementedError("ssl_amd.datapath.DiffJPEG: only differentiable=False (torch.round) is built")

    @torch.no_grad()
    def forward(self, x, quality):
        if not x.is_cuda:
            raise RuntimeError("ssl_amd.datapath.DiffJPEG: tensor must be on the GPU (there is no CPU path)")
        xi = x.detach().to(torch.float32).contiguous()
        B, C, H, W = xi.shape
        if C != 3:
            raise ValueError("DiffJPEG expects RGB images (B,3,H,W)")
        out = torch.empty_like(xi)
        if isinstance(quality, (int, float)):
            qd, qh = None, float(quality)
        else:
            qd, qh = quality.detach().to(device=xi.device, dtype=torch.float32).contiguous(), 0.0
            if qd.numel() != B:
                raise ValueError("quality tensor needs one entry per sample")
        with torch.cuda.device(xi.device):
            _lib.check(_lib.lib().ssg_diffjpeg(xi.data_ptr(), out.data_ptr(), B, H, W, None if qd is None else qd.data_ptr(),
                                               qh, torch.cuda.current_stream().cuda_stream))
        return out.to(x.dtype)

