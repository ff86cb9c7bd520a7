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



# ------------------------------------------------------------------ the rest of the degradation chain ----
_RESIZE_MODES = {"area": 0, "bilinear": 1, "bicubic": 2}


def _gpu_f32(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"ssl_amd.datapath.{what}: tensor must be on the GPU (there is no CPU path)")
    return t.detach().to(torch.float32).contiguous()


@torch.no_grad()
def interpolate(img, size=None, scale_factor=None, mode="bilinear"):
    """torch.nn.functional.interpolate(img, size= | scale_factor=, mode='area' | 'bilinear' | 'bicubic') as the model
    calls it (basicsr/models/realesrganssl_model.py:185,203,224,255,280,293: align_corners unset, no antialias) on the
    HIP engine (ssg_resize).  img (B,C,H,W) float32 CUDA."""
    if mode not in _RESIZE_MODES:
        raise NotImplementedError(f"ssl_amd.datapath.interpolate: mode {mode!r} (the chain uses area, bilinear, bicubic)")
    if (size is None) == (scale_factor is None):
        raise ValueError("only one of size or scale_factor should be defined")
    x = _gpu_f32(img, "interpolate")
    B, C, H, W = x.shape
    if scale_factor is not None:
        sfh, sfw = (scale_factor, scale_factor) if not isinstance(scale_factor, (tuple, list)) else scale_factor
        import math
        Ho, Wo = int(math.floor(float(H) * sfh)), int(math.floor(float(W) * sfw))
    else:
        sfh = sfw = 0.0
        Ho, Wo = (size, size) if isinstance(size, int) else size
    if Ho <= 0 or Wo <= 0:
        raise RuntimeError(f"Input and output sizes should be greater than 0, but got input (H: {H}, W: {W}) "
                           f"output (H: {Ho}, W: {Wo})")
    out = torch.empty((B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssg_resize(_ptr(x), _ptr(out), B, C, H, W, Ho, Wo, _RESIZE_MODES[mode], float(sfh),
                                         float(sfw), _stream()))
    return out.to(img.dtype)


@torch.no_grad()
def clamp_round(img, clip=True, rounds=True):
    """`torch.clamp((out * 255.0).round(), 0, 255) / 255.` (realesrganssl_model.py:206,297) for clip = rounds = True."""
    x = _gpu_f32(img, "clamp_round")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssg_clamp_round(_ptr(x), _ptr(out), x.numel(), int(clip), int(rounds), _stream()))
    return out.to(img.dtype)


class Draws:
    """The random draws of `feed_data`, in the reference's call order and from the reference's generators (python
    `random`, `numpy.random`, torch's generator of the tensors' device).  Tests subclass it to replay the draws a run
    of the reference recorded (fixture F17); the arithmetic around the draws is the engine's either way."""

    def choices(self, population, weights):            # random.choices(...)[0]
        return random.choices(population, weights)[0]

    def choice(self, seq):
        return random.choice(seq)

    def uniform(self, lo=0.0, hi=1.0):                 # np.random.uniform
        import numpy as np
        return float(np.random.uniform(lo, hi))

    def randint(self, lo, hi):
        return random.randint(lo, hi)

    def rand(self, n, device):
        return torch.rand(n, dtype=torch.float32, device=device)

    def randn(self, shape, device):
        return torch.randn(*shape, dtype=torch.float32, device=device)

    def poisson(self, rates):
        return torch.poisson(rates)

    def jpeg_quality(self, n, lo, hi, device):          # out.new_zeros(b).uniform_(lo, hi)
        return torch.zeros(n, dtype=torch.float32, device=device).uniform_(lo, hi)

    def any_gray(self, gray, channels=3, gray_prob=None):
        """Whether a gray-noise field has to be drawn.  The reference asks `torch.sum(gray_noise) > 0` on the host (a
        device synchronisation per noise stage, degradations.py:479,619); here the answer comes from what the host
        already knows: never for images that are not 3-channel (the gray blend is defined for RGB; the reference runs
        1-channel input with gray_prob = 0) or when gray_prob == 0, otherwise always -- the field is multiplied by
        gray = 0 where unused, so the pipeline never waits for the device.  Consequence for seeded runs: with
        gray_prob > 0 the gray field is drawn even when no sample of the batch turned out gray, so torch's generator
        stream advances differently from a seeded reference run from that stage on (`ReferenceOrderDraws` below pays
        the synchronisation and keeps the reference's order)."""
        if channels != 3 or (gray_prob is not None and gray_prob <= 0):
            return False
        return True


class ReferenceOrderDraws(Draws):
    """Draws that ask the device, like the reference does, whether any sample has gray noise before drawing the gray
    field (one host synchronisation per noise stage): the generator stream then matches a seeded reference run."""

    def any_gray(self, gray, channels=3, gray_prob=None):
        return channels == 3 and bool(torch.sum(gray) > 0)


@torch.no_grad()
def add_gaussian_noise(img, sigma, gray, field_color, field_gray=None, clip=True, rounds=False):
    """add_gaussian_noise_pt (degradations.py:455-507) given its draws: sigma (B,), gray (B,) in {0,1}, field_color
    (B,C,H,W) = torch.randn, field_gray (H,W) = the ONE torch.randn(h,w) field the reference shares across the batch
    (None: no gray noise)."""
    x = _gpu_f32(img, "add_gaussian_noise")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    f32 = lambda t: None if t is None else t.detach().to(device=x.device, dtype=torch.float32).contiguous()
    sg, gr, fc, fg = f32(sigma), f32(gray), f32(field_color), f32(field_gray)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssg_gaussian_noise(_ptr(x), _ptr(out), _ptr(fc), _ptr(fg), _ptr(sg), _ptr(gr), B, C, H, W,
                                                 int(clip), int(rounds), _stream()))
    return out.to(img.dtype)


@torch.no_grad()
def poisson_rates(img, with_gray):
    """What add_poisson_noise_pt calls torch.poisson on (degradations.py:622-639): (rate_color (B,C,H,W), rate_gray
    (B,1,H,W) | None, vals (B,2)); the level census replaces the reference's per-sample torch.unique host loop."""
    x = _gpu_f32(img, "poisson_rates")
    B, C, H, W = x.shape
    L = _lib.lib()
    rate = torch.empty_like(x)
    rg = torch.empty((B, 1, H, W), dtype=torch.float32, device=x.device) if with_gray else None
    vals = torch.empty((B, 2), dtype=torch.float32, device=x.device)
    scratch = torch.empty(L.ssg_poisson_scratch_bytes(B), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.ssg_poisson_rates(_ptr(x), _ptr(rate), _ptr(rg), _ptr(vals), _ptr(scratch), B, C, H, W, _stream()))
    return rate, rg, vals


@torch.no_grad()
def add_poisson_noise(img, scale, gray, vals, draw_color, draw_gray=None, clip=True, rounds=False):
    """add_poisson_noise_pt after its draws (degradations.py:631-674): draw_color = torch.poisson(rate_color), draw_gray
    = torch.poisson(rate_gray) or None, vals from poisson_rates."""
    x = _gpu_f32(img, "add_poisson_noise")
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    f32 = lambda t: None if t is None else t.detach().to(device=x.device, dtype=torch.float32).contiguous()
    sc, gr, dc, dg, vv = f32(scale), f32(gray), f32(draw_color), f32(draw_gray), f32(vals)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ssg_poisson_noise(_ptr(x), _ptr(out), _ptr(dc), _ptr(dg), _ptr(vv), _ptr(sc), _ptr(gr), B, C,
                                                H, W, int(clip), int(rounds), _stream()))
    return out.to(img.dtype)


def random_add_gaussian_noise(img, sigma_range, gray_prob, draws, clip=True, rounds=False):
    """random_add_gaussian_noise_pt (degradations.py:536-548): draws in the reference's order (sigma, gray flags,
    [gray field], colour field)."""
    B, C, H, W = img.shape
    sigma = draws.rand(B, img.device) * (sigma_range[1] - sigma_range[0]) + sigma_range[0]
    gray = (draws.rand(B, img.device) < gray_prob).float()
    fg = draws.randn((H, W), img.device) if draws.any_gray(gray, C, gray_prob) else None
    fc = draws.randn((B, C, H, W), img.device)
    return add_gaussian_noise(img, sigma, gray, fc, fg, clip, rounds)


def random_add_poisson_noise(img, scale_range, gray_prob, draws, clip=True, rounds=False):
    """random_add_poisson_noise_pt (degradations.py:706-720)."""
    B, C = img.shape[0], img.shape[1]
    scale = draws.rand(B, img.device) * (scale_range[1] - scale_range[0]) + scale_range[0]
    gray = (draws.rand(B, img.device) < gray_prob).float()
    with_gray = draws.any_gray(gray, C, gray_prob)
    rate, rate_gray, vals = poisson_rates(img, with_gray)
    dg = draws.poisson(rate_gray) if with_gray else None
    dc = draws.poisson(rate)
    return add_poisson_noise(img, scale, gray, vals, dc, dg, clip, rounds)


class Degradation:
    """`RealESRGANSSLModel.feed_data` (basicsr/models/realesrganssl_model.py:148-316) on the GPU: USM sharpening, the
    one- or two-stage degradation chain (blur, random resize, Gaussian / Poisson noise, JPEG, [resize back + sinc]),
    clamp-round, the joint random crop and the training pair pool -- every stage a kernel of this engine, composed in
    the reference's order with the reference's option names.  `opt` is the model's option dict (the keys feed_data
    reads); `draws` supplies the random choices (default: the reference's generators).

    feed(data) takes the dataloader's dict (gt, gt_mask, kernel1, kernel2, sinc_kernel on the GPU) and returns
    dict(lq, gt, gt_usm | None, gt_mask)."""

    def __init__(self, opt, draws=None):
        self.opt = opt
        self.draws = draws if draws is not None else Draws()
        self.jpeger = DiffJPEG(differentiable=False)
        self.usm_sharpener = USMSharp()
        self.pool = PairPool(opt["queue_size"]) if opt.get("queue_size") else None

    def _random_resize(self, out, prob_key, range_key, target=None):
        """(resized, mode): the random resize of a stage (:216-224 / :246-255)."""
        opt, d = self.opt, self.draws
        updown = d.choices(['up', 'down', 'keep'], opt[prob_key])
        if updown == 'up':
            scale = d.uniform(1, opt[range_key][1])
        elif updown == 'down':
            scale = d.uniform(opt[range_key][0], 1)
        else:
            scale = 1
        mode = d.choice(['area', 'bilinear', 'bicubic'])
        if target is None:
            return interpolate(out, scale_factor=scale, mode=mode), mode                                # :224
        return interpolate(out, size=(int(target[0] * scale), int(target[1] * scale)), mode=mode), mode  # :254-255

    def _noise(self, out, sfx):
        opt, d = self.opt, self.draws
        if d.uniform() < opt['gaussian_noise_prob' + sfx]:
            return random_add_gaussian_noise(out, opt['noise_range' + sfx], opt['gray_noise_prob' + sfx], d)
        return random_add_poisson_noise(out, opt['poisson_scale_range' + sfx], opt['gray_noise_prob' + sfx], d)

    def _jpeg(self, out, range_key):
        q = self.draws.jpeg_quality(out.size(0), *self.opt[range_key], out.device)
        return self.jpeger(out, quality=q)      # (noise with clip=True has clamped to [0,1]; the JPEG kernel clamps its input too)

    @torch.no_grad()
    def feed(self, data):
        opt, d = self.opt, self.draws
        gt, gt_mask = data['gt'], data['gt_mask']
        _need_gpu(gt, gt_mask)
        sharpen = opt.get('Use_sharpen', None) is not None
        gt_usm = self.usm_sharpener(gt) if sharpen else None
        ori_h, ori_w = gt.shape[2:4]
        s = opt['scale']
        out = filter2D(gt_usm if (sharpen and opt['Sharpen_before_degra']) else gt, data['kernel1'])
        out, mode = self._random_resize(out, 'resize_prob', 'resize_range')
        out = self._noise(out, '')
        out = self._jpeg(clamp_round(out, True, False), 'jpeg_range')
        size = (ori_h // s, ori_w // s)
        if opt['degradation_order'] == 'one':
            out = interpolate(out, size=size, mode=mode)                                            # :203
        elif opt['degradation_order'] == 'two':
            if d.uniform() < opt['second_blur_prob']:
                out = filter2D(out, data['kernel2'])
            out, _ = self._random_resize(out, 'resize_prob2', 'resize_range2', target=(ori_h / s, ori_w / s))
            out = self._noise(out, '2')
            if d.uniform() < 0.5:                                                                   # :277-287
                mode = d.choice(['area', 'bilinear', 'bicubic'])
                out = filter2D(interpolate(out, size=size, mode=mode), data['sinc_kernel'])
                out = self._jpeg(clamp_round(out, True, False), 'jpeg_range2')
            else:                                                                                   # :288-294
                out = self._jpeg(clamp_round(out, True, False), 'jpeg_range2')
                mode = d.choice(['area', 'bilinear', 'bicubic'])
                out = filter2D(interpolate(out, size=size, mode=mode), data['sinc_kernel'])
        else:
            raise ValueError(f"degradation_order {opt['degradation_order']!r}")
        lq = clamp_round(out)
        # joint random crop (transforms.py:93-149): one (top, left) for the batch
        gt_size = opt['datasets']['train']['gt_size']
        lq_patch = gt_size // s
        h_lq, w_lq = lq.shape[-2:]
        if h_lq < lq_patch or w_lq < lq_patch:
            raise ValueError(f'LQ ({h_lq}, {w_lq}) is smaller than patch size ({lq_patch}, {lq_patch}).')
        top, left = d.randint(0, h_lq - lq_patch), d.randint(0, w_lq - lq_patch)
        B = gt.shape[0]
        lq = augment_crop(lq, (lq_patch, lq_patch), [(top, left)] * B)
        tl = [(top * s, left * s)] * B
        gt = augment_crop(gt.to(torch.float32), (gt_size, gt_size), tl)
        mk = gt_mask if gt_mask.dtype == torch.uint8 else gt_mask.to(torch.float32)
        mk = augment_crop(mk, (gt_size, gt_size), tl)
        if self.pool is not None:
            lq, gt, mk = self.pool.exchange(lq, gt, mk)
        gt_usm = self.usm_sharpener(gt) if sharpen else None      # (:313-315: sharpened again after the pool)
        return dict(lq=lq.contiguous(), gt=gt, gt_usm=gt_usm, gt_mask=mk)
