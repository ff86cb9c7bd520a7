"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

numpy restatement of the data-path slice, each function citing the reference lines it
follows (GAN-Based-SR/).  Pinned by tests/golden/f11_datapath.npz, produced by running the reference's own
basicsr/data/transforms.py (cv2.flip stubbed with its numpy equivalent: cv2 is not installed) and the
`_dequeue_and_enqueue` method of basicsr/models/realesrganssl_model.py (tests/golden/make_golden.py f11).
"""
import numpy as np


def augment(img_hwc, hflip, vflip, rot90):
    """transforms.py:178-187 `_augment`: horizontal flip, vertical flip, transpose(1, 0, 2), in that order."""
    a = img_hwc
    if hflip:
        a = a[:, ::-1]
    if vflip:
        a = a[::-1]
    if rot90:
        a = a.transpose(1, 0, 2)
    return np.ascontiguousarray(a)


def crop_nchw(x, top, left, size):
    """transforms.py:127,134,140 (Tensor branch): v[:, :, top:top+size, left:left+size]."""
    return np.ascontiguousarray(x[:, :, top:top + size, left:left + size])


def augment_crop_nchw(x, top, left, out_hw, flips):
    """per-sample augment (on HWC views) then crop, returned NCHW: the composition ssg_augment_crop computes."""
    out = []
    for b in range(x.shape[0]):
        a = augment(x[b].transpose(1, 2, 0), *flips[b]).transpose(2, 0, 1)
        out.append(a[:, top:top + out_hw[0], left:left + out_hw[1]])
    return np.ascontiguousarray(np.stack(out))


class PairPool:
    """realesrganssl_model.py:327-367, literally (the queue IS permuted here, like the reference's queue[idx])."""

    def __init__(self, queue_size):
        self.queue_size, self.ptr, self.q = queue_size, 0, None

    def exchange(self, tensors, idx_fn):
        """tensors: list of (b, ...) arrays; idx_fn() -> the permutation drawn when the pool is full."""
        b = tensors[0].shape[0]
        if self.q is None:
            self.q = [np.zeros((self.queue_size,) + t.shape[1:], t.dtype) for t in tensors]
        if self.ptr == self.queue_size:
            idx = idx_fn()
            self.q = [q[idx] for q in self.q]
            out = [q[:b].copy() for q in self.q]
            for q, t in zip(self.q, tensors):
                q[:b] = t
            return out
        for q, t in zip(self.q, tensors):
            q[self.ptr:self.ptr + b] = t
        self.ptr += b
        return [t.copy() for t in tensors]


def gaussian_kernel_1d(ksize, sigma=0.0):
    """cv2.getGaussianKernel(ksize, sigma) (n,) float64 as OpenCV documents it (imgproc, getGaussianKernel): the fixed
    tables for ksize <= 7 with sigma <= 0, else exp(-(i-(ksize-1)/2)^2 / (2 sigma^2)) normalised, sigma <= 0 meaning
    0.3*((ksize-1)*0.5 - 1) + 0.8.  cv2 is not installed: this function is pinned by the documentation only (the
    callers below are pinned by the reference's own code, fixture F12)."""
    fixed = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in fixed:
        return np.asarray(fixed[ksize], np.float64)
    sg = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
    x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
    k = np.exp(-0.5 * x * x / (sg * sg))
    return k / k.sum()


def filter2d_reflect(img, k1d):
    """img_process_util.py:7-31 `filter2D` for one kernel shared by the batch: reflect-pad by k//2 (F.pad 'reflect'),
    correlate every (b, c) plane with the k x k kernel -- here with the separable factors of USMSharp's outer product
    (img_process_util.py:71), in the dtype of `img`."""
    r = len(k1d) // 2
    p = np.pad(img, ((0, 0), (0, 0), (r, r), (r, r)), mode="reflect")
    H, W = img.shape[-2:]
    t = sum(k1d[i] * p[..., :, i:i + W] for i in range(len(k1d)))
    return sum(k1d[i] * t[..., i:i + H, :] for i in range(len(k1d)))


def usm_sharp(img, radius=50, sigma=0.0, weight=0.5, threshold=10.0, return_parts=False):
    """USMSharp.__init__ + forward (img_process_util.py:63-83) on (B,C,H,W) in [0,1], computed in float64."""
    img = np.asarray(img, np.float64)
    ksize = radius + 1 if radius % 2 == 0 else radius
    k = gaussian_kernel_1d(ksize, sigma)
    blur = filter2d_reflect(img, k)
    residual = img - blur
    mask = (np.abs(residual) * 255 > threshold).astype(np.float64)
    soft = filter2d_reflect(mask, k)
    sharp = np.clip(img + weight * residual, 0, 1)
    out = soft * sharp + (1 - soft) * img
    return (out, residual, mask) if return_parts else out


def filter2d(img, kernels):
    """img_process_util.py:7-31 `filter2D`: img (B,C,H,W), kernels (B,k,k) or (1,k,k), k odd; reflect-pad by k//2 and
    correlate (F.conv2d does not flip) every plane of sample b with kernel b (the shared one).  float64."""
    img = np.asarray(img, np.float64)
    kernels = np.asarray(kernels, np.float64)
    k = kernels.shape[-1]
    if k % 2 != 1:
        raise ValueError("Wrong kernel size")
    r = k // 2
    B, C, H, W = img.shape
    p = np.pad(img, ((0, 0), (0, 0), (r, r), (r, r)), mode="reflect")
    out = np.zeros_like(img)
    for b in range(B):
        kb = kernels[0 if kernels.shape[0] == 1 else b]
        for ky in range(k):
            for kx in range(k):
                out[b] += kb[ky, kx] * p[b, :, ky:ky + H, kx:kx + W]
    return out


# ---- DiffJPEG(differentiable=False) (basicsr/utils/diffjpeg.py), float64 with the module's float32 parameters ----
_JPEG_Y = np.array([[16, 11, 10, 16, 24, 40, 51, 61], [12, 12, 14, 19, 26, 58, 60, 55], [14, 13, 16, 24, 40, 57, 69, 56],
                    [14, 17, 22, 29, 51, 87, 80, 62], [18, 22, 37, 56, 68, 109, 103, 77], [24, 35, 55, 64, 81, 104, 113, 92],
                    [49, 64, 78, 87, 103, 121, 120, 101], [72, 92, 95, 98, 112, 100, 103, 99]], np.float32).T   # :14-19
_JPEG_C = np.full((8, 8), 99, np.float32)                                                                      # :20-23
_JPEG_C[:4, :4] = np.array([[17, 18, 24, 47], [18, 21, 26, 66], [24, 26, 56, 99], [47, 66, 99, 99]]).T


def jpeg_quality_to_factor(quality):
    """diffjpeg.py:32-46."""
    quality = 5000. / quality if quality < 50 else 200. - quality * 2
    return quality / 100.


def diffjpeg(x, quality, return_quotients=False):
    """DiffJPEG(differentiable=False).forward (diffjpeg.py:449-487) on x (B,3,H,W) in [0,1]; quality: scalar or (B,).
    float64 arithmetic on the module's float32 constants (what `.double()` leaves them as).  return_quotients: also
    the pre-rounding quotients of every coefficient, (B, Hp, Wp) for Y and (B, Hp/2, Wp/2) x 2 for chroma, so that
    a caller can see which roundings are decided at fp32."""
    x = np.asarray(x, np.float64)
    B, _, H, W = x.shape
    q = np.broadcast_to(np.asarray(quality, np.float64), (B,))
    # (the tensor branch evaluates quality_to_factor in float32, :465-466)
    factor = np.array([np.float32(np.float32(5000.) / np.float32(v)) / np.float32(100.) if v < 50
                       else np.float32(np.float32(200.) - np.float32(v) * np.float32(2)) / np.float32(100.) for v in q],
                      np.float64) if np.ndim(quality) else np.full(B, jpeg_quality_to_factor(float(quality)))
    hp, wp = (16 - H % 16) % 16, (16 - W % 16) % 16
    img = np.pad(x, ((0, 0), (0, 0), (0, hp), (0, wp))) * 255                                   # :474-480, :232
    M1 = np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]], np.float32).astype(np.float64)
    ycc = np.einsum("bchw,jc->bjhw", img, M1) + np.array([0., 128., 128.])[None, :, None, None]  # :52-70
    Hp, Wp = H + hp, W + wp
    planes = [ycc[:, 0], ycc[:, 1].reshape(B, Hp // 2, 2, Wp // 2, 2).mean((2, 4)),
              ycc[:, 2].reshape(B, Hp // 2, 2, Wp // 2, 2).mean((2, 4))]                            # :86-95
    cosv = np.array([[np.cos((2 * a + 1) * u * np.pi / 16) for u in range(8)] for a in range(8)])
    T = np.einsum("xu,yv->xyuv", cosv, cosv).astype(np.float32).astype(np.float64)                # :125-128
    alpha = np.array([1. / np.sqrt(2)] + [1] * 7)
    scale = (np.outer(alpha, alpha) * 0.25).astype(np.float32).astype(np.float64)                 # :129-131
    alpha2 = np.outer(alpha, alpha).astype(np.float32).astype(np.float64)                         # :301-302
    Ti = np.einsum("ux,vy->xyuv", cosv, cosv).astype(np.float32).astype(np.float64)               # :303-306
    rec, quots = [], []
    for k, pl in enumerate(planes):
        h, w = pl.shape[1:]
        blocks = pl.reshape(B, h // 8, 8, w // 8, 8).transpose(0, 1, 3, 2, 4)                      # :110-118
        coef = scale * np.einsum("bmnxy,xyuv->bmnuv", blocks - 128, T)                            # :142-145
        tab = (_JPEG_Y if k == 0 else _JPEG_C).astype(np.float64)[None, None, None] * factor[:, None, None, None, None]
        quot = coef / tab                                                                          # :163-170
        quots.append(quot.transpose(0, 1, 3, 2, 4).reshape(B, h, w))
        deq = np.round(quot) * tab                                                                 # np.round = half to even = torch.round
        pix = 0.25 * np.einsum("bmnxy,xyuv->bmnuv", deq * alpha2, Ti) + 128                       # :317-321
        rec.append(pix.transpose(0, 1, 3, 2, 4).reshape(B, h, w))                                  # :338-345
    up = lambda c: np.repeat(np.repeat(c, 2, axis=1), 2, axis=2)                                   # :362-372
    ycc2 = np.stack([rec[0], up(rec[1]), up(rec[2])], 1) + np.array([0., -128., -128.])[None, :, None, None]
    M2 = np.array([[1., 0., 1.402], [1, -0.344136, -0.714136], [1, 1.772, 0]], np.float32).astype(np.float64)
    rgb = np.clip(np.einsum("bchw,jc->bjhw", ycc2, M2), 0, 255) / 255                             # :394-398, :445-446
    out = rgb[:, :, :H, :W]
    return (out, quots) if return_quotients else out


# ------------------------------------------------------------------ resizes of the degradation chain ----
# F.interpolate(x, scale_factor=s | size=(h, w), mode='area' | 'bilinear' | 'bicubic') as the reference calls it
# (GAN-Based-SR/basicsr/models/realesrganssl_model.py:185,203,224,255,280,293): align_corners=None (= False),
# antialias=False, recompute_scale_factor=None.  Semantics restated from torch's documented behaviour and pinned by
# fixture F15 (tests/golden/make_golden.py f15: the same calls run by torch on the CPU in fp32 and fp64):
#   output size   size=, or floor(in * scale_factor)                               (torch.nn.functional.interpolate)
#   bilinear      src = scale * (dst + 0.5) - 0.5 clamped at 0; scale = 1 / scale_factor when a scale_factor was
#                 given (it is NOT recomputed from the sizes), in / out otherwise; taps i0 = int(src), i0 + 1
#                 (clamped to the last pixel), weights 1 - t, t
#   bicubic       same source index without the clamp, i0 = floor(src), taps i0 - 1 .. i0 + 2 clamped to the image,
#                 cubic convolution coefficients with A = -0.75
#   area          adaptive_avg_pool2d: window [floor(o in / out), ceil((o + 1) in / out)) per axis, plain mean
def interp_out_size(n_in, size=None, scale_factor=None):
    return int(size) if size is not None else int(np.floor(np.float64(n_in) * np.float64(scale_factor)))


def _interp_scale(n_in, n_out, scale_factor, dt):
    # area_pixel_compute_scale: 1 / scale_factor in double rounded to the kernel's type, else in / out in that type
    if scale_factor is not None and scale_factor > 0:
        return dt(1.0 / np.float64(scale_factor))
    return dt(n_in) / dt(n_out)


def _linear_taps(n_in, n_out, scale_factor, dt):
    scale = _interp_scale(n_in, n_out, scale_factor, dt)
    o = np.arange(n_out).astype(dt)
    src = scale * (o + dt(0.5)) - dt(0.5)
    src = np.maximum(src, dt(0)).astype(dt)
    i0 = np.minimum(src.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    t = np.clip((src - i0.astype(dt)).astype(dt), dt(0), dt(1))
    return i0, i1, (dt(1) - t).astype(dt), t


def _cubic_taps(n_in, n_out, scale_factor, dt):
    A = dt(-0.75)
    scale = _interp_scale(n_in, n_out, scale_factor, dt)
    o = np.arange(n_out).astype(dt)
    src = (scale * (o + dt(0.5)) - dt(0.5)).astype(dt)
    i0 = np.floor(src).astype(np.int64)
    t = (src - i0.astype(dt)).astype(dt)

    def cc1(x):   # |x| <= 1
        return (((A + dt(2)) * x - (A + dt(3))) * x * x + dt(1)).astype(dt)

    def cc2(x):   # 1 < |x| < 2
        return ((((A * x - dt(5) * A) * x + dt(8) * A) * x) - dt(4) * A).astype(dt)

    w = np.stack([cc2(t + dt(1)), cc1(t), cc1(dt(1) - t), cc2(dt(2) - t)], 0)     # (4, n_out)
    idx = np.stack([np.clip(i0 + k, 0, n_in - 1) for k in (-1, 0, 1, 2)], 0)
    return idx, w


def interpolate(x, size=None, scale_factor=None, mode="bilinear", dtype=np.float32):
    """x (B,C,H,W) -> F.interpolate(x, size= | scale_factor=, mode=) evaluated in `dtype` (np.float32 like the fp32
    run of the reference, np.float64 for the yardstick)."""
    dt = dtype
    x = np.asarray(x, dt)
    B, C, H, W = x.shape
    sf_h = sf_w = None
    if scale_factor is not None:
        sf_h, sf_w = (scale_factor, scale_factor) if np.isscalar(scale_factor) else scale_factor
        Ho, Wo = interp_out_size(H, None, sf_h), interp_out_size(W, None, sf_w)
    else:
        Ho, Wo = size
    if mode == "area":
        out = np.empty((B, C, Ho, Wo), dt)
        ys = [(int(np.floor(np.float32(o * H) / np.float32(Ho))), int(np.ceil(np.float32((o + 1) * H) / np.float32(Ho)))) for o in range(Ho)]
        xs = [(int(np.floor(np.float32(o * W) / np.float32(Wo))), int(np.ceil(np.float32((o + 1) * W) / np.float32(Wo)))) for o in range(Wo)]
        for oy, (y0, y1) in enumerate(ys):
            for ox, (x0, x1) in enumerate(xs):
                win = x[:, :, y0:y1, x0:x1].reshape(B, C, -1)
                acc = np.zeros((B, C), dt)
                for k in range(win.shape[-1]):     # plain running sum in row-major order, then one division
                    acc = (acc + win[:, :, k]).astype(dt)
                out[:, :, oy, ox] = acc / dt((y1 - y0) * (x1 - x0))
        return out
    if mode == "bilinear":
        y0, y1, wy0, wy1 = _linear_taps(H, Ho, sf_h, dt)
        x0, x1, wx0, wx1 = _linear_taps(W, Wo, sf_w, dt)
        top = (x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1).astype(dt)
        bot = (x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1).astype(dt)
        return (top * wy0[:, None] + bot * wy1[:, None]).astype(dt)
    if mode == "bicubic":
        iy, wy = _cubic_taps(H, Ho, sf_h, dt)
        ix, wx = _cubic_taps(W, Wo, sf_w, dt)
        out = np.zeros((B, C, Ho, Wo), dt)
        for a in range(4):
            row = np.zeros((B, C, Ho, Wo), dt)
            for b in range(4):
                row = (row + x[:, :, iy[a]][:, :, :, ix[b]] * wx[b]).astype(dt)
            out = (out + row * wy[a][:, None]).astype(dt)
        return out
    raise ValueError(mode)


# ------------------------------------------------------------------ noise of the degradation chain ----
# basicsr/data/degradations.py `*_pt` functions with the RANDOM FIELDS AS INPUTS (torch.randn / torch.poisson draws
# cannot be reproduced by another generator; everything after the draw is deterministic arithmetic, restated here
# in the order the reference evaluates it, fp32 by default).
def clip_round(out, clip=True, rounds=False, dtype=np.float32):
    """The common tail of add_*_noise_pt (degradations.py:501-507, 668-674); clip=rounds=True is also the final
    `torch.clamp((out * 255.0).round(), 0, 255) / 255.` of feed_data (realesrganssl_model.py:206,297).
    torch.round = round half to even = np.rint."""
    dt = dtype
    out = np.asarray(out, dt)
    if clip and rounds:
        return (np.clip(np.rint(out * dt(255.0)), 0, 255) / dt(255.0)).astype(dt)
    if clip:
        return np.clip(out, 0, 1).astype(dt)
    if rounds:
        return (np.rint(out * dt(255.0)) / dt(255.0)).astype(dt)
    return out


def gaussian_noise(img, sigma, gray, field_color, field_gray=None, clip=True, rounds=False, dtype=np.float32):
    """add_gaussian_noise_pt(img, sigma (b,), gray_noise (b,) in {0,1}) (degradations.py:455-507) given
    field_color = the torch.randn(b,c,h,w) draw and field_gray = the torch.randn(h,w) draw (ONE h x w field shared
    by the batch: `torch.randn(*img.size()[2:4]) * sigma / 255.` broadcasts (h,w) against (b,1,1,1), :481-482), used
    only when sum(gray) > 0."""
    dt = dtype
    img = np.asarray(img, dt)
    b = img.shape[0]
    s = np.asarray(sigma, dt).reshape(b, 1, 1, 1)
    g = np.asarray(gray, dt).reshape(b, 1, 1, 1)
    noise = (np.asarray(field_color, dt) * s / dt(255.0)).astype(dt)
    if g.sum() > 0:
        ng = (np.asarray(field_gray, dt)[None, None] * s / dt(255.0)).astype(dt)     # (b,1,h,w)
        noise = (noise * (dt(1) - g) + ng * g).astype(dt)
    return clip_round((img + noise).astype(dt), clip, rounds, dt)


def rgb_to_grayscale(img, dtype=np.float32):
    """torchvision.transforms.functional_tensor.rgb_to_grayscale (documented formula; torchvision is not installed in
    the build container): 0.2989 R + 0.587 G + 0.114 B, one output channel."""
    dt = dtype
    img = np.asarray(img, dt)
    return (dt(0.2989) * img[:, 0:1] + dt(0.587) * img[:, 1:2] + dt(0.114) * img[:, 2:3]).astype(dt)


def poisson_vals(img_rounded):
    """2 ** ceil(log2(number of distinct values)) per sample (degradations.py:627-629, 636-638)."""
    b = img_rounded.shape[0]
    n = [len(np.unique(img_rounded[i])) for i in range(b)]
    return np.array([2.0 ** np.ceil(np.log2(v)) for v in n], np.float64)


def poisson_rates(img, any_gray, dtype=np.float32):
    """What torch.poisson is called on (degradations.py:622-639): (img_r * vals, vals) for the colour image and, when
    any sample asks for gray noise, for its grayscale.  img_r = clamp(round(img * 255), 0, 255) / 255."""
    dt = dtype
    img = np.asarray(img, dt)
    b = img.shape[0]
    out = {}
    if any_gray:
        gr = clip_round(rgb_to_grayscale(img, dt), True, True, dt)
        vg = poisson_vals(gr).astype(dt).reshape(b, 1, 1, 1)
        out.update(gray_r=gr, vals_gray=vg, rate_gray=(gr * vg).astype(dt))
    ir = clip_round(img, True, True, dt)
    vc = poisson_vals(ir).astype(dt).reshape(b, 1, 1, 1)
    out.update(img_r=ir, vals=vc, rate=(ir * vc).astype(dt))
    return out


def poisson_noise(img, scale, gray, draw_color, draw_gray=None, clip=True, rounds=False, dtype=np.float32):
    """add_poisson_noise_pt(img, scale (b,), gray_noise (b,)) (degradations.py:601-674) given draw_color = the
    torch.poisson(img_r * vals) draw and draw_gray = the torch.poisson(gray_r * vals_gray) draw (b,1,h,w)."""
    dt = dtype
    img = np.asarray(img, dt)
    b = img.shape[0]
    g = np.asarray(gray, dt).reshape(b, 1, 1, 1)
    any_gray = bool(g.sum() > 0)
    r = poisson_rates(img, any_gray, dt)
    noise = (np.asarray(draw_color, dt) / r["vals"] - r["img_r"]).astype(dt)
    if any_gray:
        ng = (np.asarray(draw_gray, dt) / r["vals_gray"] - r["gray_r"]).astype(dt)
        noise = (noise * (dt(1) - g) + ng * g).astype(dt)
    noise = (noise * np.asarray(scale, dt).reshape(b, 1, 1, 1)).astype(dt)
    return clip_round((img + noise).astype(dt), clip, rounds, dt)


# ------------------------------------------------------------------ the degradation chain, composed ----
class RecordedDraws:
    """The random draws of one `feed_data` call replayed from fixture F17 in call order (names as recorded by
    tests/golden/make_golden.py f17: 'random.choices', 'random.choice', 'random.randint', 'np.random.uniform',
    'torch.rand', 'torch.randn', 'torch.poisson', 'torch.randperm', 'jpeg_q')."""

    def __init__(self, names, values):
        self.names, self.values, self.i = list(names), list(values), 0

    def pop(self, name):
        assert self.i < len(self.names), f"draw {self.i}: nothing left (wanted {name})"
        assert self.names[self.i] == name, f"draw {self.i}: recorded {self.names[self.i]}, asked for {name}"
        v = self.values[self.i]
        self.i += 1
        return v

    def peek(self):
        return self.names[self.i] if self.i < len(self.names) else None

    def done(self):
        return self.i == len(self.names)


def feed_data(gt, mask, kernel1, kernel2, sinc_kernel, opt, draws, dtype=np.float64):
    """realesrganssl_model.py:148-316 (degradation_order 'two') up to and including the joint random crop, with the
    draws replayed from `draws` (RecordedDraws).  Returns dict(lq_full, gt, gt_usm | None, lq, mask).  `dtype` is the
    arithmetic type of the element-wise stages (resize, noise, rounding); blur / JPEG / USM are the float64 oracles."""
    dt = dtype
    scale_sr = opt["scale"]
    gt = np.asarray(gt, np.float64)
    sharpen = opt.get("Use_sharpen") is not None
    gt_usm = usm_sharp(gt) if sharpen else None                                    # :164-165
    ori_h, ori_w = gt.shape[2:]
    out = filter2d(gt_usm if (sharpen and opt["Sharpen_before_degra"]) else gt, kernel1)   # :211-214

    def random_resize(out, prob_key, range_key, second):
        updown = str(draws.pop("random.choices")[0])                               # :216 / :246
        if updown == "up":
            s = float(draws.pop("np.random.uniform"))
        elif updown == "down":
            s = float(draws.pop("np.random.uniform"))
        else:
            s = 1
        mode = str(draws.pop("random.choice"))
        if second:                                                                 # :254-255
            size = (int(ori_h / scale_sr * s), int(ori_w / scale_sr * s))
            return interpolate(out, size=size, mode=mode, dtype=dt)
        return interpolate(out, scale_factor=s, mode=mode, dtype=dt)               # :224

    def noise(out, sfx):
        gray_prob = opt["gray_noise_prob" + sfx]
        if float(draws.pop("np.random.uniform")) < opt["gaussian_noise_prob" + sfx]:      # :227 / :258
            lo, hi = opt["noise_range" + sfx]
            sigma = (np.asarray(draws.pop("torch.rand"), np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)
            gray = (np.asarray(draws.pop("torch.rand"), np.float32) < gray_prob).astype(np.float32)
            fg = draws.pop("torch.randn") if gray.sum() > 0 else None
            fc = draws.pop("torch.randn")
            return gaussian_noise(out, sigma, gray, fc, fg, True, False, dt)
        lo, hi = opt["poisson_scale_range" + sfx]
        sc = (np.asarray(draws.pop("torch.rand"), np.float32) * np.float32(hi - lo) + np.float32(lo)).astype(np.float32)
        gray = (np.asarray(draws.pop("torch.rand"), np.float32) < gray_prob).astype(np.float32)
        dg = draws.pop("torch.poisson") if gray.sum() > 0 else None
        dc = draws.pop("torch.poisson")
        return poisson_noise(out, sc, gray, dc, dg, True, False, dt)

    def jpeg(out):
        q = np.asarray(draws.pop("jpeg_q"), np.float32)
        return diffjpeg(np.clip(out, 0, 1), q)                                     # :199-201 etc.

    out = random_resize(out, "resize_prob", "resize_range", False)
    out = noise(out, "")
    out = jpeg(out)
    if float(draws.pop("np.random.uniform")) < opt["second_blur_prob"]:             # :243-244
        out = filter2d(out, kernel2)
    out = random_resize(out, "resize_prob2", "resize_range2", True)
    out = noise(out, "2")
    size = (ori_h // scale_sr, ori_w // scale_sr)
    if float(draws.pop("np.random.uniform")) < 0.5:                                 # :277-287
        mode = str(draws.pop("random.choice"))
        out = filter2d(interpolate(out, size=size, mode=mode, dtype=dt), sinc_kernel)
        out = jpeg(out)
    else:                                                                          # :288-294
        out = jpeg(out)
        mode = str(draws.pop("random.choice"))
        out = filter2d(interpolate(out, size=size, mode=mode, dtype=dt), sinc_kernel)
    lq_full = clip_round(out, True, True, dt)                                       # :297
    gt_size = opt["datasets"]["train"]["gt_size"]
    lqp = gt_size // scale_sr
    top, left = int(draws.pop("random.randint")), int(draws.pop("random.randint"))  # transforms.py:122-123
    crop = lambda a, t, l, n: np.ascontiguousarray(a[:, :, t:t + n, l:l + n])
    res = dict(lq_full=lq_full, lq=crop(lq_full, top, left, lqp), gt=crop(gt, top * scale_sr, left * scale_sr, gt_size),
               mask=crop(np.asarray(mask), top * scale_sr, left * scale_sr, gt_size), gt_usm=None, top_left=(top, left))
    if sharpen:
        # (:299-315: GT and its sharpened copy are cropped together; once the pool has exchanged samples the
        # sharpened copy is recomputed from the cropped GT -- here, pool not yet full, it is usm(cropped GT))
        res["gt_usm"] = usm_sharp(res["gt"])
    return res
