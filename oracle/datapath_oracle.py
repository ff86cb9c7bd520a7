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
